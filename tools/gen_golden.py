#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the CPU oracle (BATCHED mode).

The reference ships no golden vectors and Taichi cannot be installed here; these fixtures are REGRESSION vectors made by the oracle and do not pin it
(the vectors that do come from the reference's own source: tools/gen_ref_golden.py, tests/golden/ref_*.npz)
to the reference -- they freeze the oracle's own output so that an accidental change of the restated semantics (or of the
HIP path that is compared against the same files on the GPU box) is caught.  Inputs are the deterministic synthetic stream."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import BATCHED, OracleOctomap, OracleTSDF  # noqa: E402
from taichislam_amd.utils import synthetic as syn  # noqa: E402
from util import lin  # noqa: E402

CFG = dict(map_scale=[10.24, 10.24], voxel_scale=0.08, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3,
           internal_voxels=10, recast_step=2)
H, W, NF = 60, 80, 3


def frames():
    K = syn.scaled_intrinsics(H, W)
    out = []
    for f in range(NF):
        R, T = syn.camera_pose(f)
        out.append((R, T, syn.sphere_room_depth(R, T, H, W, K=K)))
    return K, out


def main():
    K, fr = frames()
    o = OracleTSDF(**CFG)
    o.set_intrinsics(K)
    stats = [o.integrate_depth(R, T, d, mode=BATCHED) for R, T, d in fr]
    e = o.export_sparse()
    order = np.argsort(lin(e["indices"]))
    v, n, _, ntri = o.generate_mesh(1, 0.4, 200000)
    tri = np.concatenate([v.reshape(-1, 9), n.reshape(-1, 9)], 1)
    tri = tri[np.lexsort(tri.T[::-1])]
    oc = OracleOctomap(map_scale=[12.8, 12.8], voxel_scale=0.1, min_occupy_thres=1, max_ray_length=5.0, K=2, max_submap_num=4)
    oc.set_intrinsics(K)
    for R, T, d in fr:
        oc.integrate_depth(R, T, d)
    li, lc = oc.export_leaves()
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tsdf_small.npz"),
                        indices=e["indices"][order], tsdf_bits=e["TSDF"].view(np.uint16)[order], w_bits=e["W_TSDF"].view(np.uint16)[order],
                        occupy=e["occupy"][order], stats=np.array([[s[k] for k in sorted(s)] for s in stats], np.int64),
                        stat_keys=np.array(sorted(stats[0])), mesh=tri.astype(np.float32), ntri=ntri,
                        octo_idx=li, octo_cnt=lc)
    print("wrote tests/golden/tsdf_small.npz:", e["TSDF"].shape[0], "voxels,", ntri, "triangles,", li.shape[0], "octomap leaves")


C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3,
          internal_voxels=10, recast_step=2)
C2_FRAMES, C2_STRIDE = 2, 64


def digest_export(e):
    """Order-independent digest of a sparse export: SHA-256 of the index / TSDF / W / occupancy arrays sorted by voxel index, the
    voxel count, and every C2_STRIDE-th voxel in full (so a mismatch can be located)."""
    import hashlib
    order = np.argsort(lin(e["indices"]), kind="stable")
    idx = np.ascontiguousarray(e["indices"][order]); t = np.ascontiguousarray(np.asarray(e["TSDF"]).view(np.uint16)[order])
    w = np.ascontiguousarray(np.asarray(e["W_TSDF"]).view(np.uint16)[order]); occ = np.ascontiguousarray(e["occupy"][order])
    sha = lambda a: hashlib.sha256(a.tobytes()).hexdigest()
    return {"n": np.int64(idx.shape[0]), "sha_idx": sha(idx), "sha_tsdf": sha(t), "sha_w": sha(w), "sha_occ": sha(occ),
            "s_idx": idx[::C2_STRIDE], "s_tsdf": t[::C2_STRIDE], "s_w": w[::C2_STRIDE], "s_occ": occ[::C2_STRIDE]}


def main_c2():
    """BASELINE configs[1] at full size (640x480 -> 512^3 / 2 cm), two frames, both oracle modes, as digests + a 1/64 sample."""
    from oracle import FAITHFUL
    fr = list(syn.sphere_room_stream(C2_FRAMES))
    out = {}
    for name, mode in (("batched", BATCHED), ("faithful", FAITHFUL)):
        o = OracleTSDF(**C2)
        o.set_intrinsics(syn.K_DEPTH)
        stats = [o.integrate_depth(R, T, d, mode=mode) for R, T, d in fr]
        for k, v in digest_export(o.export_sparse()).items():
            out[f"{name}_{k}"] = v
        out[f"{name}_stats"] = np.array([[s[k] for k in sorted(s)] for s in stats], np.int64)
    out["stat_keys"] = np.array(sorted(stats[0]))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tsdf_c2_digest.npz"), **out)
    print("wrote tests/golden/tsdf_c2_digest.npz:", int(out["batched_n"]), "voxels")


if __name__ == "__main__":
    main()
    main_c2()
