"""The reference's orchestration against real kernels, without vendoring it.

tests/golden/submap_trace.json is the record of every call the REFERENCE's taichi_slam/mapping/submap_mapping.py (:126-181 submap life
cycle, :226-253 wire format) made into its map classes for one scenario (tools/gen_submap_trace.py, generated on the dev box where the
reference tree lives; arguments stored by value).
  * CPU: the package's own SubmapMapping, driven through the same scenario on the same recording stand-ins, must make exactly those
    calls -- same order, same arguments bit for bit, same buffers on the wire.
  * GPU: the recorded calls are replayed one by one on the HIP-backed DenseTSDF shims; every value the reference looked at must come back
    as recorded, and the global map the calls build must equal the oracle's replay of the same calls bit for bit."""
import json
import os

import numpy as np
import pytest

import submap_trace as st

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "submap_trace.json")


def _gold():
    return json.load(open(GOLD))


def test_package_orchestration_reproduces_the_reference_trace():
    from taichislam_amd.mapping import submap_mapping as mine
    gold = _gold()
    old = (mine.DenseTSDF, mine.Octomap)
    mine.DenseTSDF, mine.Octomap = st.RecTSDF, st.RecOcto          # the package class picks its defaults by map type
    try:
        got = st.record(mine.SubmapMapping)
    finally:
        mine.DenseTSDF, mine.Octomap = old
    assert len(gold["trace"]) == len(got["trace"]) >= 30
    for i, (a, b) in enumerate(zip(gold["trace"], got["trace"])):
        assert (a["map"], a["method"]) == (b["map"], b["method"]), f"call {i}: reference {a['map']}.{a['method']} != package {b['map']}.{b['method']}"
        assert st.same(st.dec(a["args"], {}) if a["method"] == "__init__" else _strip(a["args"]), st.dec(b["args"], {}) if b["method"] == "__init__" else _strip(b["args"])), f"call {i} ({a['method']}): arguments differ"
        assert st.same(st.dec(a["ret"], {}), st.dec(b["ret"], {})), f"call {i} ({a['method']}): return value differs"
    assert st.same([st.dec(x, {}) for x in gold["sent"]], [st.dec(x, {}) for x in got["sent"]]) and len(gold["sent"]) == 2
    assert gold["submaps"] == got["submaps"] == {"0": 0, "3": 1, "6": 2}


def _strip(args):
    """arguments with map / field references left as their tags (they are compared by tag, not resolved)"""
    def walk(x):
        if isinstance(x, list):
            return [walk(v) for v in x]
        if isinstance(x, dict) and ("__map__" in x or "__field__" in x):
            return tuple(sorted(x.items()))
        return st.dec(x, {}) if isinstance(x, dict) else x
    return walk(args)


@pytest.mark.gpu
def test_reference_trace_replayed_on_the_hip_shims(hip_lib):
    from oracle import BATCHED, OracleTSDF
    from taichislam_amd.mapping import DenseTSDF
    from util import sort_export
    gold = _gold()
    maps, oracles = {}, {}
    nrecast = 0
    for i, c in enumerate(gold["trace"]):
        tag, method = c["map"], c["method"]
        if method == "__init__":
            kw = st.dec(c["args"][0], {})
            maps[tag] = DenseTSDF(**kw)
            oracles[tag] = OracleTSDF(**{k: v for k, v in kw.items() if k in ("map_scale", "voxel_scale", "num_voxel_per_blk_axis", "max_ray_length", "max_submap_num", "is_global_map")})
            continue
        args = st.dec(c["args"], maps)
        m, o = maps[tag], oracles[tag]
        if method.startswith("setattr "):
            setattr(m, method.split(" ", 1)[1], args[0])
            continue
        ret = getattr(m, method)(*args)
        want = st.dec(c["ret"], maps)
        if method in ("get_active_submap_id", "switch_to_next_submap"):
            assert ret == want, f"call {i}: {method} returned {ret}, the reference saw {want}"
        # the oracle replays the calls that change a map
        if method == "set_dep_camera_intrinsic":
            o.set_intrinsics(args[0])
        elif method == "set_base_pose_submap":
            o.set_base_pose_submap(args[0], args[1], args[2])
        elif method == "switch_to_next_submap":
            o.set_active_submap(o.get_active_submap() + 1)
        elif method == "recast_depth_to_map":
            o.integrate_depth(args[0], args[1], args[2], mode=BATCHED); nrecast += 1
        elif method == "fuse_submaps":
            o.fuse_submaps(oracles[c["args"][0]["__map__"]], mode=BATCHED)
        elif method == "export_submap":
            e = sort_export(ret); w = sort_export(o.export_sparse())      # what the reference puts on the wire next
            assert np.array_equal(e["indices"], w["indices"]) and np.array_equal(e["TSDF"], w["TSDF"]) and np.array_equal(e["W_TSDF"], w["W_TSDF"])
    assert nrecast == st.NFRAMES
    gtag = [t for t in maps if t.startswith("global")][0]
    a, b = sort_export(maps[gtag].export_submap()), sort_export(oracles[gtag].export_sparse())
    assert np.array_equal(a["indices"], b["indices"]) and a["indices"].shape[0] > 50000
    ok = ~np.isnan(a["TSDF"].view(np.float16))
    assert np.array_equal(a["TSDF"][ok], b["TSDF"][ok]) and np.array_equal(a["W_TSDF"], b["W_TSDF"]) and np.array_equal(a["occupy"], b["occupy"])
