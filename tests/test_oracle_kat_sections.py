"""Known-answer tests for the oracle's fusion, marching-cubes, Octomap and ESDF sections (hand-computed; the vectors from the reference's own source are in tests/test_ref_golden.py:
these pin the restatement to hand / numpy replays written directly from the reference lines, independent of the oracle's C code).

  fusion          taichi_slam/mapping/dense_tsdf.py:272-307, mapping_common.py:221-232
  marching cubes  taichi_slam/mapping/marching_cube_mesher.py:44-60 (vertexInterp), :84-93 (generate_normal), :127-177
  Octomap insert  taichi_slam/mapping/taichi_octomap.py:116-124,134-145, mapping_common.py:240-266
  ESDF            taichi_slam/mapping/dense_esdf.py:228-230,282-297,313-329 (definition)"""
import numpy as np

from oracle import BATCHED, FAITHFUL, OracleOctomap, OracleTSDF
from util import sort_export

f16, f32 = np.float16, np.float32
CFG = dict(map_scale=[5.12, 5.12], voxel_scale=0.04, num_voxel_per_blk_axis=16, max_ray_length=3.0, min_ray_length=0.3,
           internal_voxels=10, recast_step=2, texture_enabled=False)


def _h(x):
    return np.asarray(x, dtype=f32).astype(f16)


def _rnd(x):
    x = f32(x); r = np.trunc(x)
    return int(r + (np.copysign(f32(1), x) if abs(x - r) >= f32(0.5) else f32(0)))


# ---------------------------------------------------------------------------------------------------------------------
# fusion
# ---------------------------------------------------------------------------------------------------------------------
def _fusion_case():
    vs = f32(0.04)
    idx = np.array([[1, 2, 3], [2, 2, 3], [-5, 7, 0], [1, 3, 3]], np.int16)
    t = np.array([0.0512, -0.03, 0.1, 0.02], f16)
    w = np.array([2.5, 1.25, 4.0, 0.75], f16)
    occ = np.array([1, 0, 2, 1], np.int8)
    a = np.deg2rad(10.0)
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    T = np.array([0.013, -0.027, 0.041])
    return vs, idx, t, w, occ, R, T


def _splat_contributions(vs, idx, t, w, occ, R, T):
    """Per destination voxel, in struct-for order of the sources: (w_tsdf f32, tsdf f32, occ) -- dense_tsdf.py:291-307."""
    Rf, Tf = R.astype(f32), T.astype(f32)
    order = np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0]))                 # for s, i, j, k in TSDF (one 16^3 block here)
    out = {}
    for q in order:
        i, j, k = (int(v) for v in idx[q])
        p = [f32(f32(i) * vs), f32(f32(j) * vs), f32(f32(k) * vs)]       # ijk_to_xyz  mapping_common.py:221-223
        xyz = [f32(f32(f32(f32(Rf[a, 0] * p[0]) + f32(Rf[a, 1] * p[1])) + f32(Rf[a, 2] * p[2])) + Tf[a]) for a in range(3)]   # :229-232
        fi = [f32(x / vs) for x in xyz]                                     # :294
        lo = [int(np.floor(v)) for v in fi]                                 # :296
        for di in range(2):
            for dj in range(2):
                for dk in range(2):
                    if di + dj + dk == 0:                                   # :300
                        continue
                    c = (lo[0] + di, lo[1] + dj, lo[2] + dk)
                    wt = f32(f32(f32(f32(1) - abs(f32(f32(c[0]) - fi[0]))) * f32(f32(1) - abs(f32(f32(c[1]) - fi[1])))) * f32(f32(1) - abs(f32(f32(c[2]) - fi[2]))))   # :303
                    out.setdefault(c, []).append((f32(f32(w[q]) * wt), f32(t[q]), int(occ[q])))   # :307
    return out


def _fuse(mode):
    vs, idx, t, w, occ, R, T = _fusion_case()
    sub = OracleTSDF(**CFG, max_submap_num=4)
    sub.import_sparse(0, idx, t, w, occ)
    sub.set_active_submap(1)                                                # fuse_submaps refreshes the poses of submaps [0, active)  :286-290,:315
    g = OracleTSDF(**CFG, is_global_map=True, max_submap_num=4)
    g.set_base_pose_submap(0, R, T)
    g.fuse_submaps(sub, mode=mode)
    e = sort_export(g.export_sparse())
    got = {tuple(int(v) for v in i3): (tt, ww, oo) for i3, tt, ww, oo in zip(e["indices"], e["TSDF"], e["W_TSDF"], e["occupy"])}
    return got, _splat_contributions(vs, idx, t, w, occ, R, T)


def test_fusion_faithful_matches_the_sequential_f16_replay():
    got, exp = _fuse(FAITHFUL)
    assert set(got) == set(exp) and len(exp) >= 7 * 3
    assert max(len(v) for v in exp.values()) >= 2                          # some destination voxel is reached from two sources
    for c, lst in exp.items():
        T0 = f16(0); W0 = f16(0); o = 0
        for (wt, tt, oc) in lst:                                           # fuse_with_interploation  :272-280, f16 fields
            w_new = f32(wt + f32(W0))
            T0 = _h(f32(f32(f32(_h(f32(W0) * f32(T0))) + f32(wt * tt)) / w_new))
            W0 = _h(w_new)
            o = (o + oc + 128) % 256 - 128                                   # i8 occupancy add (:280)
        assert got[c][0] == T0.view(np.uint16) and got[c][1] == W0.view(np.uint16) and got[c][2] == o, c


def test_fusion_batched_matches_exact_fixed_point_sums():
    got, exp = _fuse(BATCHED)
    assert set(got) == set(exp)
    for c, lst in exp.items():
        num = sum(int(np.rint(f32(f32(wt * tt) * f32(16777216.0)))) for wt, tt, _ in lst)
        den = sum(int(np.rint(f32(wt * f32(16777216.0)))) for wt, _, _ in lst)
        numf, denf = f32(np.float64(num) / 16777216.0), f32(np.float64(den) / 16777216.0)
        o = (sum(oc for _, _, oc in lst) + 128) % 256 - 128
        assert got[c][0] == _h(numf / denf).view(np.uint16) and got[c][1] == _h(denf).view(np.uint16) and got[c][2] == o, c


# ---------------------------------------------------------------------------------------------------------------------
# marching cubes: one cube, one corner inside
# ---------------------------------------------------------------------------------------------------------------------
def test_marching_cubes_single_corner_by_hand():
    vs = f32(0.04)
    grid = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]      # marching_cube_mesher.py:196-206
    vals = np.array([-0.02, 0.04, 0.06, 0.03, 0.05, 0.07, 0.08, 0.065], f16)
    base = (3, -2, 5)
    idx = np.array([[base[0] + g[0], base[1] + g[1], base[2] + g[2]] for g in grid], np.int16)
    m = OracleTSDF(**CFG)
    m.import_sparse(0, idx, vals, np.ones(8, f16), np.zeros(8, np.int8))
    v, nrm, _, n = m.generate_mesh(step=1, surface_thres=0.2)
    # only the cube anchored at `base` has all 8 corners observed (:137-139); corner 0 < 0 -> cubeindex 1, Bourke: edges 0, 8, 3
    assert n == 1 and v.shape == (3, 3)
    val = {tuple(int(x) for x in i3): f32(t) for i3, t in zip(idx, vals)}

    def interp(a, b):                                                       # vertexInterp :44-60, isolevel 0
        va, vb = val[a], val[b]
        pa, pb = np.array(a, f32), np.array(b, f32)
        if abs(f32(0) - va) < f32(1e-6):
            return pa
        if abs(f32(0) - vb) < f32(1e-6):
            return pb
        mu = f32(f32(f32(0) - va) / f32(f16(vb) - f16(va)))                   # valp2 - valp1: two f16 values, an f16 operation (rounded to f16)
        return np.array([f32(pa[q] + f32(mu * f32(pb[q] - pa[q]))) for q in range(3)], f32)

    c = [tuple(int(x) for x in i3) for i3 in idx]
    verts = [interp(c[0], c[1]), interp(c[0], c[4]), interp(c[3], c[0])]     # edges 0 = (0,1), 8 = (0,4), 3 = (3,0)   :208-221
    for q in range(3):
        assert np.array_equal(v[q], (verts[q] * vs).astype(f32)), (q, v[q], verts[q] * vs)      # ijk_to_xyz :41-42

    def tsdf_h(i, j, k):                                                    # inactive cells read 0 (A7)
        return f16(val.get((i, j, k), f32(0)))

    for q in range(3):                                                      # generate_normal :84-93 (f16 field arithmetic, A4)
        p = [_rnd(x) for x in verts[q]]
        g = [_h(f32(tsdf_h(p[0] + 1, p[1], p[2])) - f32(tsdf_h(p[0] - 1, p[1], p[2]))),
             _h(f32(tsdf_h(p[0], p[1] + 1, p[2])) - f32(tsdf_h(p[0], p[1] - 1, p[2]))),
             _h(f32(tsdf_h(p[0], p[1], p[2] + 1)) - f32(tsdf_h(p[0], p[1], p[2] - 1)))]
        sq = [_h(f32(x) * f32(x)) for x in g]
        nr = _h(np.sqrt(f32(_h(f32(_h(f32(sq[0]) + f32(sq[1]))) + f32(sq[2])))))
        inv = _h(f32(1) / f32(nr))
        exp = np.array([f32(_h(f32(inv) * f32(x))) for x in g], f32)
        assert np.array_equal(nrm[q], exp), (q, nrm[q], exp)


def test_marching_cubes_vertex_on_a_corner():
    """|isolevel - val| < EPS picks the corner itself (:49-54)."""
    grid = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    vals = np.array([-0.02, 0.0, 0.06, 0.03, 0.05, 0.07, 0.08, 0.065], f16)    # corner 1 exactly on the surface
    idx = np.array(grid, np.int16)
    m = OracleTSDF(**CFG)
    m.import_sparse(0, idx, vals, np.ones(8, f16), np.zeros(8, np.int8))
    v, _, _, n = m.generate_mesh(step=1, surface_thres=0.2)
    assert n == 1 and np.array_equal(v[0], np.array([1, 0, 0], f32) * f32(0.04))


# ---------------------------------------------------------------------------------------------------------------------
# Octomap insert
# ---------------------------------------------------------------------------------------------------------------------
def test_octomap_insert_by_hand():
    vs = 0.05
    o = OracleOctomap(map_scale=[6.4, 6.4], voxel_scale=vs, K=2, texture_enabled=True, min_occupy_thres=1, max_ray_length=5.0)
    pts = np.array([[0.26, -0.126, 0.0], [0.24, -0.13, 0.024], [1.0, 1.0, 1.0], [-0.075, 0.075, -0.125], [0.2501, -0.1499, 0.0249]], f32)
    rgb = np.array([[10, 20, 30], [40, 50, 60], [70, 80, 90], [1, 2, 3], [200, 100, 50]], np.uint8)
    R = np.eye(3); T = np.array([0.0, 0.0, 0.0])
    o.integrate_points(R, T, pts, rgb)
    idx, cnt, col = o.export_leaves(with_color=True)
    exp = {}
    for p, c in zip(pts, rgb):                                              # process_point :116-124, xyz_to_ijk mapping_common.py:263-266
        key = tuple(_rnd(f32(f32(x) / f32(vs))) for x in p)
        n, _ = exp.get(key, (0, None))
        exp[key] = (n + 1, (f32(c[2]) / f32(255.0), f32(c[1]) / f32(255.0), f32(c[0]) / f32(255.0)))   # BGR -> RGB, last writer wins
    got = {tuple(int(v) for v in i3): (int(n), tuple(f32(x) for x in cc)) for i3, n, cc in zip(idx, cnt, col)}
    assert got == exp and max(n for n, _ in exp.values()) == 3              # (5,-3,0) receives three of the five points
    xyz = o.occupied_voxels(level=0)                                        # occupy > min_occupy_thres (:86-88): only the leaf with three hits
    xyz = xyz[0] if isinstance(xyz, tuple) else xyz
    assert xyz.shape[0] == 1 and np.allclose(xyz[0], np.array([5, -3, 0], f32) * f32(vs))


# ---------------------------------------------------------------------------------------------------------------------
# ESDF definition vs an independent numpy relaxation
# ---------------------------------------------------------------------------------------------------------------------
def _esdf_reference(idx, t, gamma, max_dist, vs):
    """Fixed point of d[v] = min(d[v], d[u] + |u-v| * vs) over the 26-neighbourhood restricted to one side of the surface, seeded by
    the band |TSDF| < gamma -- plain numpy sweeps until nothing changes (dense_esdf.py:228-230,282-297,313-329)."""
    lo = idx.min(0) - 1
    shp = tuple(idx.max(0) - lo + 2)
    d = np.full(shp, np.inf, f32); side = np.zeros(shp, np.int8)
    for (i, j, k), tv in zip(idx - lo, t):
        side[i, j, k] = -1 if tv < 0 else 1
        d[i, j, k] = abs(f32(tv)) if abs(f32(tv)) < f32(gamma) else f32(max_dist)
    fixed = np.isfinite(d) & (np.abs(d) < f32(gamma)) & (side != 0)
    for (i, j, k), tv in zip(idx - lo, t):
        fixed[i, j, k] = abs(f32(tv)) < f32(gamma)
    offs = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) != (0, 0, 0)]
    changed = True
    while changed:
        changed = False
        for (a, b, c) in offs:
            cost = f32(np.sqrt(f32(a * a + b * b + c * c)) * f32(vs))
            src = np.roll(d, (a, b, c), (0, 1, 2)); sside = np.roll(side, (a, b, c), (0, 1, 2))
            cand = (src + cost).astype(f32)
            ok = (side != 0) & (sside == side) & ~fixed & (cand < d)
            if ok.any():
                d[ok] = cand[ok]; changed = True
    out = np.zeros(len(t), f32)
    for q, ((i, j, k), tv) in enumerate(zip(idx - lo, t)):
        out[q] = f32(tv) if abs(f32(tv)) < f32(gamma) else f32(np.sign(f32(tv))) * d[i, j, k]
    return out


def test_esdf_matches_an_independent_relaxation():
    vs = 0.04
    r = np.arange(-9, 10, dtype=np.int16)
    ii, jj, kk = np.meshgrid(r, r, r, indexing="ij")
    idx = np.stack([ii, jj, kk], -1).reshape(-1, 3)
    keep = (np.abs(idx).sum(1) % 7) != 3                                    # some unobserved holes inside the block
    idx = idx[keep]
    p = idx.astype(f32) * f32(vs)
    t = (np.sqrt((p * p).sum(1)) - f32(0.17)).astype(f16)                   # sphere, radius 0.17 m
    m = OracleTSDF(**CFG)
    m.import_sparse(0, idx, t, np.ones(len(t), f16), np.zeros(len(t), np.int8))
    oi, oe = m.esdf(gamma=vs, max_dist=3.0)
    order = np.lexsort((oi[:, 2], oi[:, 1], oi[:, 0])); oi, oe = oi[order], oe[order]
    order = np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0])); idx, t = idx[order], t[order]
    assert np.array_equal(oi, idx)
    exp = _esdf_reference(idx.astype(np.int64), t.astype(f32), vs, 3.0, vs)
    assert np.array_equal(oe, exp)
    assert (np.sign(oe) == np.sign(t.astype(f32))).all() and np.abs(oe).max() < 3.0   # every node reached, sign kept


def test_esdf_plane_by_hand():
    """A plane half a voxel below the k = 0 layer: the band |TSDF| < voxel is the two layers k = 0 (+0.5 voxel) and k = -1 (-0.5 voxel);
    every other layer is reached straight up / down in steps of one voxel, summed in f32 from the band value (f16)."""
    vs = 0.04
    r = np.arange(-4, 5, dtype=np.int16); z = np.arange(-6, 7, dtype=np.int16)
    ii, jj, kk = np.meshgrid(r, r, z, indexing="ij")
    idx = np.stack([ii, jj, kk], -1).reshape(-1, 3)
    t = ((idx[:, 2].astype(f32) + f32(0.5)) * f32(vs)).astype(f16)
    m = OracleTSDF(**CFG)
    m.import_sparse(0, idx, t, np.ones(len(t), f16), np.zeros(len(t), np.int8))
    oi, oe = m.esdf(gamma=vs, max_dist=3.0)
    seed = f32(f16(0.5 * vs))
    acc = [seed]
    for _ in range(7):
        acc.append(f32(acc[-1] + f32(vs)))
    for (i, j, k), e in zip(oi, oe):
        k = int(k)
        assert e == (acc[k] if k >= 0 else -acc[-k - 1]), (i, j, k, e)
