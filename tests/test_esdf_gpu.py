"""GPU ESDF vs the oracle's Dijkstra over the same graph, and vs brute-force Euclidean distance."""
import numpy as np
import pytest

from util import SMALL, lin, make_pair, small_stream

pytestmark = pytest.mark.gpu


def test_esdf_matches_oracle_and_brute_force(hip_lib):
    from oracle import BATCHED
    K, frames = small_stream(2)
    g, o = make_pair(SMALL, K)
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
        o.integrate_depth(R, T, d, mode=BATCHED)
    iters = g.update_esdf(max_dist=2.0)
    assert 1 <= iters < 400
    gi, ge = g.export_esdf()
    oi, oe = o.esdf(max_dist=2.0)
    a, b = np.argsort(lin(gi)), np.argsort(lin(oi))
    assert gi.shape[0] == oi.shape[0] > 50000 and np.array_equal(gi[a], oi[b])
    assert np.array_equal(ge[a], oe[b]), f"max diff {np.abs(ge[a] - oe[b]).max()}"
    # sanity against geometry: behind/in front of the surface the magnitude grows away from the zero band
    e = g.export_submap()
    t = e["TSDF"].astype(np.float32)[np.argsort(lin(e["indices"]))]
    esdf = ge[a]
    fixed = np.abs(t) < SMALL["voxel_scale"]
    assert np.array_equal(esdf[fixed], t[fixed])
    assert (np.sign(esdf[~fixed]) == np.sign(t[~fixed])).all() and np.abs(esdf).max() <= 2.0


def test_esdf_analytic_sphere(hip_lib):
    """Sphere SDF: the quasi-Euclidean 26-neighbour distance over-estimates the true distance by at most ~9 %."""
    from taichislam_amd.mapping import DenseTSDF
    g = DenseTSDF(map_scale=[6.4, 6.4], voxel_scale=0.05, num_voxel_per_blk_axis=16)
    g.init_sphere(voxels=60, radius=0.8)
    g.update_esdf(max_dist=3.0)
    idx, esdf = g.export_esdf()
    true = np.linalg.norm(idx.astype(np.float64) * 0.05, axis=1) - 0.8
    assert np.abs(esdf - true).max() < 0.09 * np.abs(true).max() + 0.05
    assert np.abs(esdf - true)[np.abs(true) < 0.05].max() < 1e-3
