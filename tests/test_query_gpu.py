"""GPU parity of the batched map queries (mapping_common.py:165-204, dense_tsdf.py:148-155)."""
import numpy as np
import pytest

from util import SMALL, make_pair, small_stream

pytestmark = pytest.mark.gpu


def test_queries_match_oracle(hip_lib):
    from oracle import BATCHED
    K, frames = small_stream(3)
    g, o = make_pair(SMALL, K)
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None); o.integrate_depth(R, T, d, mode=BATCHED)
    rng = np.random.default_rng(5)
    pts = rng.uniform(-4.5, 4.5, size=(20000, 3)).astype(np.float32)
    pts[:100] *= 3.0                                   # some outside the volume
    for mode, fn in ((0, g.is_pos_occupy), (1, g.is_pos_unobserved)):
        assert np.array_equal(fn(pts), o.query_points(mode, pts))
    assert np.array_equal(g.is_near_pos_occupy(pts[:4000], 2), o.query_points(2, pts[:4000], 2))
    assert 0.05 < g.is_pos_unobserved(pts).mean() < 0.999
    origin = np.tile(frames[0][1].astype(np.float32), (4096, 1))
    d = rng.normal(size=(4096, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    gh, ge, gl = g.raycast(origin, d, 4.0)
    oh, oe, ol = o.raycast(origin, d, 4.0)
    assert np.array_equal(gh, oh) and np.array_equal(ge, oe) and np.array_equal(gl, ol)
    assert gh.any()
