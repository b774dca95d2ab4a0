"""bench.py's parity reporting (round 5): where two sparse exports differ must be said precisely -- a single boolean over eight vectors hid for a round which one failed."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _export(n, seed=0):
    rng = np.random.default_rng(seed)
    idx = np.stack([np.arange(n), np.zeros(n, int), np.arange(n) % 7], 1).astype(np.int16)
    return {"indices": idx, "TSDF": rng.integers(0, 60000, n).astype(np.uint16), "W_TSDF": rng.integers(0, 30000, n).astype(np.uint16), "occupy": np.zeros(n, np.int8)}


def test_first_difference_reports_counts_first_voxel_and_distance():
    import bench
    a, b = _export(100), _export(100)
    b["W_TSDF"] = b["W_TSDF"].copy(); b["W_TSDF"][[40, 41, 90]] += 3
    d = bench.first_difference(b, a, ("TSDF", "W_TSDF", "occupy"), sensor_xyz=[0.0, 0.0, 0.0], voxel_scale=0.02)
    assert set(d) == {"W_TSDF"} and d["W_TSDF"]["differing_voxels"] == 3 and d["W_TSDF"]["first_index"] == [40, 0, 5]
    assert d["W_TSDF"]["got_bits"] == [int(b["W_TSDF"][40])] and d["W_TSDF"]["want_bits"] == [int(a["W_TSDF"][40])]
    lo, med, hi = d["W_TSDF"]["distance_from_sensor_m_min_median_max"]
    assert abs(lo - 0.02 * np.hypot(40, 5)) < 1e-9 and lo <= med <= hi
    assert bench.first_difference(a, a, ("TSDF", "W_TSDF", "occupy")) == {}


def test_first_difference_reports_a_missing_map():
    import bench
    a = _export(50)
    empty = {k: v[:0] for k, v in a.items()}
    assert bench.first_difference(empty, a, ("TSDF",)) == {"voxel_sets_differ": True, "voxels_got": 0, "voxels_want": 50}
