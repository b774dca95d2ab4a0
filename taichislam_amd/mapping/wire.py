"""Wire format of submaps and trajectories exchanged between agents.

Compatible with the reference's LCM payloads (taichi_slam/mapping/submap_mapping.py:226-265): `numpy.save` of a dict (pickled object
array), zlib level 1.  A submap dict is what DenseTSDF.export_submap() returns (dense_tsdf.py:456-476) plus `frame_id` and `pose`;
a trajectory is {frame_id: (R, T)}."""
import io
import zlib

import numpy as np


def pack(obj, level=1):
    """-> (compressed bytes, raw size)"""
    raw = io.BytesIO()
    np.save(raw, obj)
    view = raw.getbuffer()
    return zlib.compress(view, level=level), len(view)


def unpack(buf):
    return np.load(io.BytesIO(zlib.decompress(buf)), allow_pickle=True).item()
