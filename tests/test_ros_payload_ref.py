"""The PointCloud2 payload (taichislam_amd.utils.ros_adapters, csrc k_pack_pointcloud2) against the reference's own helper,
taichi_slam/utils/ros_pcl_transfer.py:point_cloud -- imported unmodified where the reference tree exists, with the ROS message classes it needs
(rospy / sensor_msgs / std_msgs / geometry_msgs / ros_numpy are not installable here) replaced by plain records."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

REF = "/root/reference/taichi_slam/utils/ros_pcl_transfer.py"


class _Rec:
    def __init__(self, *a, **kw):
        self.__dict__.update(kw)


def _load_reference_helper():
    mods = {}
    for name in ("ros_numpy", "rospy", "sensor_msgs", "sensor_msgs.msg", "geometry_msgs", "geometry_msgs.msg", "std_msgs", "std_msgs.msg"):
        mods[name] = types.ModuleType(name)
    mods["sensor_msgs.msg"].PointField = type("PointField", (_Rec,), {"FLOAT32": 7})
    mods["sensor_msgs.msg"].PointCloud2 = type("PointCloud2", (_Rec,), {})
    mods["sensor_msgs.msg"].PointCloud = type("PointCloud", (_Rec,), {})
    mods["geometry_msgs.msg"].Point32 = type("Point32", (_Rec,), {})
    mods["geometry_msgs.msg"].PoseStamped = type("PoseStamped", (_Rec,), {})
    mods["std_msgs.msg"].Header = type("Header", (_Rec,), {})
    mods["rospy"].Time = type("Time", (), {"now": staticmethod(lambda: 0)})
    mods["sensor_msgs"].msg, mods["geometry_msgs"].msg, mods["std_msgs"].msg = mods["sensor_msgs.msg"], mods["geometry_msgs.msg"], mods["std_msgs.msg"]
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        spec = importlib.util.spec_from_file_location("ref_ros_pcl_transfer", REF)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference tree (dev box)")
@pytest.mark.parametrize("has_rgb", [False, True])
def test_pointcloud2_payload_equals_the_reference_helper(has_rgb):
    from taichislam_amd.utils import ros_adapters
    ref = _load_reference_helper()
    rng = np.random.default_rng(1)
    xyz = rng.normal(size=(257, 3)).astype(np.float32)
    rgb = rng.uniform(size=(257, 3)).astype(np.float32)
    # scripts/taichislam_node.py:420-425 (pub_to_ros): xyz | colours as float, or xyz alone
    pts = np.concatenate((xyz, rgb.astype(float)), axis=1) if has_rgb else xyz
    msg = ref.point_cloud(pts, "world", has_rgb=has_rgb)
    mine = ros_adapters.pointcloud2_payload(np.concatenate([xyz, rgb], axis=1) if has_rgb else xyz, has_rgb)
    assert (msg.height, msg.width, msg.is_dense, msg.is_bigendian, msg.point_step, msg.row_step) == \
        (mine["height"], mine["width"], mine["is_dense"], mine["is_bigendian"], mine["point_step"], mine["row_step"])
    assert [(f.name, f.offset, f.datatype, f.count) for f in msg.fields] == [(f["name"], f["offset"], f["datatype"], f["count"]) for f in mine["fields"]]
    assert bytes(msg.data) == mine["data"]


@pytest.mark.skipif(not os.path.exists("/root/reference/scripts/taichislam_node.py"), reason="needs the reference tree (dev box)")
def test_every_name_the_reference_node_uses_on_the_maps_exists_in_the_shims():
    """scripts/taichislam_node.py drives the map, the mesher and the submap orchestration through these attributes / methods; a drop-in must have them all."""
    import re
    src = open("/root/reference/scripts/taichislam_node.py").read()
    used = set(re.findall(r"(?:self\.mapping|mapping|self\.mesher|mesher)\.([A-Za-z_][A-Za-z0-9_]*)", src))
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "taichislam_amd", "mapping")
    mine = "".join(open(os.path.join(root, f)).read() for f in os.listdir(root) if f.endswith(".py"))
    missing = [n for n in sorted(used) if not re.search(r"\b" + re.escape(n) + r"\b", mine)]
    assert len(used) >= 20 and not missing, missing
