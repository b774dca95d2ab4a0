"""ROS-side glue without a ROS dependency: the payload of the messages the reference's node publishes.

The reference builds a sensor_msgs/PointCloud2 from numpy copies of export_TSDF_xyz / export_color (taichi_slam/utils/ros_pcl_transfer.py:96-136,
scripts/taichislam_node.py:420-425): FLOAT32 fields x y z [r g b] at offsets 0, 4, ..., point_step 12 or 24, height 1, little endian,
is_dense False.  DenseTSDF.pointcloud2() interleaves the rows on the device and hands back exactly that layout; `to_ros` wraps it in the
message types when rospy / sensor_msgs are importable (they are not in this image)."""
import numpy as np

FLOAT32 = 7            # sensor_msgs/PointField.FLOAT32


def pointcloud2_payload(rows, has_rgb):
    """rows: float32 [n, 3] or [n, 6] (already interleaved).  Returns the fields of the PointCloud2 message as a dict."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    names = "xyzrgb" if has_rgb else "xyz"
    assert rows.ndim == 2 and rows.shape[1] == len(names)
    step = 4 * len(names)
    return {"height": 1, "width": int(rows.shape[0]), "is_dense": False, "is_bigendian": False,
            "fields": [{"name": c, "offset": 4 * i, "datatype": FLOAT32, "count": 1} for i, c in enumerate(names)],
            "point_step": step, "row_step": step * int(rows.shape[0]), "data": rows.tobytes()}


def to_ros(payload, frame_id="world", stamp=None):
    """sensor_msgs/PointCloud2 from a payload dict (needs rospy + sensor_msgs)."""
    import rospy
    import sensor_msgs.msg as sensor_msgs
    import std_msgs.msg as std_msgs
    fields = [sensor_msgs.PointField(name=f["name"], offset=f["offset"], datatype=f["datatype"], count=f["count"]) for f in payload["fields"]]
    header = std_msgs.Header(frame_id=frame_id, stamp=stamp if stamp is not None else rospy.Time.now())
    return sensor_msgs.PointCloud2(header=header, height=payload["height"], width=payload["width"], is_dense=payload["is_dense"],
                                   is_bigendian=payload["is_bigendian"], fields=fields, point_step=payload["point_step"],
                                   row_step=payload["row_step"], data=payload["data"])
