"""SubmapMapping -- keyframe-stepped submaps over a DenseTSDF / Octomap collection, fused into one global map.

Written against this package's map classes for the orchestration the reference keeps in taichi_slam/mapping/submap_mapping.py:9-267
(the caller of the hot path in swarm mode): same constructor, method and attribute names, same call sequence into the map classes
-- tests/test_reference_callers.py drives the reference's own file and this class through one scenario and compares the traces.

    frames --recast_*_by_frame--> [PGO re-anchoring] -> submap collection (one active submap)
                                     every `keyframe_step` keyframes: export + send the finished submap, open the next, fuse -> global map
    remote submaps / trajectories --input_remote_*--> collection / pose table -> fuse -> global map

Differences from the reference, all opt-in or harmless: `autosave_path` replaces the hard-coded /home/xuhao/... save (:144-145; None =
no autosave), the send handles default to no-ops, and `recast_depth_to_map(R, T, ...)` (stale in the reference: it calls
need_create_new_submap with the wrong arity, :205-210) treats every frame as a keyframe."""
import time

from . import wire
from .dense_tsdf import DenseTSDF
from .taichi_octomap import Octomap

_COMMON = dict(voxel_scale=0.05, texture_enabled=False, min_ray_length=0.3, max_ray_length=3.0, max_disp_particles=1024 * 1024)
# per map type: (extra keywords of the submap collection, extra keywords of the global map)   submap_mapping.py:13-34, :60-82
_TYPE_OPTS = {
    "tsdf": (dict(num_voxel_per_blk_axis=10, max_submap_num=1000), dict(num_voxel_per_blk_axis=10, max_submap_num=1024)),
    "octo": (dict(K=2, max_submap_num=1000), dict(K=2, max_submap_num=1000)),
}


class _Anchor:
    """Ego-motion -> pose-graph frame.  Poses from the odometry drift; whenever the pose graph publishes an optimised pose for a frame
    the odometry has also seen, later frames are re-expressed relative to that frame (submap_mapping.py:163-170, :106-111)."""

    def __init__(self):
        self.ego, self.pgo, self.last = {}, {}, None

    def note_optimised(self, frame_id):
        if frame_id in self.ego and (self.last is None or frame_id > self.last):
            self.last = frame_id

    def __call__(self, frame_id, R, T):
        self.ego[frame_id] = (R, T)
        if self.last is None:
            return R, T
        Re, Te = self.ego[self.last]
        Rp, Tp = self.pgo[self.last]
        A = Rp @ Re.T
        return A @ R, A @ (T - Te) + Tp


class SubmapMapping:
    def __init__(self, submap_type=DenseTSDF, keyframe_step=20, sub_opts={}, global_opts={}, autosave_path=None):
        self.submap_type = submap_type
        self._kind = "octo" if (isinstance(submap_type, type) and issubclass(submap_type, Octomap)) else "tsdf"
        self.keyframe_step = keyframe_step
        self.autosave_path = autosave_path
        self.sub_opts = dict(_COMMON, map_scale=[10, 10], **_TYPE_OPTS[self._kind][0])
        self.sub_opts.update(sub_opts)
        self.submaps = {}                         # frame id of a submap's first frame -> submap id in the collection / global pose table
        self.frame_count = 0
        self.first_init = True
        self.active_submap_frame_id = 0
        self.exporting_global = False
        self.export_TSDF_xyz = self.export_color = self.export_x = None
        self.post_local_to_global_callback = None
        self.map_send_handle = lambda buf: None
        self.traj_send_handle = lambda buf: None
        self._anchor = _Anchor()
        self.submap_collection = submap_type(**self.sub_opts)
        self.global_map = self.create_globalmap(global_opts)
        self.enable_texture = self.global_map.enable_texture
        self.set_exporting_global()

    # the reference exposes these three dicts / the frame id as plain attributes (scripts/taichislam_node.py reads them)
    ego_motion_poses = property(lambda self: self._anchor.ego)
    pgo_poses = property(lambda self: self._anchor.pgo)
    last_frame_id = property(lambda self: self._anchor.last)

    def create_globalmap(self, global_opts={}):
        opts = dict(_COMMON, map_scale=[100, 100], is_global_map=True, **_TYPE_OPTS[self._kind][1])
        opts.update(global_opts)
        return self.submap_type(**opts)

    # ---- camera / export selection ---------------------------------------------------------------------------------------------------
    def set_dep_camera_intrinsic(self, K):
        self.submap_collection.set_dep_camera_intrinsic(K)

    def set_color_camera_intrinsic(self, K):
        self.submap_collection.set_color_camera_intrinsic(K)

    def set_export_submap(self, new_submap):
        self.export_color = new_submap.export_color
        if self._kind == "tsdf":
            self.export_TSDF_xyz, self.num_TSDF_particles = new_submap.export_TSDF_xyz, new_submap.num_TSDF_particles
        else:
            self.export_x, self.num_export_particles = new_submap.export_x, new_submap.num_export_particles

    def set_exporting_global(self):
        self.exporting_global = True
        self.set_export_submap(self.global_map)

    def set_exporting_local(self):
        self.exporting_global = False
        self.set_export_submap(self.submap_collection)

    # ---- pose graph --------------------------------------------------------------------------------------------------------------------
    def set_frame_poses(self, frame_poses, from_remote=False):
        """Optimised poses {frame_id: (R, T)}: re-anchor the odometry and move the base poses of the submaps that start on these frames."""
        self._anchor.pgo.update(frame_poses)
        moved = {}
        for fid, (R, T) in ((f, (p[0], p[1])) for f, p in frame_poses.items()):
            self._anchor.note_optimised(fid)
            sid = self.submaps.get(fid)
            if sid is not None:
                self.global_map.set_base_pose_submap(sid, R, T)
                moved[fid] = frame_poses[fid]
        if not from_remote:                      # our own optimisation: tell the other agents which submaps moved
            self.send_traj(moved)

    def convert_by_pgo(self, frame_id, R, T):
        return self._anchor(frame_id, R, T)

    # ---- submap life cycle -----------------------------------------------------------------------------------------------------------------
    def need_create_new_submap(self, is_keyframe, R=None, T=None):
        return self.frame_count == 0 or (bool(is_keyframe) and self.frame_count % self.keyframe_step == 0)

    def create_new_submap(self, frame_id, R, T):
        col = self.submap_collection
        if self.first_init:
            self.first_init = False
        else:                                    # close the active submap: ship it, open the next slot, refresh the global map
            self.send_submap(col.export_submap())
            col.switch_to_next_submap()
            col.clear_last_TSDF_exporting = True
            self.local_to_global()
        sid = col.get_active_submap_id()
        for m in (self.global_map, col):
            m.set_base_pose_submap(sid, R, T)
        self.submaps[frame_id] = sid
        self._anchor.pgo[frame_id] = (R, T)
        self.active_submap_frame_id = frame_id
        print(f"[SubmapMapping] frame {frame_id} opens submap {sid} ({sid + 1} local submaps)")
        if self.autosave_path and sid % 2 == 0:
            self.saveMap(self.autosave_path)
        return col

    def local_to_global(self):
        self.global_map.fuse_submaps(self.submap_collection)
        if self.post_local_to_global_callback is not None:
            self.post_local_to_global_callback(self.global_map)

    def _frame(self, frame_id, is_keyframe, pose, ext):
        """Common part of the two by-frame entry points: anchored body pose, submap roll-over, camera pose."""
        R, T = self._anchor(frame_id, pose[0], pose[1])
        if self.need_create_new_submap(is_keyframe, R, T):
            self.create_new_submap(frame_id, R, T)
        return R @ ext[0], T + R @ ext[1]

    def recast_depth_to_map_by_frame(self, frame_id, is_keyframe, pose, ext, depthmap, texture):
        Rc, Tc = self._frame(frame_id, is_keyframe, pose, ext)
        self.submap_collection.recast_depth_to_map(Rc, Tc, depthmap, texture)
        self.frame_count += 1

    def recast_pcl_to_map_by_frame(self, frame_id, is_keyframe, pose, ext, pcl, rgb_array):
        Rc, Tc = self._frame(frame_id, is_keyframe, pose, ext)
        self.submap_collection.recast_pcl_to_map(Rc, Tc, pcl, rgb_array)
        self.frame_count += 1

    def recast_depth_to_map(self, R, T, depthmap, texture):
        if self.need_create_new_submap(True, R, T):
            self.create_new_submap(self.frame_count, R, T)          # no frame ids from the caller: the frame counter stands in
        self.submap_collection.recast_depth_to_map(R, T, depthmap, texture)
        self.frame_count += 1

    # ---- visualisation exports ---------------------------------------------------------------------------------------------------------------
    def _shown(self):
        return self.global_map if self.exporting_global else self.submap_collection

    def cvt_TSDF_to_voxels_slice(self, z):
        self._shown().cvt_TSDF_to_voxels_slice(z)

    def cvt_TSDF_surface_to_voxels(self):
        if not self.submaps:
            return
        self._shown().cvt_TSDF_surface_to_voxels()
        if self.exporting_global:                # the active submap is not fused yet: append it to the global map's particles
            g = self.global_map
            self.submap_collection.cvt_TSDF_surface_to_voxels_to(g.num_TSDF_particles, g.max_disp_particles, self.export_TSDF_xyz, self.export_color)

    def cvt_occupy_to_voxels(self, level):
        self._shown().cvt_occupy_to_voxels(level)
        if self.exporting_global:
            g = self.global_map
            self.submap_collection.cvt_occupy_voxels_to(level, g.num_export_particles, g.max_disp_particles, self.export_x, self.export_color)

    # ---- exchange with the other agents ----------------------------------------------------------------------------------------------------------
    def send_submap(self, submap):
        submap["frame_id"] = self.active_submap_frame_id
        submap["pose"] = self._anchor.pgo[self.active_submap_frame_id]
        t0 = time.time()
        buf, raw = wire.pack(submap)
        self.map_send_handle(buf)
        print(f"[SubmapMapping] submap sent: {raw / 1024.0:.1f} kB -> {len(buf) / 1024:.1f} kB on the wire, packed in {(time.time() - t0) * 1000:.1f} ms")

    def send_traj(self, traj):
        t0 = time.time()
        buf, raw = wire.pack(traj)
        self.traj_send_handle(buf)
        print(f"[SubmapMapping] trajectory sent: {len(traj)} poses, {len(buf) / 1024:.1f} kB on the wire ({(time.time() - t0) * 1000:.1f} ms)")

    def input_remote_submap(self, buf):
        submap = wire.unpack(buf)
        print(f"[SubmapMapping] remote submap of frame {submap['frame_id']}: {len(buf) / 1024:.1f} kB on the wire, {len(submap['TSDF']) if 'TSDF' in submap else 0} voxels")
        sid = self.submap_collection.input_remote_submap(submap)
        self.global_map.set_base_pose_submap(sid, submap["pose"][0], submap["pose"][1])
        self.local_to_global()
        self.submaps[submap["frame_id"]] = sid

    def input_remote_traj(self, buf):
        traj = wire.unpack(buf)
        self.set_frame_poses(traj, True)
        print(f"[SubmapMapping] remote trajectory: {len(traj)} poses ({len(buf) / 1024.0:.1f} kB)")

    def saveMap(self, filename):
        self.global_map.saveMap(filename)

    def export_submap(self):
        return self.submap_collection.export_submap()
