"""Recording / replay of the calls an orchestration (SubmapMapping) makes into the map classes -- shared by tools/gen_submap_trace.py (dev box:
drives the REFERENCE's own taichi_slam/mapping/submap_mapping.py, loaded by path, and commits the trace as tests/golden/submap_trace.json)
and tests/test_submap_trace.py (any box: replays the committed trace on the HIP shims / compares the package's own class against it).

A trace is a JSON list of calls {"map": tag, "method": name, "args": [...], "ret": value} in the order the orchestration made them.
Arrays are stored by value (dtype, shape, zlib + base64 of the bytes) -- the depth images of the scenario are 120x160 uint16 --, map
objects and their field-likes by tag."""
import base64
import io
import zlib

import numpy as np

H, W, NFRAMES, KEYFRAME_STEP = 120, 160, 8, 3
OPTS = dict(map_scale=[10.24, 10.24], voxel_scale=0.04, num_voxel_per_blk_axis=16, max_ray_length=5.0, max_submap_num=16)


def enc(x):
    if isinstance(x, np.ndarray):
        return {"__nd__": str(x.dtype), "shape": list(x.shape), "z": base64.b64encode(zlib.compress(np.ascontiguousarray(x).tobytes(), 6)).decode()}
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    if isinstance(x, (list, tuple)):
        return [enc(v) for v in x]
    if isinstance(x, dict):
        return {"__dict__": [[enc(k), enc(v)] for k, v in x.items()]}
    if isinstance(x, RecField):
        return {"__field__": x.name, "owner": x.owner.tag}
    if isinstance(x, RecMap):
        return {"__map__": x.tag}
    if x is None or isinstance(x, (bool, int, float, str)):
        return x
    raise TypeError(f"cannot record {type(x)}")


def dec(x, maps=None):
    if isinstance(x, list):
        return [dec(v, maps) for v in x]
    if isinstance(x, dict):
        if "__nd__" in x:
            return np.frombuffer(zlib.decompress(base64.b64decode(x["z"])), dtype=np.dtype(x["__nd__"])).reshape(x["shape"]).copy()
        if "__dict__" in x:
            return {dec(k, maps): dec(v, maps) for k, v in x["__dict__"]}
        if "__map__" in x:
            return maps[x["__map__"]]
        if "__field__" in x:
            return getattr(maps[x["owner"]], x["__field__"])
    return x


class RecField:
    def __init__(self, owner, name):
        self.owner, self.name, self.v = owner, name, 0
        self._owner = owner                      # (what the shims' DeviceArrayField exposes: cvt_*_to looks the destination map up through it)

    def __getitem__(self, k):
        return self.v

    def __setitem__(self, k, v):
        self.v = v


class RecBase:
    pass


class RecMap(RecBase):
    """Stand-in for DenseTSDF / Octomap: every call is appended to RecMap.TRACE; the handful of calls whose results steer the orchestration
    (active submap id, export_submap, input_remote_submap) answer like the real classes."""
    TRACE = None
    COUNT = 0

    def __init__(self, **kw):
        RecMap.COUNT += 1
        object.__setattr__(self, "tag", ("global" if kw.get("is_global_map") else "collection") + f"#{RecMap.COUNT}")
        self.enable_texture = kw.get("texture_enabled", False)
        self.max_disp_particles = kw.get("max_disp_particles", 0)
        self.max_submap_num = kw.get("max_submap_num", 0)
        self.active, self.remote = 0, 0
        for f in ("export_color", "export_TSDF_xyz", "num_TSDF_particles", "export_x", "num_export_particles"):
            object.__setattr__(self, f, RecField(self, f))
        RecMap.TRACE.append({"map": self.tag, "method": "__init__", "args": [enc(kw)], "ret": None})

    def _rec(self, name, args, ret=None):
        RecMap.TRACE.append({"map": self.tag, "method": name, "args": [enc(a) for a in args], "ret": enc(ret)})
        return ret

    def __setattr__(self, k, v):
        if k == "clear_last_TSDF_exporting":
            self._rec("setattr clear_last_TSDF_exporting", (v,))
        object.__setattr__(self, k, v)

    def get_active_submap_id(self):
        return self._rec("get_active_submap_id", (), self.active)

    def switch_to_next_submap(self):
        self.active += 1
        return self._rec("switch_to_next_submap", (), self.active)

    def export_submap(self):
        self._rec("export_submap", ())
        n = 5 + self.active
        return {"indices": np.arange(3 * n, dtype=np.int16).reshape(n, 3), "TSDF": np.linspace(0, 1, n).astype(np.float16),
                "W_TSDF": np.ones(n, np.float16), "color": np.array([]), "occupy": np.zeros(n, np.int8), "map_scale": [10.0, 10.0],
                "voxel_scale": 0.05, "texture_enabled": False, "num_voxel_per_blk_axis": 10}

    def input_remote_submap(self, submap):
        self.remote += 1
        return self._rec("input_remote_submap", (submap,), self.max_submap_num - self.remote)

    def __getattr__(self, name):                      # every other method: record and return None
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a: self._rec(name, a)


class RecTSDF(RecMap):
    pass


class RecOcto(RecMap):
    pass


def frames():
    from taichislam_amd.utils import synthetic as syn
    K = syn.scaled_intrinsics(H, W)
    out = []
    for f in range(NFRAMES):
        R, T = syn.camera_pose(f)
        out.append((R, T, syn.sphere_room_depth(R, T, H, W, K=K)))
    return K, out


def drive(SM, map_cls):
    """The scenario: eight depth frames, a new submap every three keyframes (two finished submaps exported and sent), one pose-graph
    update that moves the second submap, then local_to_global.  Returns (SubmapMapping instance, what went on the wire)."""
    K, fr = frames()
    sm = SM(map_cls, keyframe_step=KEYFRAME_STEP, sub_opts=dict(OPTS), global_opts=dict(OPTS))
    sent = []
    sm.map_send_handle = sent.append
    sm.traj_send_handle = lambda b: None
    if hasattr(sm, "autosave_path"):
        sm.autosave_path = None
    else:                                             # the reference saves to a hard-coded path (submap_mapping.py:144-145)
        sm.saveMap = lambda filename: None
    sm.set_dep_camera_intrinsic(K)
    ext = (np.eye(3), np.zeros(3))
    for f, (R, T, d) in enumerate(fr):
        if f == 5:                                    # the pose graph nudges the frame that opened submap 1
            a = 0.01
            dR = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
            sm.set_frame_poses({3: (dR @ fr[3][0], fr[3][1] + np.array([0.01, -0.02, 0.0]))})
        sm.recast_depth_to_map_by_frame(f, True, (R, T), ext, d, np.array([], dtype=int))
    sm.local_to_global()
    return sm, sent


def record(SM):
    RecMap.TRACE, RecMap.COUNT = [], 0
    sm, sent = drive(SM, RecTSDF)
    decode = lambda b: np.load(io.BytesIO(zlib.decompress(b)), allow_pickle=True).item()
    return {"trace": RecMap.TRACE, "sent": [enc(decode(b)) for b in sent], "submaps": {str(k): int(v) for k, v in sm.submaps.items()}}


def same(a, b):
    """deep equality of decoded values (arrays by dtype / shape / bytes)"""
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()
    if isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
    return a == b
