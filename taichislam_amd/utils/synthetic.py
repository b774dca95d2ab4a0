"""Deterministic synthetic RGB-D streams for parity tests and bench.py (no dataset, no RNG needed).

Scene: the camera is inside a sphere of radius `radius` centred at the map
origin and moves on a circle of radius `orbit` while yawing one degree per frame.  Depth is the
analytic optical-axis depth of the sphere hit, in uint16 millimetres.
"""
import numpy as np

# RealSense D435 depth intrinsics used by the reference node (scripts/taichislam_node.py:69-72)
FX = FY = 384.2377014160156
CX = 323.4873046875
CY = 235.0628204345703
K_DEPTH = np.array([FX, 0.0, CX, 0.0, FY, CY, 0.0, 0.0, 1.0])

# optical (x right, y down, z forward) -> body (x forward, y left, z up)
R_BODY_OPTICAL = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])


def camera_pose(frame, orbit=0.5, deg_per_frame=1.0, start_deg=0.0):
    """Camera-to-world (R, T) of frame `frame` as float64, optical convention."""
    th = np.deg2rad(start_deg + deg_per_frame * frame)
    c, s = np.cos(th), np.sin(th)
    Rz = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    R = Rz @ R_BODY_OPTICAL
    T = orbit * np.array([c, s, 0.0])
    return R, T


def sphere_room_depth(R, T, h=480, w=640, radius=3.0, K=K_DEPTH, noise_mm=0, seed=1234):
    """uint16[h,w] millimetre depth image of the inside of a sphere seen from pose (R, T)."""
    fx, fy, cx, cy = K[0], K[4], K[2], K[5]
    ii, jj = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    d = np.stack([(ii - cx) / fx, (jj - cy) / fy, np.ones_like(ii)], axis=-1)
    dw = d @ R.T
    a = np.sum(dw * dw, axis=-1)
    b = 2.0 * (dw @ T)
    c = float(T @ T) - radius * radius
    t = (-b + np.sqrt(b * b - 4.0 * a * c)) / (2.0 * a)
    mm = np.rint(1000.0 * t)
    if noise_mm:
        rng = np.random.Generator(np.random.PCG64(seed))
        mm = mm + rng.integers(-noise_mm, noise_mm + 1, size=mm.shape)
    return np.clip(mm, 0, 65535).astype(np.uint16)


def sphere_room_stream(n_frames, h=480, w=640, radius=3.0, orbit=0.5, start_deg=0.0, first=0, noise_mm=0):
    """Yield (R, T, depth) for frames first..first+n_frames-1."""
    for f in range(first, first + n_frames):
        R, T = camera_pose(f, orbit=orbit, start_deg=start_deg)
        yield R, T, sphere_room_depth(R, T, h, w, radius, noise_mm=noise_mm, seed=1234 + f)


def scaled_intrinsics(h, w):
    """K for a reduced-resolution image with the same field of view as the 640x480 sensor."""
    s = w / 640.0
    return np.array([FX * s, 0.0, CX * s, 0.0, FY * s, CY * h / 480.0, 0.0, 0.0, 1.0])
