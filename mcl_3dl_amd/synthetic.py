"""Seeded synthetic maps / scans / particle sets for the BASELINE.json configs.

Everything is generated with ``numpy.random.default_rng(seed)`` (seed 12345 unless stated) so the CPU
oracle, the golden fixtures and the GPU run see bit-identical float32 inputs.

Scene family (after the reference's own test scenes: hollow box ``test/src/test_expansion_resetting.cpp:78-91``,
sensor noise N(0, 0.01) ``test/src/test_global_localization.cpp:63``):
  * map  : hollow axis-aligned cube, 6 faces, n x n points per face on a ``spacing`` lattice whose in-plane
           coordinates are ``(i + 0.5) * spacing - n * spacing / 2``; all labels 0 unless ``label_wall`` is set.
  * pose : 3 m from two walls, 1.5 m above the floor, yaw 0.3 rad.
  * scan : map points seen from the true pose (robot frame) that pass the reference's own clip filter
           (``src/lidar_measurement_model_likelihood.cpp:79-103``: clip_near <= xy-range <= clip_far,
           clip_z_min <= z <= clip_z_max), N drawn without replacement, + N(0, 0.01^2) per axis.
  * particles: true pose + N(0; sigma_xyz=(0.2,0.2,0.05) m, sigma_rpy=(0.02,0.02,0.1) rad), weights 1/N.
"""
from dataclasses import dataclass, field

import numpy as np

CONFIGS = {
    # name: (cube n, N_p, N_s, N_b)
    "C1": dict(n=91, n_p=64, n_s=1000, n_b=0),
    "C2": dict(n=408, n_p=4096, n_s=16384, n_b=0),
    "C3": dict(n=408, n_p=4096, n_s=16384, n_b=512),
    "C4": dict(n=408, n_p=262144, n_s=16384, n_b=0),
    "C5": dict(n=1291, n_p=65536, n_s=65536, n_b=2048),
}


def cube_map(n, spacing=0.1, dtype=np.float32):
    """Hollow cube map: 6 * n * n points. Face planes at +-n*spacing/2."""
    half = n * spacing / 2.0
    c = (np.arange(n, dtype=np.float64) + 0.5) * spacing - half
    u, v = np.meshgrid(c, c, indexing="ij")
    u = u.ravel()
    v = v.ravel()
    lo = np.full_like(u, -half)
    hi = np.full_like(u, half)
    faces = [
        np.stack([u, v, lo], 1),  # floor
        np.stack([u, v, hi], 1),  # ceiling
        np.stack([lo, u, v], 1),
        np.stack([hi, u, v], 1),
        np.stack([u, lo, v], 1),
        np.stack([u, hi, v], 1),
    ]
    return np.ascontiguousarray(np.concatenate(faces, 0).astype(dtype))


def quat_from_rpy(rpy):
    """Quat::setRPY (include/mcl_3dl/quat.h:204-217), float64 in, (x,y,z,w) out."""
    rpy = np.asarray(rpy, dtype=np.float64)
    r, p, y = rpy[..., 0], rpy[..., 1], rpy[..., 2]
    t2, t3 = np.cos(r / 2), np.sin(r / 2)
    t4, t5 = np.cos(p / 2), np.sin(p / 2)
    t0, t1 = np.cos(y / 2), np.sin(y / 2)
    return np.stack([
        t0 * t3 * t4 - t1 * t2 * t5,
        t0 * t2 * t5 + t1 * t3 * t4,
        t1 * t2 * t4 - t0 * t3 * t5,
        t0 * t2 * t4 + t1 * t3 * t5,
    ], -1)


def quat_to_matrix(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


@dataclass
class Scene:
    map_xyz: np.ndarray            # (N_m, 3) float32
    map_label: np.ndarray          # (N_m,) uint32
    true_pose: np.ndarray          # (7,) float32: px,py,pz,qx,qy,qz,qw
    poses: np.ndarray              # (N_p, 7) float32
    weights: np.ndarray            # (N_p,) float32
    scan_lik: np.ndarray           # (N_s, 3) float32, robot frame
    scan_beam: np.ndarray          # (N_b, 3) float32, robot frame
    scan_beam_label: np.ndarray    # (N_b,) uint32 (= origin index)
    origins: np.ndarray            # (N_o, 3) float32, robot frame
    odom_err: np.ndarray           # (N_p, 3) float32 (State6DOF::odom_err_integ_lin_)
    meta: dict = field(default_factory=dict)


def _visible(map_xyz, pos, rot_m, clip_near, clip_far, z_min, z_max, need, rng):
    """Map points in the robot frame that pass the reference's clip filter; grows clip_far / z window
    until at least ``need`` points qualify."""
    local = (map_xyz.astype(np.float64) - pos) @ rot_m  # R^T (p - t)
    far = clip_far
    zlo, zhi = z_min, z_max
    for _ in range(32):
        r2 = local[:, 0] ** 2 + local[:, 1] ** 2
        ok = (r2 <= far * far) & (r2 >= clip_near * clip_near) & (local[:, 2] >= zlo) & (local[:, 2] <= zhi)
        idx = np.nonzero(ok)[0]
        if len(idx) >= need:
            return local, idx
        far *= 1.25
        zlo *= 1.25
        zhi *= 1.25
    raise RuntimeError("not enough visible map points for the requested scan size")


def make_scene(n=91, n_p=64, n_s=1000, n_b=0, seed=12345, spacing=0.1, label_wall=None,
               sigma_xyz=(0.2, 0.2, 0.05), sigma_rpy=(0.02, 0.02, 0.1), scan_noise=0.01,
               lik_clip=(0.5, 10.0, -2.0, 2.0), beam_clip=(0.5, 4.0, -2.0, 2.0), global_lattice=False, map_jitter=0.0):
    rng = np.random.default_rng(seed)
    map_xyz = cube_map(n, spacing)
    if map_jitter:
        # a voxel-filtered real map: one CENTROID per occupied voxel, i.e. lattice points displaced inside their voxel
        # (the BASELINE maps are exact lattices; this is the stress case for the candidate-voxel index)
        jit = np.random.default_rng(seed + 777).uniform(-map_jitter, map_jitter, map_xyz.shape)
        map_xyz = (map_xyz + jit).astype(np.float32)
    map_label = np.zeros(len(map_xyz), np.uint32)
    if label_wall is not None:
        # give one wall (x = -half) a semantic label, after test/src/test_beam_label.cpp:69-85
        nn = n * n
        map_label[2 * nn:3 * nn] = label_wall
    half = n * spacing / 2.0
    pos = np.array([-half + 3.0, -half + 3.0, -half + 1.5])
    q_true = quat_from_rpy([0.0, 0.0, 0.3])
    rot_m = quat_to_matrix(q_true)
    true_pose = np.concatenate([pos, q_true]).astype(np.float32)

    local, idx = _visible(map_xyz, pos, rot_m, *lik_clip, need=max(n_s, 1), rng=rng)
    pick = rng.choice(idx, size=n_s, replace=False) if n_s else np.zeros(0, np.int64)
    scan_lik = (local[pick] + rng.normal(0.0, scan_noise, (n_s, 3))).astype(np.float32)

    if n_b:
        localb, idxb = _visible(map_xyz, pos, rot_m, *beam_clip, need=n_b, rng=rng)
        pickb = rng.choice(idxb, size=n_b, replace=False)
        scan_beam = (localb[pickb] + rng.normal(0.0, scan_noise, (n_b, 3))).astype(np.float32)
    else:
        scan_beam = np.zeros((0, 3), np.float32)
    scan_beam_label = np.zeros(len(scan_beam), np.uint32)
    origins = np.array([[0.0, 0.0, 0.5]], np.float32)

    if global_lattice:
        # global-localisation style hypotheses (src/mcl_3dl.cpp:1076-1095): floor lattice x 12 yaws
        div_yaw = 12
        n_pos = (n_p + div_yaw - 1) // div_yaw
        side = int(np.ceil(np.sqrt(n_pos)))
        g = (np.arange(side) + 0.5) * (2 * half - 1.0) / side - half + 0.5
        gx, gy = np.meshgrid(g, g, indexing="ij")
        pts = np.stack([gx.ravel(), gy.ravel(), np.full(side * side, -half + 1.5)], 1)[:n_pos]
        ppos = np.repeat(pts, div_yaw, 0)[:n_p]
        yaw = np.tile(2.0 * np.pi * np.arange(div_yaw) / div_yaw, n_pos)[:n_p]
        quat = quat_from_rpy(np.stack([np.zeros(n_p), np.zeros(n_p), yaw], 1))
    else:
        ppos = pos + rng.normal(0.0, 1.0, (n_p, 3)) * np.asarray(sigma_xyz)
        rpy = np.array([0.0, 0.0, 0.3]) + rng.normal(0.0, 1.0, (n_p, 3)) * np.asarray(sigma_rpy)
        quat = quat_from_rpy(rpy)
    poses = np.concatenate([ppos, quat], 1).astype(np.float32)
    weights = np.full(n_p, 1.0 / max(n_p, 1), np.float32)
    odom_err = (rng.normal(0.0, 0.05, (n_p, 3))).astype(np.float32)
    return Scene(map_xyz, map_label, true_pose, poses, weights, scan_lik, scan_beam, scan_beam_label, origins,
                 odom_err, meta=dict(n=n, n_p=n_p, n_s=n_s, n_b=n_b, seed=seed, spacing=spacing))


def make_config(name, **override):
    cfg = dict(CONFIGS[name])
    cfg.update(override)
    if name == "C4" and "global_lattice" not in override:
        cfg["global_lattice"] = True
    return make_scene(**cfg)
