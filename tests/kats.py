"""Known-answer vectors re-hosted from the reference's own unit tests (paths relative to at-wat/mcl_3dl v0.7.0).
Shared by the CPU oracle tests and the GPU tests so both are pinned to the same upstream goldens."""
import numpy as np

# ---- test/src/test_raycast_dda.cpp:185-244 (Waypoints) ---------------------------------------------------------
WAYPOINT_MAP = np.array([[0.9, 0.6, 0.0], [-2.05, -2.05, -2.05], [2.05, 2.05, 2.05]], np.float32)
WAYPOINT_CASTER = dict(map_grid=(0.1, 0.1, 0.1), dda_grid_size=0.1, ray_angle_half=0.5, hit_tolerance=0.0)
WAYPOINT_CASES = [
    dict(name="Waypoints#1", begin=(0.0, 0.0, 0.0), end=(1.2, 0.8, 0.0), collision=True, expected=[
        (0.1, 0.0, 0.0), (0.1, 0.1, 0.0), (0.2, 0.1, 0.0), (0.2, 0.2, 0.0), (0.3, 0.2, 0.0), (0.4, 0.2, 0.0),
        (0.4, 0.3, 0.0), (0.5, 0.3, 0.0), (0.5, 0.4, 0.0), (0.6, 0.4, 0.0), (0.7, 0.4, 0.0), (0.7, 0.5, 0.0),
        (0.8, 0.5, 0.0), (0.8, 0.6, 0.0), (0.9, 0.6, 0.0)]),
    dict(name="Waypoints#2", begin=(-0.04, 0.04, 0.0), end=(1.16, 0.84, 0.0), collision=True, expected=[
        (0.0, 0.1, 0.0), (0.1, 0.1, 0.0), (0.1, 0.2, 0.0), (0.2, 0.2, 0.0), (0.3, 0.2, 0.0), (0.3, 0.3, 0.0),
        (0.4, 0.3, 0.0), (0.4, 0.4, 0.0), (0.5, 0.4, 0.0), (0.6, 0.4, 0.0), (0.6, 0.5, 0.0), (0.7, 0.5, 0.0),
        (0.7, 0.6, 0.0), (0.8, 0.6, 0.0), (0.9, 0.6, 0.0)]),
]

# ---- test/src/test_raycast_dda.cpp:246-286 (Intersection) -------------------------------------------------------
INTERSECTION_MAP = np.array([[0.6, -0.4, 0.0], [-2.1, -2.1, -2.1], [2.1, 2.1, 2.1]], np.float32)
INTERSECTION_CASTER = dict(map_grid=(0.05, 0.05, 0.05), dda_grid_size=0.2, ray_angle_half=0.01, hit_tolerance=0.0)
INTERSECTION_CASES = [
    dict(name="Intersection#1", begin=(0.0, 0.0, 0.0), end=(1.0, -0.55, 0.0), collision=True, expected=[
        (0.2, 0.0, 0.0), (0.2, -0.2, 0.0), (0.4, -0.2, 0.0), (0.6, -0.2, 0.0), (0.6, -0.4, 0.0)]),
    # the ray passes through the obstacle's voxel without hitting it
    dict(name="Intersection#2", begin=(0.0, 0.0, 0.0), end=(1.1, -0.55, 0.0), collision=False, expected=[
        (0.2, 0.0, 0.0), (0.2, -0.2, 0.0), (0.4, -0.2, 0.0), (0.6, -0.2, 0.0), (0.6, -0.4, 0.0), (0.8, -0.4, 0.0),
        (1.0, -0.4, 0.0)]),
]


# ---- test/src/test_raycast_dda.cpp:40-104 (Collision) -----------------------------------------------------------
def frange(start, stop, step, inclusive=False):
    """`for (float v = start; v < stop; v += step)` exactly as the C++ test loops evaluate it: the comparison and the
    increment are done in double (the literals are doubles), the result is narrowed back to float."""
    out = []
    v = np.float32(start)
    while (float(v) <= stop) if inclusive else (float(v) < stop):
        out.append(v)
        v = np.float32(float(v) + step)
    return out


def collision_wall_map():
    pts = [(0.5, y, z) for y in frange(-1.0, 1.0, 0.1) for z in frange(-1.0, 1.0, 0.1)]
    pts += [(-2.05, -2.05, -2.05), (2.05, 2.05, 2.05)]
    return np.array(pts, np.float32)


COLLISION_HIT_RANGE = np.float32(np.sqrt(3.0) * 0.1)
COLLISION_CASTER = dict(map_grid=(0.1, 0.1, 0.1), dda_grid_size=0.1, ray_angle_half=0.5,
                        hit_tolerance=float(COLLISION_HIT_RANGE))


def collision_rays_must_hit():
    """every ray must collide within 0.2 of (0.5, y, z) — test_raycast_dda.cpp:59-77"""
    return [((0.0, 0.0, 0.0), (1.0, float(np.float32(float(y) * 2.0)), float(np.float32(float(z) * 2.0))),
             (0.5, float(y), float(z)))
            for y in frange(-0.8, 0.8, 0.11) for z in frange(-0.8, 0.8, 0.13)]


def collision_rays_must_miss():
    """rays that stop epsilon short of the wall (:78-92) and one that leaves sideways (:93-103)"""
    eps = np.float32(0.05)
    x = float(np.float32(0.5 - float(COLLISION_HIT_RANGE) - float(eps)))
    rays = [((0.0, 0.0, 0.0), (x, float(y), float(z))) for y in frange(-1.0, 1.0, 0.11) for z in frange(-1.0, 1.0, 0.13)]
    rays.append(((0.0, 0.0, 0.0), (0.5, 3.0, 0.0)))
    return rays


# ---- test/src/test_raycast_dda.cpp:106-155 (CollisionTolerance) --------------------------------------------------
def tolerance_wall_map():
    pts = [(0.5, y, z) for y in frange(-1.0, 1.0, 0.05) for z in frange(-1.0, 1.0, 0.1)]
    pts += [(-2.05, -2.05, -2.05), (2.05, 2.05, 2.05)]
    return np.array(pts, np.float32)


TOLERANCE_CASES = [
    dict(caster=dict(map_grid=(0.05, 0.1, 0.1), dda_grid_size=0.1, ray_angle_half=0.5,
                     hit_tolerance=float(np.sqrt(3.0) * 0.1)),
         begin=(0.0, 0.0, 0.0), end=(0.5, 0.0, 0.0), collision=True),
    dict(caster=dict(map_grid=(0.1, 0.15, 0.15), dda_grid_size=0.15, ray_angle_half=0.5,
                     hit_tolerance=float(np.float32(np.sqrt(3.0) * 0.15))),
         begin=(0.0, 0.0, 0.0), end=(float(np.float32(0.5 - float(np.float32(np.sqrt(3.0) * 0.15)))), 0.0, 0.0),
         collision=False),
]

# ---- test/src/test_chunked_kdtree.cpp:38-88 ------------------------------------------------------------------------
KDTREE_MAP = np.array([[0.5, 0.5, 0.5], [0.8, 0.0, 0.0], [1.3, 0.0, 0.0], [0.0, 0.2, 0.0], [0.0, -0.3, 0.0]], np.float32)
KDTREE_CHUNK = (1.0, 0.3)
KDTREE_RADIUS = 0.3
KDTREE_QUERIES = np.array([[0.5, 0.5, 0.5], [0.5, 0.4, 0.5], [1.05, 0.0, 0.0], [1.1, 0.0, 0.0], [0.0, -0.05, 0.0],
                           [0.0, -0.15, 0.0]], np.float32)
KDTREE_EXPECTED = [0, 0, 1, 2, 3, 4]


# ---- test/src/test_quat.cpp:234-273 (vector rotation table), :164-173 (norm) ----------------------------------------
def quat_axis_angle(axis, ang):
    """Quat::setAxisAng, include/mcl_3dl/quat.h:218-227 (float)."""
    a = np.asarray(axis, np.float32)
    a = a / np.float32(np.sqrt(np.float32(a @ a)))
    s = np.float32(np.sin(np.float32(ang) / np.float32(2)))
    q = np.array([a[0] * s, a[1] * s, a[2] * s, np.float32(np.cos(np.float32(ang) / np.float32(2)))], np.float32)
    return q / np.float32(np.sqrt(np.float32(q @ q)))


QUAT_ROTATIONS = [((1.0, 0.0, 0.0), np.pi / 2.0), ((0.0, 1.0, 0.0), -np.pi / 2.0), ((0.0, 0.0, 1.0), -np.pi / 2.0)]
QUAT_VECS = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)]
# v_ans[j][i] = r[j] * v[i]
QUAT_ANSWERS = [
    [(1.0, 0.0, 0.0), (0.0, 0.0, 1.0), (0.0, -1.0, 0.0)],
    [(0.0, 0.0, 1.0), (0.0, 1.0, 0.0), (-1.0, 0.0, 0.0)],
    [(0.0, -1.0, 0.0), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0)],
]

# ---- test/src/test_pf.cpp:330-391 (entropy) ----------------------------------------------------------------------
PF_ENTROPY_CASES = [
    dict(lik=[1.0] + [0.0] * 9, entropy=0.0, tol=0.0),
    dict(lik=[0.1] * 10, entropy=2.303, tol=1e-3),
]
PF_ENTROPY_ORDER = (
    [0.025] * 4 + [0.4] * 2 + [0.025] * 4,   # narrower peak -> lower entropy
    [0.025] * 2 + [0.15] * 6 + [0.025] * 2,
)


# ---- test/src/test_beam_likelihood.cpp:81-135 fixture (upstream prints, does not assert: frozen as our golden) --------
def beam_wall_fixture():
    raw = [(2.0, float(y), float(z)) for y in frange(-0.2, 0.2, 0.1, True) for z in frange(-0.2, 0.2, 0.1, True)]
    raw_pc = np.array(raw + [(0.0, 0.0, 5.0), (0.0, 0.0, -5.0)], np.float32)
    pc_map = np.concatenate([raw_pc, np.array([[-100.05, 100.0, -0.05], [100.0, -100.05, 4.0]], np.float32)], 0)
    return raw_pc, pc_map
