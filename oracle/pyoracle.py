"""ctypes front-end for the two CPU oracles.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (mcl_3dl_amd) never does.

  Oracle("ref")  -> oracle/_ref/libmcl3dl_ref.so  : the real reference sources (built by oracle/Makefile
                    from /root/reference + stand-in headers), C-ABI prefix ``ref_``
  Oracle("port") -> oracle/libmcl3dl_oracle.so    : the plain-C restatement (mcl3dl_oracle.c), prefix ``orc_``

Both libraries export the same functions with the same signatures, so the wrapper is shared.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {
    "ref": (os.path.join(_HERE, "_ref", "libmcl3dl_ref.so"), "ref_"),
    "port": (os.path.join(_HERE, "libmcl3dl_oracle.so"), "orc_"),
}

BEAM_STATUS = {0: "SHORT", 1: "HIT", 2: "LONG", 3: "TOTAL_REFLECTION"}

_f = C.c_float
_d = C.c_double
_sz = C.c_size_t
_u32 = C.c_uint32
_u64 = C.c_uint64
_i = C.c_int
_p = C.c_void_p


def available(kind):
    return os.path.exists(_LIBS[kind][0])


def _fp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


def _u32a(a, n):
    if a is None:
        return np.zeros(n, dtype=np.uint32)
    return np.ascontiguousarray(a, dtype=np.uint32).reshape(-1)


class LikelihoodParams:
    """Defaults of LidarMeasurementModelLikelihoodParameters (include/mcl_3dl/parameters.h:67-76)."""

    def __init__(self, **kw):
        self.match_dist_min = 0.2
        self.match_dist_flat = 0.05
        self.match_weight = 5.0
        self.num_points = 96
        self.num_points_global = 8
        self.clip_near = 0.5
        self.clip_far = 10.0
        self.clip_z_min = -2.0
        self.clip_z_max = 2.0
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


class BeamParams:
    """Defaults of LidarMeasurementModelBeamParameters (include/mcl_3dl/parameters.h:96-112),
    except use_raycast_using_dda, which this build always wants on (north_star names the DDA)."""

    def __init__(self, **kw):
        self.map_grid_x = 0.1
        self.map_grid_y = 0.1
        self.map_grid_z = 0.1
        self.dda_grid_size = 0.2
        self.ray_angle_half = 0.25 * np.pi / 180.0
        self.hit_range = 0.3
        self.beam_likelihood_min = 0.2
        self.num_points = 3
        self.num_points_global = 0
        self.ang_total_ref = np.pi / 6.0
        self.filter_label_max = 0xFFFFFFFF
        self.add_penalty_short_only_mode = True
        self.use_raycast_using_dda = True
        self.clip_near = 0.5
        self.clip_far = 4.0
        self.clip_z_min = -2.0
        self.clip_z_max = 2.0
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


class Oracle:
    def __init__(self, kind="ref", chunk_length=20.0, max_search_radius=0.4):
        path, prefix = _LIBS[kind]
        if not os.path.exists(path):
            raise RuntimeError("oracle library %s is not built (run `make -C oracle`)" % path)
        self.kind = kind
        self.lib = C.CDLL(path)
        self.px = prefix
        self._sig()
        self.h = self._fn("create")(chunk_length, max_search_radius)
        self.n_map = 0

    def _fn(self, name):
        return getattr(self.lib, self.px + name)

    def _sig(self):
        def s(name, res, args):
            f = self._fn(name)
            f.restype = res
            f.argtypes = args

        s("create", _p, [_f, _f])
        s("destroy", None, [_p])
        s("max_threads", _i, [])
        s("set_map", None, [_p, _p, _p, _sz, _u64, _p, _f])
        s("set_likelihood_params", None, [_p, _f, _f, _f, _u32, _u32, _f, _f, _f, _f])
        s("set_beam_params", None, [_p, _f, _f, _f, _f, _f, _f, _f, _u32, _u32, _f, _u32, _i, _i, _f, _f, _f, _f])
        s("radius_search", None, [_p, _p, _sz, _f, _p, _p, _p])
        s("transform", None, [_p, _p, _sz, _p])
        s("quat_rotate", None, [_p, _p, _p])
        s("likelihood_measure", _d, [_p, _p, _sz, _p, _p, _sz, _p, _p, _i])
        s("beam_measure", _d, [_p, _p, _sz, _p, _p, _sz, _p, _sz, _p, _p, _i])
        s("beam_status", None, [_p, _p, _p, _sz, _p, _p])
        s("dda_waypoints", _i, [_p, _d, _d, _d, _d, _d, _d, _p, _p, _p, _i, _p, _p, _i])
        s("pf_measure", _i, [_p, _p, _sz, _p])
        s("expectation", None, [_p, _p, _p, _sz, _p, _p, _p])
        s("covariance", None, [_p, _p, _sz, _p, _p])
        if self.kind == "ref":
            s("set_flann_epsilon_mode", None, [_i])
            s("resample", None, [_p, _p, _sz, C.c_uint, _p, _p, _p])
            s("resample_draws", None, [C.c_uint, _f, _p, _sz, _p, _p])
            s("resize", None, [_p, _p, _sz, _sz, _p, _p])
            s("voxel_grid", _sz, [_p, _p, _sz, _p, _p, _p, _sz])
            s("clip", _sz, [_p, _i, _p, _p, _sz, _p, _p, _sz, C.POINTER(_sz)])
            s("filter_uniform", _sz, [_p, _i, _p, _p, _sz, C.c_uint, _p, _p, _sz])
            s("uniform_indices", None, [C.c_uint, _sz, _sz, _p])
            s("match_split", None, [_p, _p, _p, _sz, _d, _d, _p, _p])
        else:
            s("resample_pstep", _f, [_p, _sz, _sz])
            s("resample_plan", None, [_p, _sz, _sz, _i, _f, _p, _p])
            s("resample_apply", None, [_p, _p, _p, _p, _sz, _p])
        s("measure_update", _d, [_p, _p, _p, _p, _sz, _p, _sz, _p, _p, _sz, _p, _sz, _f,
                                 _p, _p, _p, _p, _p, _p, _p])

    def close(self):
        if self.h:
            self._fn("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_flann_epsilon_mode(self, mode):
        """Reference-backed oracle only. 1: the kd-tree stand-in honours setEpsilon() through a restated FLANN
        KDTreeSingleIndex with (1 + eps) pruning — a probe of what the node's eps = map_grid_min / 16 is worth; 0 (default):
        exact search, the definition parity is measured against. PROCESS-WIDE: set it back to 0 when done."""
        self._fn("set_flann_epsilon_mode")(int(mode))

    # ---- SURVEY.md 8f-2 / 8f-4 (reference-backed oracle only) ---------------------------------------------------------
    def voxel_grid(self, xyz, label, leaf):
        """pcl::VoxelGrid as the node configures it (restated shim, oracle/shims/pcl/filters/voxel_grid.h)."""
        xyz = _f32(xyz, 3)
        lab = _u32a(label, len(xyz))
        lf = _f32(leaf)
        ox = np.zeros((len(xyz), 3), np.float32)
        ol = np.zeros(len(xyz), np.uint32)
        n = self._fn("voxel_grid")(_fp(xyz), _fp(lab), len(xyz), _fp(lf), _fp(ox), _fp(ol), len(xyz))
        return ox[:n].copy(), ol[:n].copy()

    def clip(self, model, xyz, label=None):
        """The clip step of the reference's filter() (model 0 = likelihood, 1 = beam) and the num_points_ it would draw."""
        xyz = _f32(xyz, 3)
        lab = _u32a(label, len(xyz))
        ox = np.zeros((len(xyz), 3), np.float32)
        ol = np.zeros(len(xyz), np.uint32)
        num = C.c_size_t(0)
        n = self._fn("clip")(self.h, model, _fp(xyz), _fp(lab), len(xyz), _fp(ox), _fp(ol), len(xyz), C.byref(num))
        return ox[:n].copy(), ol[:n].copy(), int(num.value)

    def filter_uniform(self, model, xyz, label, seed, capacity):
        """The reference's filter() with the reference's PointCloudUniformSampler (engine seeded with `seed`)."""
        xyz = _f32(xyz, 3)
        lab = _u32a(label, len(xyz))
        ox = np.zeros((capacity, 3), np.float32)
        ol = np.zeros(capacity, np.uint32)
        n = self._fn("filter_uniform")(self.h, model, _fp(xyz), _fp(lab), len(xyz), seed, _fp(ox), _fp(ol), capacity)
        assert n <= capacity
        return ox[:n].copy(), ol[:n].copy()

    def uniform_indices(self, seed, n_clipped, num):
        out = np.zeros(num, np.uint32)
        self._fn("uniform_indices")(seed, n_clipped, num, _fp(out))
        return out

    def match_split(self, pose7, xyz, unmatch_dist, match_dist):
        pose7 = _f32(pose7)
        xyz = _f32(xyz, 3)
        cls = np.zeros(len(xyz), np.uint8)
        out = np.zeros((len(xyz), 3), np.float32)
        self._fn("match_split")(self.h, _fp(pose7), _fp(xyz), len(xyz), unmatch_dist, match_dist, _fp(cls), _fp(out))
        return cls, out

    def max_threads(self):
        return int(self._fn("max_threads")())

    # ---- configuration -------------------------------------------------------------------------
    def set_map(self, xyz, label=None, stamp=1, dist_weight=(1.0, 1.0, 1.0), epsilon=-1.0):
        xyz = _f32(xyz, 3)
        lab = _u32a(label, len(xyz))
        dw = None if dist_weight is None else _f32(dist_weight)
        self.n_map = len(xyz)
        self._fn("set_map")(self.h, _fp(xyz), _fp(lab), len(xyz), stamp, _fp(dw), epsilon)

    def set_likelihood_params(self, p=None):
        p = p or LikelihoodParams()
        self._fn("set_likelihood_params")(self.h, p.match_dist_min, p.match_dist_flat, p.match_weight,
                                          int(p.num_points), int(p.num_points_global), p.clip_near, p.clip_far,
                                          p.clip_z_min, p.clip_z_max)

    def set_beam_params(self, p=None):
        p = p or BeamParams()
        self._fn("set_beam_params")(self.h, p.map_grid_x, p.map_grid_y, p.map_grid_z, p.dda_grid_size,
                                    p.ray_angle_half, p.hit_range, p.beam_likelihood_min, int(p.num_points),
                                    int(p.num_points_global), p.ang_total_ref, int(p.filter_label_max),
                                    int(bool(p.add_penalty_short_only_mode)), int(bool(p.use_raycast_using_dda)),
                                    p.clip_near, p.clip_far, p.clip_z_min, p.clip_z_max)

    # ---- primitives ----------------------------------------------------------------------------
    def radius_search(self, q, radius):
        q = _f32(q, 3)
        n = len(q)
        found = np.zeros(n, np.int32)
        idx = np.zeros(n, np.int32)
        sq = np.zeros(n, np.float32)
        self._fn("radius_search")(self.h, _fp(q), n, radius, _fp(found), _fp(idx), _fp(sq))
        return found, idx, sq

    def transform(self, pose7, xyz):
        pose7 = _f32(pose7)
        xyz = _f32(xyz, 3)
        out = np.empty_like(xyz)
        self._fn("transform")(_fp(pose7), _fp(xyz), len(xyz), _fp(out))
        return out

    def quat_rotate(self, q4, v3):
        q4 = _f32(q4)
        v3 = _f32(v3)
        out = np.zeros(3, np.float32)
        self._fn("quat_rotate")(_fp(q4), _fp(v3), _fp(out))
        return out

    def likelihood_measure(self, poses, scan_xyz, threads=1, return_time=False):
        poses = _f32(poses, 7)
        scan = _f32(scan_xyz, 3)
        lab = _u32a(None, len(scan))
        lik = np.zeros(len(poses), np.float32)
        qual = np.zeros(len(poses), np.float32)
        dt = self._fn("likelihood_measure")(self.h, _fp(poses), len(poses), _fp(scan), _fp(lab), len(scan),
                                            _fp(lik), _fp(qual), threads)
        return (lik, qual, dt) if return_time else (lik, qual)

    def beam_measure(self, poses, scan_xyz, scan_label, origins, threads=1, return_time=False):
        poses = _f32(poses, 7)
        scan = _f32(scan_xyz, 3)
        lab = _u32a(scan_label, len(scan))
        org = _f32(origins, 3)
        lik = np.zeros(len(poses), np.float32)
        qual = np.zeros(len(poses), np.float32)
        dt = self._fn("beam_measure")(self.h, _fp(poses), len(poses), _fp(scan), _fp(lab), len(scan), _fp(org),
                                      len(org), _fp(lik), _fp(qual), threads)
        return (lik, qual, dt) if return_time else (lik, qual)

    def beam_status(self, begin, end):
        b = _f32(begin, 3)
        e = _f32(end, 3)
        st = np.zeros(len(b), np.int32)
        hit = np.zeros(len(b), np.int32)
        self._fn("beam_status")(self.h, _fp(b), _fp(e), len(b), _fp(st), _fp(hit))
        return st, hit

    def dda_waypoints(self, map_grid, dda_grid_size, ray_angle_half, hit_tolerance, begin, end, max_out=4096,
                      stop_at_collision=True):
        b = _f32(begin)
        e = _f32(end)
        out = np.zeros((max_out, 3), np.float32)
        col = C.c_int(0)
        hit = C.c_int(-1)
        n = self._fn("dda_waypoints")(self.h, map_grid[0], map_grid[1], map_grid[2], dda_grid_size, ray_angle_half,
                                      hit_tolerance, _fp(b), _fp(e), _fp(out), max_out, C.byref(col), C.byref(hit),
                                      int(stop_at_collision))
        return out[:min(n, max_out)].copy(), bool(col.value), int(hit.value), n

    def pf_measure(self, weights, likelihood):
        w = _f32(weights).copy()
        lk = _f32(likelihood)
        ent = C.c_float(0)
        restored = self._fn("pf_measure")(_fp(w), _fp(lk), len(w), C.byref(ent))
        return w, float(ent.value), bool(restored)

    def expectation(self, poses, weights, bias=None):
        """pf::expectationBiased + max + maxBiased (pf.h:294-303,361-390)."""
        poses = _f32(poses, 7)
        w = _f32(weights)
        b = None if bias is None else _f32(bias)
        mean = np.zeros(7, np.float32)
        im, ib = C.c_int(0), C.c_int(0)
        self._fn("expectation")(_fp(poses), _fp(w), _fp(b), len(poses), _fp(mean), C.byref(im), C.byref(ib))
        return mean, int(im.value), int(ib.value)

    def covariance(self, poses, weights):
        """pf::covariance(1.0, 1.0) (pf.h:304-360); returns (6x6 covariance, the expectation(1.0) it is centred on)."""
        poses = _f32(poses, 7)
        w = _f32(weights)
        cov = np.zeros((6, 6), np.float32)
        mean = np.zeros(7, np.float32)
        self._fn("covariance")(_fp(poses), _fp(w), len(poses), _fp(cov), _fp(mean))
        return cov, mean

    # ---- resampling: the real pf.h code with a seeded engine ("ref") / the deterministic restatement ("port") ------
    def resample(self, state13, weights, seed, sigma6):
        assert self.kind == "ref"
        s = _f32(state13, 13)
        w = _f32(weights)
        out = np.zeros_like(s)
        wout = np.zeros_like(w)
        self._fn("resample")(_fp(s), _fp(w), len(s), int(seed), _fp(_f32(sigma6)), _fp(out), _fp(wout))
        return out, wout

    def resample_draws(self, seed, pstep, sigma6, n_dup):
        assert self.kind == "ref"
        ip = C.c_float(0)
        noise = np.zeros((max(n_dup, 1), 13), np.float32)
        self._fn("resample_draws")(int(seed), float(pstep), _fp(_f32(sigma6)), n_dup, C.byref(ip), _fp(noise))
        return float(ip.value), noise[:n_dup]

    def resize(self, state13, weights, n_out):
        assert self.kind == "ref"
        s = _f32(state13, 13)
        w = _f32(weights)
        out = np.zeros((n_out, 13), np.float32)
        wout = np.zeros(n_out, np.float32)
        self._fn("resize")(_fp(s), _fp(w), len(s), n_out, _fp(out), _fp(wout))
        return out, wout

    def resample_pstep(self, weights, n_out):
        assert self.kind == "port"
        w = _f32(weights)
        return float(self._fn("resample_pstep")(_fp(w), len(w), n_out))

    def resample_plan(self, weights, n_out, mode, initial_p):
        assert self.kind == "port"
        w = _f32(weights)
        src = np.zeros(n_out, np.uint32)
        dup = np.zeros(n_out, np.uint8)
        self._fn("resample_plan")(_fp(w), len(w), n_out, int(mode), float(initial_p), _fp(src), _fp(dup))
        return src, dup

    def resample_apply(self, state13, source, dup, noise13):
        assert self.kind == "port"
        s = _f32(state13, 13)
        src = np.ascontiguousarray(source, np.uint32)
        d = np.ascontiguousarray(dup, np.uint8)
        nz = _f32(noise13 if len(noise13) else np.zeros((1, 13)), 13)
        out = np.zeros((len(src), 13), np.float32)
        self._fn("resample_apply")(_fp(s), _fp(src), _fp(d), _fp(nz), len(src), _fp(out))
        return out

    def measure_update(self, poses, weights, scan_lik, scan_beam, scan_beam_label, origins, odom_err=None,
                       odom_sigma=1.0):
        """The node's LiDAR measurement update (src/mcl_3dl.cpp:398-426 + pf.h:252-279)."""
        poses = _f32(poses, 7)
        n_p = len(poses)
        w = _f32(weights).copy()
        sl = _f32(scan_lik, 3)
        sb = _f32(scan_beam, 3)
        sbl = _u32a(scan_beam_label, len(sb))
        org = _f32(origins, 3)
        oe = None if odom_err is None else _f32(odom_err, 3)
        lik = np.zeros(n_p, np.float32)
        beam = np.zeros(n_p, np.float32)
        qual = np.zeros(n_p, np.float32)
        ent = C.c_float(0)
        rmin = C.c_float(0)
        rmax = C.c_float(0)
        rest = C.c_int(0)
        dt = self._fn("measure_update")(self.h, _fp(poses), _fp(oe), _fp(w), n_p, _fp(sl), len(sl), _fp(sb),
                                        _fp(sbl), len(sb), _fp(org), len(org), odom_sigma, _fp(lik), _fp(beam),
                                        _fp(qual), C.byref(ent), C.byref(rmin), C.byref(rmax), C.byref(rest))
        return dict(weights=w, lik=lik, beam=beam, quality=qual, entropy=float(ent.value),
                    match_ratio_min=float(rmin.value), match_ratio_max=float(rmax.value),
                    restored=bool(rest.value), seconds=dt)
