"""ctypes binding of the C ABI in include/mcl3dl_hip.h (libmcl3dl_hip.so, built in-tree by __graft_entry__.build()).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is usable, loading / creating an
engine raises.  The product never imports anything from oracle/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCL3DL_HIP_LIB: another build of the same library (A/B measurements: mcl_3dl_amd/variants/); it must exist — a missing
# file is an error, never a reason to fall back to anything
LIB_PATH = os.environ.get("MCL3DL_HIP_LIB") or os.path.join(_HERE, "libmcl3dl_hip.so")

KERNEL_LIKELIHOOD, KERNEL_BEAM, KERNEL_PF, KERNEL_UPDATE, KERNEL_STAGE = 0, 1, 2, 3, 4
BEAM_STATUS = {0: "SHORT", 1: "HIT", 2: "LONG", 3: "TOTAL_REFLECTION"}

_f, _d, _sz, _u32, _u64, _i, _p = C.c_float, C.c_double, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p

# name -> (restype, argtypes): every symbol include/mcl3dl_hip.h declares
SIGNATURES = {
    "mcl3dl_hip_abi_version": (_i, []),
    "mcl3dl_hip_create": (_i, [C.POINTER(_p), _i]),
    "mcl3dl_hip_destroy": (None, [_p]),
    "mcl3dl_hip_last_error": (C.c_char_p, [_p]),
    "mcl3dl_hip_set_stream": (_i, [_p, _p]),
    "mcl3dl_hip_get_stream": (_p, [_p]),
    "mcl3dl_hip_synchronize": (_i, [_p]),
    "mcl3dl_hip_set_map": (_i, [_p, _p, _p, _sz, _u64, _p]),
    "mcl3dl_hip_set_likelihood_params": (_i, [_p, _f, _f, _f]),
    "mcl3dl_hip_set_beam_params": (_i, [_p, _f, _f, _f, _f, _f, _f, _f, _u32, _f, _u32, _i]),
    "mcl3dl_hip_upload_poses": (_i, [_p, _p, _sz]),
    "mcl3dl_hip_scan_order": (_i, [_p, _p, _sz]),
    "mcl3dl_hip_scan_order_host": (_i, [_p, _sz, _p]),
    "mcl3dl_hip_measure_batch": (_i, [_p, _p, _sz, _p, _sz, _p, _p, _sz, _p, _sz, _p, _p, _p]),
    "mcl3dl_hip_measure_batch_begin": (_i, [_p, _p, _sz, _p, _sz, _p, _p, _sz, _p, _sz, _p, _p, _p, _sz]),
    "mcl3dl_hip_measure_batch_wait": (_i, [_p, _sz, _p]),
    "mcl3dl_hip_measure_batch_end": (_i, [_p]),
    "mcl3dl_hip_pf_measure": (_i, [_p, _p, _p, _p, _p, _p, _sz, _p, _p, _p, _p]),
    "mcl3dl_hip_measure_update": (_i, [_p, _p, _p, _p, _sz, _p, _sz, _p, _p, _sz, _p, _sz, _p, _p, _p, _p, _p, _p, _p]),
    "mcl3dl_hip_host_alloc": (_i, [_p, _sz, C.POINTER(_p)]),
    "mcl3dl_hip_host_free": (_i, [_p, _p]),
    "mcl3dl_hip_beam_status": (_i, [_p, _p, _p, _sz, _p, _p]),
    "mcl3dl_hip_radius_search": (_i, [_p, _p, _sz, _f, _p, _p]),
    "mcl3dl_hip_device_count": (_i, []),
    "mcl3dl_hip_dda_trace": (_i, [_p, _p, _p, _p, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "mcl3dl_hip_sort_pairs": (_i, [_p, _p, _p, _sz, _i, _p, _p]),
    "mcl3dl_hip_expectation": (_i, [_p, _p, _p, _p, _sz, _p, _p, _p, _p]),
    "mcl3dl_hip_covariance": (_i, [_p, _p, _p, _sz, _p, _sz, _p, _p]),
    "mcl3dl_hip_expectation_device": (_i, [_p, _p, _p, _p, _sz, _p, _p, _p, _p]),
    "mcl3dl_hip_covariance_device": (_i, [_p, _p, _p, _sz, _p, _sz, _p, _p]),
    "mcl3dl_hip_resample_begin": (_i, [_p, _p, _sz, _sz, C.POINTER(_f)]),
    "mcl3dl_hip_resample_plan": (_i, [_p, _i, _f, _p, _p, C.POINTER(_sz)]),
    "mcl3dl_hip_resample_apply": (_i, [_p, _p, _p, _sz, _p]),
    "mcl3dl_hip_resample_apply_device": (_i, [_p, _p, _p, _sz, _p]),
    "mcl3dl_hip_moments_partial_device": (_i, [_p, _p, _p, _p, _sz, _p]),
    "mcl3dl_hip_moments_finish": (_i, [_p, _i, _p, _p, C.POINTER(_f), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mcl3dl_hip_covariance_partial_device": (_i, [_p, _p, _p, _sz, _p, _sz, _p, _p]),
    "mcl3dl_hip_covariance_finish": (_i, [_p, _p]),
    "mcl3dl_hip_update_device": (_i, [_p, _p, _sz, _p, _p, _p, _p, _p, _p]),
    "mcl3dl_hip_resample_begin_device": (_i, [_p, _p, _sz, _sz, C.POINTER(_f)]),
    "mcl3dl_hip_resample_apply_slice_device": (_i, [_p, _p, _p, _sz, _sz, _sz, _p]),
    "mcl3dl_hip_upload_scan": (_i, [_p, _p, _sz, _p, _p, _sz, _p, _sz]),
    "mcl3dl_hip_measure_device": (_i, [_p, _p, _sz, _p, _p, _p]),
    "mcl3dl_hip_pf_partial_device": (_i, [_p, _p, _p, _p, _p, _p, _sz, _i, _i, _p]),
    "mcl3dl_hip_pf_apply_device": (_i, [_p, _p, _sz, _i, _p, _p]),
    "mcl3dl_hip_set_kernel_timing": (_i, [_p, _i]),
    "mcl3dl_hip_get_kernel_time": (_i, [_p, _i, C.POINTER(_d), C.POINTER(_u64)]),
    "mcl3dl_hip_reset_kernel_time": (_i, [_p]),
    "mcl3dl_hip_workload_stats": (_i, [_p, _p, _sz, _p]),
    "mcl3dl_hip_memory_footprint": (_i, [_p, _p]),
    "mcl3dl_hip_set_option": (_i, [_p, C.c_char_p, _d]),
    "mcl3dl_hip_scan_begin": (_i, [_p, _p, _p, _sz, _p, _p, _p, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    "mcl3dl_hip_scan_begin_pointcloud2": (_i, [_p, _p, _sz, _u32, _i, _i, _i, _i, _u32, _p, _p, _p, C.POINTER(_sz),
                                              C.POINTER(_sz), C.POINTER(_sz)]),
    "mcl3dl_hip_scan_finish": (_i, [_p, _p, _sz, _p, _sz, _p, _sz]),
    "mcl3dl_hip_scan_download": (_i, [_p, _i, _p, _p, _sz, C.POINTER(_sz)]),
    "mcl3dl_hip_set_map_pointcloud2": (_i, [_p, _p, _sz, _u32, _i, _i, _i, _i, _p, _u64, _p, C.POINTER(_sz)]),
    "mcl3dl_hip_set_map_downsampled": (_i, [_p, _p, _p, _sz, _p, _u64, _p, C.POINTER(_sz)]),
    "mcl3dl_hip_map_update": (_i, [_p, _p, _p, _sz, _p, _u64, C.POINTER(_sz), _p]),
    "mcl3dl_hip_map_update_pointcloud2": (_i, [_p, _p, _sz, _u32, _i, _i, _i, _i, _p, _u64, C.POINTER(_sz), _p]),
    "mcl3dl_hip_map_download": (_i, [_p, _p, _p, _sz, C.POINTER(_sz)]),
    "mcl3dl_hip_match_split": (_i, [_p, _p, _p, _sz, _f, _d, _p, _sz, C.POINTER(_sz), _p, _sz, C.POINTER(_sz)]),
    "mcl3dl_hip_group_create": (_i, [C.POINTER(_p), C.POINTER(_i), _i]),
    "mcl3dl_hip_group_destroy": (None, [_p]),
    "mcl3dl_hip_group_last_error": (C.c_char_p, [_p]),
    "mcl3dl_hip_group_size": (_i, [_p]),
    "mcl3dl_hip_group_context": (_p, [_p, _i]),
    "mcl3dl_hip_group_shard": (_i, [_sz, _i, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    "mcl3dl_hip_group_set_map": (_i, [_p, _p, _p, _sz, _u64, _p]),
    "mcl3dl_hip_group_set_likelihood_params": (_i, [_p, _f, _f, _f]),
    "mcl3dl_hip_group_set_beam_params": (_i, [_p, _f, _f, _f, _f, _f, _f, _f, _u32, _f, _u32, _i]),
    "mcl3dl_hip_group_set_option": (_i, [_p, C.c_char_p, _d]),
    "mcl3dl_hip_group_upload_poses": (_i, [_p, _p, _sz]),
    "mcl3dl_hip_group_measure_batch": (_i, [_p, _p, _sz, _p, _sz, _p, _p, _sz, _p, _sz, _p, _p, _p]),
    "mcl3dl_hip_group_measure_batch_begin": (_i, [_p, _p, _sz, _p, _sz, _p, _p, _sz, _p, _sz, _p, _p, _p, _sz]),
    "mcl3dl_hip_group_measure_batch_wait": (_i, [_p, _sz, _p]),
    "mcl3dl_hip_group_measure_batch_end": (_i, [_p]),
    "mcl3dl_hip_group_measure_update": (_i, [_p, _p, _p, _p, _sz, _p, _sz, _p, _p, _sz, _p, _sz, _p, _p, _p, _p, _p, _p,
                                            _p]),
    "mcl3dl_hip_group_collective_stats": (_i, [_p, C.POINTER(_u64), C.POINTER(_u64)]),
    "mcl3dl_hip_group_upload_state": (_i, [_p, _p, _p, _sz]),
    "mcl3dl_hip_group_download_state": (_i, [_p, _p, _p, _sz]),
    "mcl3dl_hip_group_resident": (_sz, [_p]),
    "mcl3dl_hip_group_update_resident": (_i, [_p, _p, _p, _sz, _p, _p, _sz, _p, _sz, _p, _p, _p, _p, _p, _p, _p, _p]),
    "mcl3dl_hip_group_expectation": (_i, [_p, _p, _p, _p, _p, _p]),
    "mcl3dl_hip_group_covariance": (_i, [_p, _p, _p]),
    "mcl3dl_hip_group_resample_begin": (_i, [_p, _sz, _p]),
    "mcl3dl_hip_group_resample_plan": (_i, [_p, _i, _f, _p, _p, _p]),
    "mcl3dl_hip_group_resample_apply": (_i, [_p, _p, _sz]),
    "mcl3dl_hip_get_option": (_i, [_p, C.c_char_p, C.POINTER(_d)]),
    "mcl3dl_hip_index_stats": (_i, [_p, _p]),
}

_lib = None


class EngineError(RuntimeError):
    pass


def load_library():
    """dlopen libmcl3dl_hip.so and bind every declared symbol.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_benchloop = None


def load_benchloop():
    """tools/libmcl3dl_benchloop.so (measurement helper, not product): the timed host-buffer loop in C."""
    global _benchloop
    if _benchloop is None:
        load_library()
        path = os.path.join(os.path.dirname(_HERE), "tools", "libmcl3dl_benchloop.so")
        if not os.path.exists(path):
            raise EngineError("%s is missing: run __graft_entry__.build()" % path)
        lib = C.CDLL(path)
        fn = lib.mcl3dl_benchloop_measure_update
        fn.restype = C.c_double
        fn.argtypes = [_p, _p, _p, _p, _p, _sz, _p, _sz, _p, _p, _sz, _p, _sz, _p, _p, _p, _i, C.c_double, _p]
        _benchloop = lib
    return _benchloop


def _np_f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, int):
        return C.c_void_p(a)
    return C.c_void_p(a.data_ptr())  # torch tensor (device memory)


def scan_order_host(scan_lik):
    """mcl3dl_hip_scan_order_host: the engine's order of a likelihood scan, computed on the host (no GPU needed)."""
    lib = load_library()
    sl = _np_f32(scan_lik, 3)
    order = np.zeros(len(sl), np.uint32)
    if lib.mcl3dl_hip_scan_order_host(_ptr(sl), len(sl), _ptr(order)) != 0:
        raise ValueError("mcl3dl_hip_scan_order_host rejected the scan")
    return order


def group_shard(n_p, n_devices, rank):
    """[begin, begin + count) of `rank` — the library's own shard rule (no GPU needed)."""
    lib = load_library()
    b, c = C.c_size_t(0), C.c_size_t(0)
    if lib.mcl3dl_hip_group_shard(n_p, n_devices, rank, C.byref(b), C.byref(c)) != 0:
        raise EngineError("group_shard: bad arguments")
    return int(b.value), int(c.value)


class Group:
    """N GPUs behind one handle in ONE process (include/mcl3dl_hip.h, "device groups"): host arrays in, host arrays out."""

    def __init__(self, device_ids=(0,), collective=None):
        self.lib = load_library()
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        h = C.c_void_p()
        rc = self.lib.mcl3dl_hip_group_create(C.byref(h), ids, len(device_ids))
        if rc != 0 or not h:
            raise EngineError("mcl3dl_hip_group_create(%s) failed with %d (no CPU fallback exists)" % (list(device_ids), rc))
        self.h = h
        self.n = len(device_ids)
        if collective is not None:
            self.set_option("collective", {"rccl": 0, "host": 1}[collective])

    def close(self):
        if getattr(self, "h", None):
            self.lib.mcl3dl_hip_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError("mcl3dl_hip group error %d: %s" % (rc, self.lib.mcl3dl_hip_group_last_error(self.h).decode()))

    def set_map(self, xyz, label=None, stamp=1, dist_weight=(1.0, 1.0, 1.0)):
        xyz = _np_f32(xyz, 3)
        lab = None if label is None else np.ascontiguousarray(label, dtype=np.uint32)
        dw = None if dist_weight is None else _np_f32(dist_weight)
        self._check(self.lib.mcl3dl_hip_group_set_map(self.h, _ptr(xyz), _ptr(lab), len(xyz), int(stamp), _ptr(dw)))

    def set_likelihood_params(self, match_dist_min=0.2, match_dist_flat=0.05, match_weight=5.0):
        self._check(self.lib.mcl3dl_hip_group_set_likelihood_params(self.h, match_dist_min, match_dist_flat, match_weight))

    def set_beam_params(self, map_grid=(0.1, 0.1, 0.1), dda_grid_size=0.2, ray_angle_half=0.25 * np.pi / 180.0,
                        hit_range=0.3, beam_likelihood_min=0.2, num_points=3, ang_total_ref=np.pi / 6.0,
                        filter_label_max=0xFFFFFFFF, add_penalty_short_only_mode=True):
        self._check(self.lib.mcl3dl_hip_group_set_beam_params(
            self.h, map_grid[0], map_grid[1], map_grid[2], dda_grid_size, ray_angle_half, hit_range,
            beam_likelihood_min, int(num_points), ang_total_ref, int(filter_label_max),
            int(bool(add_penalty_short_only_mode))))

    def set_option(self, name, value):
        self._check(self.lib.mcl3dl_hip_group_set_option(self.h, name.encode(), float(value)))

    def upload_poses(self, poses):
        poses = _np_f32(poses, 7)
        self._check(self.lib.mcl3dl_hip_group_upload_poses(self.h, _ptr(poses), len(poses)))
        self._n_uploaded = len(poses)

    def measure_batch_begin(self, poses, scan_lik, scan_beam=None, scan_beam_origin=None, origins=None, slice_particles=0,
                            out=None):
        """mcl3dl_hip_measure_batch_begin: returns the (lik, ratio, beam) arrays the batch fills in particle order
        (`out` = three caller-owned float32 arrays, e.g. host_array()s); measure_batch_wait(i) says how far they are valid."""
        sl, sb, so, og = Engine._scans(scan_lik, scan_beam, scan_beam_origin, origins)
        if poses is None:
            n_p = self._n_uploaded
        else:
            poses = _np_f32(poses, 7)
            n_p = len(poses)
        lik, ratio, beam = out if out is not None else (np.zeros(n_p, np.float32) for _ in range(3))
        # (a batch still open is ended by this very call and delivers into ITS arrays: they stay alive until it returns)
        previous = getattr(self, "_batch_keep", None)
        self._batch_keep = (poses, sl, sb, so, og, lik, ratio, beam, previous)
        self._check(self.lib.mcl3dl_hip_group_measure_batch_begin(self.h, _ptr(poses), n_p, _ptr(sl), len(sl), _ptr(sb), _ptr(so),
                    len(sb), _ptr(og), len(og), _ptr(lik), _ptr(ratio), _ptr(beam), int(slice_particles)))
        self._batch_keep = self._batch_keep[:8]
        return lik, ratio, beam

    def measure_batch_wait(self, particle):
        n = C.c_size_t(0)
        self._check(self.lib.mcl3dl_hip_group_measure_batch_wait(self.h, int(particle), C.byref(n)))
        return n.value

    def measure_batch_end(self):
        self._check(self.lib.mcl3dl_hip_group_measure_batch_end(self.h))
        self._batch_keep = None

    def measure_batch(self, poses, scan_lik, scan_beam=None, scan_beam_origin=None, origins=None):
        sl, sb, so, og = Engine._scans(scan_lik, scan_beam, scan_beam_origin, origins)
        if poses is None:
            n_p = self._n_uploaded
        else:
            poses = _np_f32(poses, 7)
            n_p = len(poses)
        lik, ratio, beam = (np.zeros(n_p, np.float32) for _ in range(3))
        self._check(self.lib.mcl3dl_hip_group_measure_batch(self.h, _ptr(poses), n_p, _ptr(sl), len(sl), _ptr(sb), _ptr(so),
                                                            len(sb), _ptr(og), len(og), _ptr(lik), _ptr(ratio), _ptr(beam)))
        return lik, ratio, beam

    def measure_update(self, poses, weights, scan_lik, scan_beam=None, scan_beam_origin=None, origins=None, extra=None):
        poses = _np_f32(poses, 7)
        n_p = len(poses)
        w = _np_f32(weights).copy()
        ex = None if extra is None else _np_f32(extra)
        sl, sb, so, og = Engine._scans(scan_lik, scan_beam, scan_beam_origin, origins)
        lik, ratio, beam = (np.zeros(n_p, np.float32) for _ in range(3))
        ent, rmin, rmax, rest = C.c_float(0), C.c_float(0), C.c_float(0), C.c_int(0)
        self._check(self.lib.mcl3dl_hip_group_measure_update(
            self.h, _ptr(poses), _ptr(ex), _ptr(w), n_p, _ptr(sl), len(sl), _ptr(sb), _ptr(so), len(sb), _ptr(og),
            len(og), _ptr(lik), _ptr(ratio), _ptr(beam), C.byref(ent), C.byref(rmin), C.byref(rmax), C.byref(rest)))
        return dict(weights=w, lik=lik, quality=ratio, beam=beam, entropy=float(ent.value),
                    match_ratio_min=float(rmin.value), match_ratio_max=float(rmax.value), restored=bool(rest.value))

    def collective_stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.mcl3dl_hip_group_collective_stats(self.h, C.byref(a), C.byref(b)))
        return dict(rccl=int(a.value), host=int(b.value))

    # ---- particles resident on the group's devices (include/mcl3dl_hip.h) -------------------------------------------------
    def upload_state(self, state13, weights=None):
        s = _np_f32(state13, 13)
        w = None if weights is None else _np_f32(weights)
        self._check(self.lib.mcl3dl_hip_group_upload_state(self.h, _ptr(s), _ptr(w), len(s)))

    def resident(self):
        return int(self.lib.mcl3dl_hip_group_resident(self.h))

    def download_state(self):
        n = self.resident()
        s, w = np.zeros((n, 13), np.float32), np.zeros(n, np.float32)
        self._check(self.lib.mcl3dl_hip_group_download_state(self.h, _ptr(s), _ptr(w), n))
        return s, w

    def update_resident(self, scan_lik, scan_beam=None, scan_beam_origin=None, origins=None, extra=None, fetch=True):
        n_p = self.resident()
        ex = None if extra is None else _np_f32(extra)
        sl, sb, so, og = Engine._scans(scan_lik, scan_beam, scan_beam_origin, origins)
        w, lik, ratio, beam = ((np.zeros(n_p, np.float32) for _ in range(4)) if fetch else (None, None, None, None))
        ent, rmin, rmax, rest = C.c_float(0), C.c_float(0), C.c_float(0), C.c_int(0)
        self._check(self.lib.mcl3dl_hip_group_update_resident(
            self.h, _ptr(ex), _ptr(sl), len(sl), _ptr(sb), _ptr(so), len(sb), _ptr(og), len(og), _ptr(w), _ptr(lik), _ptr(ratio),
            _ptr(beam), C.byref(ent), C.byref(rmin), C.byref(rmax), C.byref(rest)))
        return dict(weights=w, lik=lik, quality=ratio, beam=beam, entropy=float(ent.value),
                    match_ratio_min=float(rmin.value), match_ratio_max=float(rmax.value), restored=bool(rest.value))

    def expectation(self, bias=None):
        b = None if bias is None else _np_f32(bias)
        mean = np.zeros(7, np.float32)
        total = C.c_float(0)
        im, ib = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.mcl3dl_hip_group_expectation(self.h, _ptr(b), _ptr(mean), C.byref(total), C.byref(im), C.byref(ib)))
        return mean, float(total.value), int(im.value), int(ib.value)

    def covariance(self, mean7):
        m = _np_f32(mean7)
        cov = np.zeros((6, 6), np.float32)
        self._check(self.lib.mcl3dl_hip_group_covariance(self.h, _ptr(m), _ptr(cov)))
        return cov

    def resample_begin(self, n_out=0):
        pstep = C.c_float(0)
        self._check(self.lib.mcl3dl_hip_group_resample_begin(self.h, int(n_out), C.byref(pstep)))
        self._rs_n_out = n_out or self.resident()
        return float(pstep.value)

    def resample_plan(self, mode, initial_p=0.0):
        nd = C.c_size_t(0)
        n_out = getattr(self, "_rs_n_out", 0) or self.resident()
        src = np.zeros(n_out, np.uint32)
        dup = np.zeros(n_out, np.uint8)
        self._check(self.lib.mcl3dl_hip_group_resample_plan(self.h, int(mode), float(initial_p), _ptr(src), _ptr(dup),
                                                            C.byref(nd)))
        return src, dup, int(nd.value)

    def resample_apply(self, noise13=None):
        nz = None if noise13 is None or len(noise13) == 0 else _np_f32(noise13, 13)
        self._check(self.lib.mcl3dl_hip_group_resample_apply(self.h, _ptr(nz), 0 if nz is None else len(nz)))


class Engine:
    """One context = one GPU.  Host methods take numpy arrays; *_device methods take torch CUDA tensors."""

    def __init__(self, device_id=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.mcl3dl_hip_create(C.byref(h), int(device_id))
        if rc != 0 or not h:
            raise EngineError("mcl3dl_hip_create(device %d) failed with %d: no usable gfx950 device "
                              "(this engine has no CPU fallback)" % (device_id, rc))
        self.h = h
        self.device_id = device_id

    def close(self):
        if getattr(self, "h", None):
            self.lib.mcl3dl_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError("mcl3dl_hip error %d: %s" % (rc, self.lib.mcl3dl_hip_last_error(self.h).decode()))

    # ---- configuration -----------------------------------------------------------------------------------------
    def set_stream(self, stream_handle):
        self._check(self.lib.mcl3dl_hip_set_stream(self.h, C.c_void_p(stream_handle) if stream_handle else None))

    def get_stream(self):
        """The hipStream_t the engine enqueues on, as an integer handle."""
        return int(self.lib.mcl3dl_hip_get_stream(self.h) or 0)

    def synchronize(self):
        self._check(self.lib.mcl3dl_hip_synchronize(self.h))

    def set_map(self, xyz, label=None, stamp=1, dist_weight=(1.0, 1.0, 1.0)):
        xyz = _np_f32(xyz, 3)
        lab = None if label is None else np.ascontiguousarray(label, dtype=np.uint32)
        dw = None if dist_weight is None else _np_f32(dist_weight)
        self._check(self.lib.mcl3dl_hip_set_map(self.h, _ptr(xyz), _ptr(lab), len(xyz), int(stamp), _ptr(dw)))

    def set_likelihood_params(self, match_dist_min=0.2, match_dist_flat=0.05, match_weight=5.0):
        self._check(self.lib.mcl3dl_hip_set_likelihood_params(self.h, match_dist_min, match_dist_flat, match_weight))

    def set_beam_params(self, map_grid=(0.1, 0.1, 0.1), dda_grid_size=0.2, ray_angle_half=0.25 * np.pi / 180.0,
                        hit_range=0.3, beam_likelihood_min=0.2, num_points=3, ang_total_ref=np.pi / 6.0,
                        filter_label_max=0xFFFFFFFF, add_penalty_short_only_mode=True):
        self._check(self.lib.mcl3dl_hip_set_beam_params(
            self.h, map_grid[0], map_grid[1], map_grid[2], dda_grid_size, ray_angle_half, hit_range,
            beam_likelihood_min, int(num_points), ang_total_ref, int(filter_label_max),
            int(bool(add_penalty_short_only_mode))))

    # ---- host entry points -------------------------------------------------------------------------------------
    @staticmethod
    def _scans(scan_lik, scan_beam, scan_beam_origin, origins):
        sl = _np_f32(scan_lik if scan_lik is not None else np.zeros((0, 3)), 3)
        sb = _np_f32(scan_beam if scan_beam is not None else np.zeros((0, 3)), 3)
        so = (np.zeros(len(sb), np.uint32) if scan_beam_origin is None
              else np.ascontiguousarray(scan_beam_origin, dtype=np.uint32))
        og = _np_f32(origins if origins is not None else np.zeros((1, 3)), 3)
        return sl, sb, so, og

    def upload_poses(self, poses):
        poses = _np_f32(poses, 7)
        self._check(self.lib.mcl3dl_hip_upload_poses(self.h, _ptr(poses), len(poses)))
        self._n_uploaded = len(poses)

    def measure_batch(self, poses, scan_lik, scan_beam=None, scan_beam_origin=None, origins=None):
        """poses=None: the set sent by upload_poses()."""
        sl, sb, so, og = self._scans(scan_lik, scan_beam, scan_beam_origin, origins)
        if poses is None:
            n_p = self._n_uploaded
        else:
            poses = _np_f32(poses, 7)
            n_p = len(poses)
        lik = np.zeros(n_p, np.float32)
        ratio = np.zeros(n_p, np.float32)
        beam = np.zeros(n_p, np.float32)
        self._check(self.lib.mcl3dl_hip_measure_batch(self.h, _ptr(poses), n_p, _ptr(sl), len(sl), _ptr(sb), _ptr(so),
                                                      len(sb), _ptr(og), len(og), _ptr(lik), _ptr(ratio), _ptr(beam)))
        return lik, ratio, beam

    def measure_batch_begin(self, poses, scan_lik, scan_beam=None, scan_beam_origin=None, origins=None, slice_particles=0,
                            out=None):
        """mcl3dl_hip_measure_batch_begin: returns the (lik, ratio, beam) arrays the batch fills in particle order
        (`out` = three caller-owned float32 arrays, e.g. host_array()s); measure_batch_wait(i) says how far they are valid."""
        sl, sb, so, og = self._scans(scan_lik, scan_beam, scan_beam_origin, origins)
        if poses is None:
            n_p = self._n_uploaded
        else:
            poses = _np_f32(poses, 7)
            n_p = len(poses)
        lik, ratio, beam = out if out is not None else (np.zeros(n_p, np.float32) for _ in range(3))
        # (a batch still open is ended by this very call and delivers into ITS arrays: they stay alive until it returns)
        previous = getattr(self, "_batch_keep", None)
        self._batch_keep = (poses, sl, sb, so, og, lik, ratio, beam, previous)
        self._check(self.lib.mcl3dl_hip_measure_batch_begin(self.h, _ptr(poses), n_p, _ptr(sl), len(sl), _ptr(sb), _ptr(so),
                    len(sb), _ptr(og), len(og), _ptr(lik), _ptr(ratio), _ptr(beam), int(slice_particles)))
        self._batch_keep = self._batch_keep[:8]
        return lik, ratio, beam

    def measure_batch_wait(self, particle):
        n = C.c_size_t(0)
        self._check(self.lib.mcl3dl_hip_measure_batch_wait(self.h, int(particle), C.byref(n)))
        return n.value

    def measure_batch_end(self):
        self._check(self.lib.mcl3dl_hip_measure_batch_end(self.h))
        self._batch_keep = None

    def pf_measure(self, weights, lik, beam=None, extra=None, match_ratio=None):
        w = _np_f32(weights).copy()
        lk = _np_f32(lik)
        bm = None if beam is None else _np_f32(beam)
        ex = None if extra is None else _np_f32(extra)
        mr = None if match_ratio is None else _np_f32(match_ratio)
        ent, rmin, rmax, rest = C.c_float(0), C.c_float(0), C.c_float(0), C.c_int(0)
        self._check(self.lib.mcl3dl_hip_pf_measure(self.h, _ptr(w), _ptr(lk), _ptr(bm), _ptr(ex), _ptr(mr), len(w),
                                                   C.byref(ent), C.byref(rmin), C.byref(rmax), C.byref(rest)))
        return dict(weights=w, entropy=float(ent.value), match_ratio_min=float(rmin.value),
                    match_ratio_max=float(rmax.value), restored=bool(rest.value))

    def measure_update(self, poses, weights, scan_lik, scan_beam=None, scan_beam_origin=None, origins=None,
                       extra=None):
        poses = _np_f32(poses, 7)
        n_p = len(poses)
        w = _np_f32(weights).copy()
        ex = None if extra is None else _np_f32(extra)
        sl, sb, so, og = self._scans(scan_lik, scan_beam, scan_beam_origin, origins)
        lik = np.zeros(n_p, np.float32)
        ratio = np.zeros(n_p, np.float32)
        beam = np.zeros(n_p, np.float32)
        ent, rmin, rmax, rest = C.c_float(0), C.c_float(0), C.c_float(0), C.c_int(0)
        self._check(self.lib.mcl3dl_hip_measure_update(
            self.h, _ptr(poses), _ptr(ex), _ptr(w), n_p, _ptr(sl), len(sl), _ptr(sb), _ptr(so), len(sb), _ptr(og),
            len(og), _ptr(lik), _ptr(ratio), _ptr(beam), C.byref(ent), C.byref(rmin), C.byref(rmax), C.byref(rest)))
        return dict(weights=w, lik=lik, quality=ratio, beam=beam, entropy=float(ent.value),
                    match_ratio_min=float(rmin.value), match_ratio_max=float(rmax.value), restored=bool(rest.value))

    def host_array(self, shape, dtype=np.float32):
        """A numpy array in page-locked memory of this context (mcl3dl_hip_host_alloc): measure_update reads / writes such
        arrays in place. The block lives as long as the engine (or until host_free(array))."""
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        n_bytes = max(int(np.prod(shape)) * np.dtype(dtype).itemsize, 1)
        out = C.c_void_p()
        self._check(self.lib.mcl3dl_hip_host_alloc(self.h, n_bytes, C.byref(out)))
        buf = (C.c_char * n_bytes).from_address(out.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = out.value
        return arr

    def host_free(self, arr):
        base = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if base is None:
            raise EngineError("host_free: not an array of host_array()")
        self._check(self.lib.mcl3dl_hip_host_free(self.h, C.c_void_p(base)))

    def measure_update_into(self, poses, weights_inout, scan_lik, scan_beam, scan_beam_origin, origins, out_lik, out_ratio,
                            out_beam, extra=None):
        """mcl3dl_hip_measure_update on the caller's own float32 C-contiguous arrays, nothing allocated or converted here
        (weights are updated in place). Returns (entropy, match_ratio_min, match_ratio_max, restored)."""
        ent, rmin, rmax, rest = C.c_float(0), C.c_float(0), C.c_float(0), C.c_int(0)
        n_s = 0 if scan_lik is None else len(scan_lik)
        n_b = 0 if scan_beam is None else len(scan_beam)
        n_o = 0 if origins is None else len(origins)
        self._check(self.lib.mcl3dl_hip_measure_update(
            self.h, _ptr(poses), _ptr(extra), _ptr(weights_inout), len(poses), _ptr(scan_lik), n_s, _ptr(scan_beam),
            _ptr(scan_beam_origin), n_b, _ptr(origins), n_o, _ptr(out_lik), _ptr(out_ratio), _ptr(out_beam), C.byref(ent),
            C.byref(rmin), C.byref(rmax), C.byref(rest)))
        return float(ent.value), float(rmin.value), float(rmax.value), bool(rest.value)

    def time_measure_update(self, poses, w0, w, scan_lik, scan_beam, scan_beam_origin, origins, out_lik, out_ratio, out_beam,
                            steps, warm_ms=100.0, extra=None):
        """`steps` host-buffer updates timed from C (tools/benchloop.c) on the caller's own arrays (float32 / uint32,
        C-contiguous; w is reset from w0 before every step). Returns (mean ms per update, per-step ms)."""
        lib = load_benchloop()
        per = np.zeros(steps, np.float64)
        n_s = 0 if scan_lik is None else len(scan_lik)
        n_b = 0 if scan_beam is None else len(scan_beam)
        n_o = 0 if origins is None else len(origins)
        ms = lib.mcl3dl_benchloop_measure_update(self.h, _ptr(poses), _ptr(extra), _ptr(w0), _ptr(w), len(poses), _ptr(scan_lik),
                                                 n_s, _ptr(scan_beam), _ptr(scan_beam_origin), n_b, _ptr(origins), n_o,
                                                 _ptr(out_lik), _ptr(out_ratio), _ptr(out_beam), int(steps), float(warm_ms),
                                                 _ptr(per))
        if ms < 0:
            self._check(int(ms))
        return float(ms), per

    def beam_status(self, begin, end):
        b = _np_f32(begin, 3)
        e = _np_f32(end, 3)
        st = np.zeros(len(b), np.int32)
        hit = np.zeros(len(b), np.int32)
        self._check(self.lib.mcl3dl_hip_beam_status(self.h, _ptr(b), _ptr(e), len(b), _ptr(st), _ptr(hit)))
        return st, hit

    def expectation(self, poses, weights, bias=None):
        poses = _np_f32(poses, 7)
        w = _np_f32(weights)
        b = None if bias is None else _np_f32(bias)
        mean = np.zeros(7, np.float32)
        total = C.c_float(0)
        im, ib = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.mcl3dl_hip_expectation(self.h, _ptr(poses), _ptr(w), _ptr(b), len(poses), _ptr(mean),
                                                    C.byref(total), C.byref(im), C.byref(ib)))
        return mean, float(total.value), int(im.value), int(ib.value)

    def covariance(self, poses, weights, mean7, subset=None):
        poses = _np_f32(poses, 7)
        w = _np_f32(weights)
        m = _np_f32(mean7)
        sub = None if subset is None else np.ascontiguousarray(subset, dtype=np.uint32)
        cov = np.zeros((6, 6), np.float32)
        self._check(self.lib.mcl3dl_hip_covariance(self.h, _ptr(poses), _ptr(w), len(poses), _ptr(sub),
                                                   0 if sub is None else len(sub), _ptr(m), _ptr(cov)))
        return cov

    def expectation_device(self, d_pose, d_weight, d_bias, n):
        mean = np.zeros(7, np.float32)
        total = C.c_float(0)
        im, ib = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.mcl3dl_hip_expectation_device(self.h, _ptr(d_pose), _ptr(d_weight), _ptr(d_bias), n,
                                                           _ptr(mean), C.byref(total), C.byref(im), C.byref(ib)))
        return mean, float(total.value), int(im.value), int(ib.value)

    def covariance_device(self, d_pose, d_weight, n, mean7, d_subset=None, n_subset=0):
        m = _np_f32(mean7)
        cov = np.zeros((6, 6), np.float32)
        self._check(self.lib.mcl3dl_hip_covariance_device(self.h, _ptr(d_pose), _ptr(d_weight), n, _ptr(d_subset),
                                                          n_subset, _ptr(m), _ptr(cov)))
        return cov

    def moments_partial_device(self, d_pose, d_weight, d_bias, n, d_out16):
        self._check(self.lib.mcl3dl_hip_moments_partial_device(self.h, _ptr(d_pose), _ptr(d_weight), _ptr(d_bias), n,
                                                               _ptr(d_out16)))

    def moments_finish(self, parts16, index_offset=None):
        parts = np.ascontiguousarray(parts16, np.float64).reshape(-1, 16)
        off = None if index_offset is None else np.ascontiguousarray(index_offset, np.uint64)
        mean = np.zeros(7, np.float32)
        total, im, ib = C.c_float(0), C.c_int64(0), C.c_int64(0)
        rc = self.lib.mcl3dl_hip_moments_finish(_ptr(parts), len(parts), _ptr(off), _ptr(mean), C.byref(total),
                                                C.byref(im), C.byref(ib))
        if rc != 0:
            raise EngineError("moments_finish: bad arguments")
        return mean, float(total.value), int(im.value), int(ib.value)

    def covariance_partial_device(self, d_pose, d_weight, n, mean7, d_out22, d_subset=None, n_subset=0):
        m = _np_f32(mean7)
        self._check(self.lib.mcl3dl_hip_covariance_partial_device(self.h, _ptr(d_pose), _ptr(d_weight), n,
                                                                  _ptr(d_subset), n_subset, _ptr(m), _ptr(d_out22)))

    def covariance_finish(self, sums22):
        s = np.ascontiguousarray(sums22, np.float64)
        cov = np.zeros((6, 6), np.float32)
        if self.lib.mcl3dl_hip_covariance_finish(_ptr(s), _ptr(cov)) != 0:
            raise EngineError("covariance_finish: bad arguments")
        return cov

    def resample_begin(self, weights, n_out=None):
        w = _np_f32(weights)
        pstep = C.c_float(0)
        self._check(self.lib.mcl3dl_hip_resample_begin(self.h, _ptr(w), len(w), len(w) if n_out is None else n_out,
                                                       C.byref(pstep)))
        self._rs_n_out = len(w) if n_out is None else n_out
        return float(pstep.value)

    def resample_plan(self, mode, initial_p=0.0, want_plan=True):
        """want_plan=False leaves source / duplicate flags on the device (returns (None, None, n_duplicates))."""
        nd = C.c_size_t(0)
        if not want_plan:
            self._check(self.lib.mcl3dl_hip_resample_plan(self.h, int(mode), float(initial_p), None, None, C.byref(nd)))
            return None, None, int(nd.value)
        src = np.zeros(self._rs_n_out, np.uint32)
        dup = np.zeros(self._rs_n_out, np.uint8)
        self._check(self.lib.mcl3dl_hip_resample_plan(self.h, int(mode), float(initial_p), _ptr(src), _ptr(dup),
                                                      C.byref(nd)))
        return src, dup, int(nd.value)

    def resample_apply(self, state13, noise13=None):
        s = _np_f32(state13, 13)
        nz = None if noise13 is None or len(noise13) == 0 else _np_f32(noise13, 13)
        out = np.zeros((self._rs_n_out, 13), np.float32)
        self._check(self.lib.mcl3dl_hip_resample_apply(self.h, _ptr(s), _ptr(nz), 0 if nz is None else len(nz),
                                                       _ptr(out)))
        return out

    def resample_begin_device(self, d_weight, n, n_out=None):
        pstep = C.c_float(0)
        self._rs_n_out = n if n_out is None else n_out
        self._check(self.lib.mcl3dl_hip_resample_begin_device(self.h, _ptr(d_weight), n, self._rs_n_out,
                                                              C.byref(pstep)))
        return float(pstep.value)

    def resample_apply_device(self, d_state13_in, noise13, d_state13_out, out_begin=0, out_count=None):
        """Gather (+ noise on duplicated slots) into d_state13_out; a slice of the planned slots when out_count is set."""
        nz = None if noise13 is None or len(noise13) == 0 else _np_f32(noise13, 13)
        cnt = self._rs_n_out - out_begin if out_count is None else out_count
        self._check(self.lib.mcl3dl_hip_resample_apply_slice_device(self.h, _ptr(d_state13_in), _ptr(nz),
                                                                    0 if nz is None else len(nz), out_begin, cnt,
                                                                    _ptr(d_state13_out)))

    def radius_search(self, query_xyz, radius):
        q = _np_f32(query_xyz, 3)
        idx = np.zeros(len(q), np.int32)
        sq = np.zeros(len(q), np.float32)
        self._check(self.lib.mcl3dl_hip_radius_search(self.h, _ptr(q), len(q), float(radius), _ptr(idx), _ptr(sq)))
        return idx, sq

    def dda_trace(self, begin, end, max_out=4096):
        b = _np_f32(begin)
        e = _np_f32(end)
        out = np.zeros((max_out, 3), np.float32)
        n, col, hit = C.c_int(0), C.c_int(0), C.c_int(-1)
        self._check(self.lib.mcl3dl_hip_dda_trace(self.h, _ptr(b), _ptr(e), _ptr(out), max_out, C.byref(n),
                                                  C.byref(col), C.byref(hit)))
        return out[:min(n.value, max_out)].copy(), bool(col.value), int(hit.value), int(n.value)

    # ---- scan preparation / map path on the device ----------------------------------------
    @staticmethod
    def _f3(a):
        return None if a is None else _np_f32(a)

    def scan_begin(self, xyz, label=None, leaf=None, clip_lik=(0.5, 10.0, -2.0, 2.0), clip_beam=(0.5, 4.0, -2.0, 2.0)):
        """VoxelGrid + both clip filters on the device; returns (n_full, n_lik_clipped, n_beam_clipped)."""
        pts = _np_f32(xyz, 3)
        lab = None if label is None else np.ascontiguousarray(label, dtype=np.uint32)
        lf, cl, cb = self._f3(leaf), self._f3(clip_lik), self._f3(clip_beam)
        a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self._check(self.lib.mcl3dl_hip_scan_begin(self.h, _ptr(pts), _ptr(lab), len(pts), _ptr(lf), _ptr(cl), _ptr(cb),
                                                   C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def scan_begin_pointcloud2(self, data, n_points, point_step, off_x, off_y, off_z, off_label=-1, label_override=0,
                               leaf=None, clip_lik=(0.5, 10.0, -2.0, 2.0), clip_beam=(0.5, 4.0, -2.0, 2.0)):
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        lf, cl, cb = self._f3(leaf), self._f3(clip_lik), self._f3(clip_beam)
        a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self._check(self.lib.mcl3dl_hip_scan_begin_pointcloud2(self.h, _ptr(buf), n_points, point_step, off_x, off_y, off_z,
                                                               off_label, label_override, _ptr(lf), _ptr(cl), _ptr(cb),
                                                               C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def scan_finish(self, idx_lik, idx_beam=None, origins=None):
        il = np.ascontiguousarray(idx_lik if idx_lik is not None else [], dtype=np.uint32)
        ib = np.ascontiguousarray(idx_beam if idx_beam is not None else [], dtype=np.uint32)
        og = _np_f32(origins if origins is not None else np.zeros((1, 3)), 3)
        self._check(self.lib.mcl3dl_hip_scan_finish(self.h, _ptr(il), len(il), _ptr(ib), len(ib), _ptr(og), len(og)))

    def sort_pairs(self, keys, vals=None, end_bit=32):
        """The device radix sort of the cloud path on its own (stable, ascending by key bits [0, end_bit))."""
        k = np.ascontiguousarray(keys, dtype=np.uint32)
        v = None if vals is None else np.ascontiguousarray(vals, dtype=np.uint32)
        ok, ov = np.empty_like(k), np.empty_like(k)
        self._check(self.lib.mcl3dl_hip_sort_pairs(self.h, _ptr(k), _ptr(v), len(k), int(end_bit), _ptr(ok), _ptr(ov)))
        return ok, ov

    def scan_download(self, which):
        n = C.c_size_t(0)
        self._check(self.lib.mcl3dl_hip_scan_download(self.h, which, None, None, 0, C.byref(n)))
        xyz = np.zeros((n.value, 3), np.float32)
        lab = np.zeros(n.value, np.uint32)
        if n.value:
            self._check(self.lib.mcl3dl_hip_scan_download(self.h, which, _ptr(xyz), _ptr(lab), n.value, C.byref(n)))
        return xyz, lab

    def set_map_downsampled(self, xyz, label=None, leaf=(0.1, 0.1, 0.1), stamp=1, dist_weight=(1.0, 1.0, 1.0)):
        pts = _np_f32(xyz, 3)
        lab = None if label is None else np.ascontiguousarray(label, dtype=np.uint32)
        n = C.c_size_t(0)
        self._check(self.lib.mcl3dl_hip_set_map_downsampled(self.h, _ptr(pts), _ptr(lab), len(pts), _ptr(self._f3(leaf)),
                                                            int(stamp), _ptr(self._f3(dist_weight)), C.byref(n)))
        return int(n.value)

    def set_map_pointcloud2(self, data, n_points, point_step, off_x, off_y, off_z, off_label=-1, leaf=(0.1, 0.1, 0.1),
                            stamp=1, dist_weight=(1.0, 1.0, 1.0)):
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        n = C.c_size_t(0)
        self._check(self.lib.mcl3dl_hip_set_map_pointcloud2(self.h, _ptr(buf), n_points, point_step, off_x, off_y, off_z,
                                                            off_label, _ptr(self._f3(leaf)), int(stamp),
                                                            _ptr(self._f3(dist_weight)), C.byref(n)))
        return int(n.value)

    @staticmethod
    def _update_stats(st):
        return dict(bricks_recompiled=int(st[0]), bricks_added=int(st[1]), points_involved=int(st[2]),
                    device_ms=float(st[3]), overflow_appended=int(st[4]), outcome=int(st[5]))

    def map_update(self, xyz, label=None, leaf=(0.2, 0.2, 0.2), stamp=2):
        """pc_map2 = pc_map + VoxelGrid(update); returns (n_map, stats dict)."""
        pts = _np_f32(xyz if xyz is not None else np.zeros((0, 3)), 3)
        lab = None if label is None else np.ascontiguousarray(label, dtype=np.uint32)
        n = C.c_size_t(0)
        st = np.zeros(6, np.float64)
        self._check(self.lib.mcl3dl_hip_map_update(self.h, _ptr(pts), _ptr(lab), len(pts), _ptr(self._f3(leaf)), int(stamp),
                                                   C.byref(n), _ptr(st)))
        return int(n.value), self._update_stats(st)

    def map_update_pointcloud2(self, data, n_points, point_step, off_x, off_y, off_z, off_label=-1, leaf=(0.2, 0.2, 0.2),
                               stamp=2):
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        n = C.c_size_t(0)
        st = np.zeros(6, np.float64)
        self._check(self.lib.mcl3dl_hip_map_update_pointcloud2(self.h, _ptr(buf), n_points, point_step, off_x, off_y, off_z,
                                                               off_label, _ptr(self._f3(leaf)), int(stamp), C.byref(n),
                                                               _ptr(st)))
        return int(n.value), self._update_stats(st)

    def map_download(self):
        n = C.c_size_t(0)
        self._check(self.lib.mcl3dl_hip_map_download(self.h, None, None, 0, C.byref(n)))
        xyz = np.zeros((n.value, 3), np.float32)
        lab = np.zeros(n.value, np.uint32)
        self._check(self.lib.mcl3dl_hip_map_download(self.h, _ptr(xyz), _ptr(lab), n.value, C.byref(n)))
        return xyz, lab

    def match_split(self, pose7, xyz=None, unmatch_dist=0.5, match_dist=0.1):
        """Transformed points classified as (matched, unmatched); xyz=None uses the cloud scan_begin left on the device."""
        pose = _np_f32(pose7)
        pts = None if xyz is None else _np_f32(xyz, 3)
        if pts is None:
            cnt = C.c_size_t(0)
            self._check(self.lib.mcl3dl_hip_scan_download(self.h, 0, None, None, 0, C.byref(cnt)))
            n, cap = 0, cnt.value
        else:
            n = cap = len(pts)
        # one call: every point lands in exactly one of the two clouds, so `cap` points of room in each always suffice
        nm, nu = C.c_size_t(0), C.c_size_t(0)
        m = np.zeros((max(cap, 1), 3), np.float32)
        u = np.zeros((max(cap, 1), 3), np.float32)
        self._check(self.lib.mcl3dl_hip_match_split(self.h, _ptr(pose), _ptr(pts), n, unmatch_dist, match_dist, _ptr(m),
                                                    cap, C.byref(nm), _ptr(u), cap, C.byref(nu)))
        return m[:nm.value].copy(), u[:nu.value].copy()

    def match_split_into(self, pose7, out_matched, out_unmatched, xyz=None, unmatch_dist=0.5, match_dist=0.1):
        """mcl3dl_hip_match_split into the caller's own float32 (n, 3) arrays (e.g. host_array()s: written in place by the
        kernel), nothing allocated here; either may be None (count only). Returns (n_matched, n_unmatched)."""
        pose = _np_f32(pose7)
        pts = None if xyz is None else _np_f32(xyz, 3)
        nm, nu = C.c_size_t(0), C.c_size_t(0)
        self._check(self.lib.mcl3dl_hip_match_split(
            self.h, _ptr(pose), _ptr(pts), 0 if pts is None else len(pts), unmatch_dist, match_dist, _ptr(out_matched),
            0 if out_matched is None else len(out_matched), C.byref(nm), _ptr(out_unmatched),
            0 if out_unmatched is None else len(out_unmatched), C.byref(nu)))
        return nm.value, nu.value

    # ---- device entry points (torch CUDA tensors or raw device addresses) ---------------------------------------
    def upload_scan(self, scan_lik, scan_beam=None, scan_beam_origin=None, origins=None):
        sl, sb, so, og = self._scans(scan_lik, scan_beam, scan_beam_origin, origins)
        self._check(self.lib.mcl3dl_hip_upload_scan(self.h, _ptr(sl), len(sl), _ptr(sb), _ptr(so), len(sb), _ptr(og),
                                                    len(og)))

    def scan_order(self, n_s):
        """order[k] = index (in the caller's array) of the point the engine holds at position k of the likelihood scan."""
        order = np.zeros(n_s, np.uint32)
        self._check(self.lib.mcl3dl_hip_scan_order(self.h, _ptr(order), n_s))
        return order

    def measure_device(self, d_pose, n_p, d_lik, d_ratio, d_beam):
        self._check(self.lib.mcl3dl_hip_measure_device(self.h, _ptr(d_pose), n_p, _ptr(d_lik), _ptr(d_ratio),
                                                       _ptr(d_beam)))

    def update_device(self, d_pose, n_p, d_weight, d_stats4, d_extra=None, d_lik=None, d_ratio=None, d_beam=None):
        """measure_device + pf_partial_device + pf_apply_device in one call (hipGraph replay from the third call on)."""
        self._check(self.lib.mcl3dl_hip_update_device(self.h, _ptr(d_pose), n_p, _ptr(d_weight), _ptr(d_extra),
                                                      _ptr(d_lik), _ptr(d_ratio), _ptr(d_beam), _ptr(d_stats4)))

    def pf_partial_device(self, d_weight, d_lik, d_beam, d_extra, d_ratio, n_p, d_packed, rank=0, world=1):
        """d_packed: 2 + 2*world float64 on the device, ready for all_reduce(SUM) (see mcl_3dl_amd/distributed.py)."""
        self._check(self.lib.mcl3dl_hip_pf_partial_device(self.h, _ptr(d_weight), _ptr(d_lik), _ptr(d_beam),
                                                          _ptr(d_extra), _ptr(d_ratio), n_p, int(rank), int(world),
                                                          _ptr(d_packed)))

    def pf_apply_device(self, d_weight, n_p, d_packed, d_stats4, world=1):
        self._check(self.lib.mcl3dl_hip_pf_apply_device(self.h, _ptr(d_weight), n_p, int(world), _ptr(d_packed),
                                                        _ptr(d_stats4)))

    # ---- measurement support -----------------------------------------------------------------------------------
    def set_kernel_timing(self, enable):
        self._check(self.lib.mcl3dl_hip_set_kernel_timing(self.h, int(bool(enable))))

    def reset_kernel_time(self):
        self._check(self.lib.mcl3dl_hip_reset_kernel_time(self.h))

    def kernel_time(self, kernel_id):
        ms, n = C.c_double(0), C.c_uint64(0)
        self._check(self.lib.mcl3dl_hip_get_kernel_time(self.h, kernel_id, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def workload_stats(self, d_pose, n_p):
        st = np.zeros(6, np.float64)
        self._check(self.lib.mcl3dl_hip_workload_stats(self.h, _ptr(d_pose), n_p, _ptr(st)))
        return dict(sum_k=st[0], evals=st[1], dda_steps=st[2], dda_occupied=st[3], dda_tested=st[4], rays=st[5])

    def memory_footprint(self):
        b = np.zeros(8, np.uint64)
        self._check(self.lib.mcl3dl_hip_memory_footprint(self.h, _ptr(b)))
        return dict(lik_points=int(b[0]), lik_cells=int(b[1]), dda_bits=int(b[2]), dda_voxels=int(b[3]),
                    dda_points=int(b[4]), cand_table=int(b[5]), cand_start=int(b[6]), cand_points=int(b[7]))

    def set_option(self, name, value):
        self._check(self.lib.mcl3dl_hip_set_option(self.h, name.encode(), float(value)))

    def get_option(self, name):
        v = C.c_double(0)
        self._check(self.lib.mcl3dl_hip_get_option(self.h, name.encode(), C.byref(v)))
        return float(v.value)

    def index_stats(self):
        s = np.zeros(8, np.float64)
        self._check(self.lib.mcl3dl_hip_index_stats(self.h, _ptr(s)))
        return dict(bricks=int(s[0]), preliminary=int(s[1]), candidates=int(s[2]), build_ms=float(s[3]),
                    voxels_with_candidates=int(s[4]), voxels_with_overflow=int(s[5]), overflow_records=int(s[6]),
                    voxel_ratio=float(s[7]), voxels_over8=int(self.get_option("cand_voxels_over8")),
                    record_parts=int(self.get_option("cand_record_parts_in_use")),
                    packed_words=int(self.get_option("cand_packed_active")),
                    deferred_overflow=int(self.get_option("lik_defer_active")))
