"""ctypes binding of include/kssicp.h (libkssicp.so).

Host pointers are numpy arrays; device pointers are integers (e.g. torch `tensor.data_ptr()`).
No CPU fallback: a missing library or GPU raises KssError.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
NSUMS = 20
K_NN_SWEEP, K_CORR_REDUCE, K_PRESHAPE, K_ROT_SEARCH, K_POSE_APPLY, K_GRID_NN, K_GRID_BUILD, K_GRID_CHAIN, K_GRID_CHAIN_PASS, K_RESIDENT, K_RESIDENT_PASS = range(11)
NN_AUTO, NN_BRUTE, NN_GRID = 0, 1, 2
F32, F64 = 0, 1

# every symbol include/kssicp.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "kss_version", "kss_status_string", "kss_last_error", "kss_ctx_create", "kss_ctx_create_on_stream",
    "kss_ctx_destroy", "kss_ctx_synchronize", "kss_ctx_stream", "kss_ctx_set_nn_mode", "kss_profile_enable", "kss_profile_reset",
    "kss_profile_get", "kss_profile_event_overhead", "kss_grid_stats", "kss_preshape_stats", "kss_preshape_stats_dev", "kss_preshape_stats_pair_dev", "kss_pose_apply", "kss_pose_apply_dev",
    "kss_nn", "kss_nn_dev", "kss_cov", "kss_cov_dev", "kss_rigid_from_sums", "kss_rotation_search",
    "kss_rotation_search_dev", "kss_grid_angles", "kss_rotation_candidates", "kss_icp_default_params", "kss_icp",
    "kss_icp_dev", "kss_icp_batch", "kss_icp_batch_dev", "kss_transform_apply", "kss_transform_apply_dev",
    "kss_pcr_qm", "kss_register", "kss_register_batch", "kss_gather_results", "kss_rccl_allreduce_sum", "kss_transform_apply_f32", "kss_downsample_fps", "kss_downsample_aivs", "kss_downsample_aivs_pair", "kss_downsample_octree", "kss_knn", "kss_knn_dev", "kss_normals", "kss_normals_orient",
]


class KssError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = "%s failed: status %d" % (where, status)
        if detail:
            msg += " (%s)" % detail
        super().__init__(msg)


# kss_allreduce_fn: in-place sum over ranks of n doubles in host memory (kss_icp_params.allreduce)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


class RcclLink(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rccl_comm", C.c_void_p)]


class IcpParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("max_corr_dist", C.c_double),
                ("transformation_epsilon", C.c_double), ("euclidean_fitness_epsilon", C.c_double),
                ("abs_mse_epsilon", C.c_double), ("min_correspondences", C.c_int),
                ("fixed_iterations", C.c_int), ("nn_fma", C.c_int), ("compute_fitness", C.c_int),
                ("nn_sources_per_thread", C.c_int), ("nn_target_splits", C.c_int), ("nn_mode", C.c_int),
                ("trace_sums", C.POINTER(C.c_double)), ("trace_Tk", C.POINTER(C.c_float)),
                ("trace_cap", C.c_int), ("trace_n", C.POINTER(C.c_int)),
                ("fitness_idx", C.POINTER(C.c_int32)), ("fitness_d2", C.POINTER(C.c_float)),
                ("allreduce", ALLREDUCE_FN), ("allreduce_user", C.c_void_p)]


class IcpResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("fitness", C.c_double), ("last_mse", C.c_double),
                ("iterations", C.c_int32), ("converged", C.c_int32), ("state", C.c_int32), ("pair_id", C.c_int32)]

    def matrix(self):
        return np.array(self.T, dtype=np.float32).reshape(4, 4)


class Pose(C.Structure):
    _fields_ = [("shift", C.c_double * 3), ("center", C.c_double * 3), ("scale", C.c_double), ("angle", C.c_double * 3)]


class RegisterResult(C.Structure):
    _fields_ = [("scale", C.c_double), ("angle", C.c_double * 3), ("R", C.c_double * 9), ("t", C.c_double * 3),
                ("c_src", C.c_double * 3), ("c_tgt", C.c_double * 3), ("T_icp", C.c_float * 16), ("E_d_init", C.c_double), ("final_fitness", C.c_double),
                ("used_angle_list", C.c_int32), ("angle_index", C.c_int32), ("n_angle_list", C.c_int32),
                ("icp_iterations", C.c_int32), ("icp_converged", C.c_int32), ("grid", C.c_int32)]


def lib_path():
    return os.path.join(_HERE, "lib", "libkssicp.so")


def build_library(force=False):
    """Compile libkssicp.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE, "-j4"], stdout=subprocess.DEVNULL)
    return lib_path()


_LIB = None


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise KssError(-4, "load_library", "libkssicp.so not built: run __graft_entry__.build() (no CPU fallback exists)")
    # One HIP runtime per process: PyTorch ships its own libamdhip64.so; if libkssicp.so pulled in /opt/rocm's copy
    # first, torch would later load a second runtime and see no devices.  Importing torch first (when it is installed)
    # makes both resolve to the same library.  KSS_NO_TORCH=1 skips this for hosts that never use torch.
    if "torch" not in sys.modules and not os.environ.get("KSS_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(p)
    vp, i64, dbl = C.c_void_p, C.c_int64, C.c_double
    L.kss_version.restype = C.c_int
    L.kss_status_string.restype = C.c_char_p
    L.kss_status_string.argtypes = [C.c_int]
    L.kss_last_error.restype = C.c_char_p
    L.kss_last_error.argtypes = [vp]
    L.kss_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.kss_ctx_create_on_stream.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.kss_ctx_destroy.argtypes = [vp]
    L.kss_ctx_synchronize.argtypes = [vp]
    L.kss_ctx_stream.restype = vp
    L.kss_ctx_stream.argtypes = [vp]
    L.kss_ctx_set_nn_mode.argtypes = [vp, C.c_int]
    L.kss_profile_enable.argtypes = [vp, C.c_int]
    L.kss_profile_reset.argtypes = [vp]
    L.kss_grid_stats.argtypes = [vp, vp]
    L.kss_profile_get.argtypes = [vp, C.c_int, C.POINTER(dbl), C.POINTER(i64)]
    L.kss_profile_event_overhead.argtypes = [vp, C.POINTER(dbl)]
    L.kss_downsample_octree.argtypes = [vp, vp, i64, vp, i64, C.POINTER(i64), C.POINTER(dbl)]
    L.kss_normals_orient.argtypes = [vp, vp, i64, vp]
    L.kss_register_batch.argtypes = [vp, vp, vp, vp, vp, C.c_int, i64, dbl, C.c_int, C.c_int, vp, vp]
    for n in ("kss_preshape_stats", "kss_preshape_stats_dev"):
        getattr(L, n).argtypes = [vp, vp, C.c_int, i64, vp, C.POINTER(dbl)]
    L.kss_preshape_stats_pair_dev.argtypes = [vp, vp, i64, vp, i64, C.c_int, vp, C.POINTER(dbl), vp, C.POINTER(dbl)]
    for n in ("kss_pose_apply", "kss_pose_apply_dev"):
        getattr(L, n).argtypes = [vp, vp, i64, C.POINTER(Pose), vp]
    for n in ("kss_nn", "kss_nn_dev"):
        getattr(L, n).argtypes = [vp, vp, i64, vp, i64, vp, vp]
    for n in ("kss_cov", "kss_cov_dev"):
        getattr(L, n).argtypes = [vp, vp, vp, vp, i64, i64, dbl, vp]
    L.kss_rigid_from_sums.argtypes = [vp, vp]
    for n in ("kss_rotation_search", "kss_rotation_search_dev"):
        getattr(L, n).argtypes = [vp, vp, i64, vp, i64, dbl, vp, i64, C.POINTER(C.c_int)]
    L.kss_grid_angles.argtypes = [dbl, vp, C.c_int]
    L.kss_rotation_candidates.argtypes = [vp, C.c_int, dbl, vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.kss_icp_default_params.argtypes = [C.POINTER(IcpParams)]
    for n in ("kss_icp", "kss_icp_dev"):
        getattr(L, n).argtypes = [vp, vp, i64, vp, i64, C.POINTER(IcpParams), C.POINTER(IcpResult)]
    for n in ("kss_icp_batch", "kss_icp_batch_dev"):
        getattr(L, n).argtypes = [vp, vp, vp, vp, vp, C.c_int, C.POINTER(IcpParams), vp]
    for n in ("kss_transform_apply", "kss_transform_apply_dev"):
        getattr(L, n).argtypes = [vp, vp, vp, i64, vp]
    L.kss_pcr_qm.argtypes = [vp, vp, i64, vp, i64, vp]
    L.kss_transform_apply_f32.argtypes = [vp, vp, vp, i64, vp]
    L.kss_downsample_fps.argtypes = [vp, vp, i64, i64, vp, vp]
    L.kss_downsample_aivs.argtypes = [vp, vp, i64, i64, vp, i64, C.POINTER(i64), vp]
    L.kss_downsample_aivs_pair.argtypes = [vp, vp, i64, i64, vp, i64, C.POINTER(i64), vp, vp, i64, i64, vp, i64, C.POINTER(i64), vp, C.POINTER(C.c_int)]
    for n in ("kss_knn", "kss_knn_dev"):
        getattr(L, n).argtypes = [vp, vp, i64, vp, i64, C.c_int, vp, vp]
    L.kss_normals.argtypes = [vp, vp, i64, C.c_int, vp]
    L.kss_register.argtypes = [vp, vp, i64, vp, i64, vp, i64, dbl, C.c_int, vp, C.POINTER(RegisterResult)]
    L.kss_gather_results.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp]
    _LIB = L
    return L


def exported_symbols():
    L = load_library()
    return [s for s in SYMBOLS if hasattr(L, s)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 3)


# ---- host-only helpers (no GPU needed) ---------------------------------------------------------------
def grid_angles(step):
    L = load_library()
    buf = np.empty(256, np.float64)
    g = L.kss_grid_angles(float(step), _p(buf), 256)
    if g < 0:
        raise KssError(g, "kss_grid_angles")
    return buf[:g].copy()


def rotation_candidates(err, step):
    L = load_library()
    err = np.ascontiguousarray(err, dtype=np.float64)
    g = err.shape[0]
    best = np.empty(3, np.float64)
    alist = np.empty(3 * g ** 3, np.float64)
    nl = C.c_int(0)
    rc = L.kss_rotation_candidates(_p(err), g, float(step), _p(best), _p(alist), g ** 3, C.byref(nl))
    if rc != 0:
        raise KssError(rc, "kss_rotation_candidates")
    return best, alist[:3 * nl.value].reshape(-1, 3).copy()


def rigid_from_sums(sums):
    L = load_library()
    s = np.ascontiguousarray(sums, dtype=np.float64)
    T = np.empty(16, np.float32)
    rc = L.kss_rigid_from_sums(_p(s), _p(T))
    if rc != 0:
        raise KssError(rc, "kss_rigid_from_sums")
    return T.reshape(4, 4)


# ---- context --------------------------------------------------------------------------------------------
class Context:
    """One kss_ctx (one GPU, one stream)."""

    def __init__(self, device=0, stream=None):
        self.L = load_library()
        self.h = C.c_void_p()
        if stream is None:
            rc = self.L.kss_ctx_create(int(device), C.byref(self.h))
        else:
            rc = self.L.kss_ctx_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise KssError(rc, "kss_ctx_create", self.L.kss_status_string(rc).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.kss_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where):
        if rc != 0:
            raise KssError(rc, where, self.L.kss_last_error(self.h).decode() or self.L.kss_status_string(rc).decode())

    def set_nn_mode(self, mode):
        self._chk(self.L.kss_ctx_set_nn_mode(self.h, int(mode)), "kss_ctx_set_nn_mode")

    def synchronize(self):
        self._chk(self.L.kss_ctx_synchronize(self.h), "kss_ctx_synchronize")

    # ---- profiling
    def profile_enable(self, on=True):
        """False/0 = off, True/1 = time every launch, n > 1 = time every n-th launch of each kernel class."""
        self._chk(self.L.kss_profile_enable(self.h, int(on)), "kss_profile_enable")

    def profile_reset(self):
        self._chk(self.L.kss_profile_reset(self.h), "kss_profile_reset")

    def profile_event_overhead(self):
        ms = C.c_double(0)
        self._chk(self.L.kss_profile_event_overhead(self.h, C.byref(ms)), "kss_profile_event_overhead")
        return ms.value

    def profile_get(self, k):
        ms, n = C.c_double(0), C.c_int64(0)
        self._chk(self.L.kss_profile_get(self.h, k, C.byref(ms), C.byref(n)), "kss_profile_get")
        return ms.value, n.value

    def grid_stats(self):
        out = np.zeros(8, np.float64)
        self._chk(self.L.kss_grid_stats(self.h, _p(out)), "kss_grid_stats")
        return dict(zip(["h", "gx", "gy", "gz", "occupied_cells", "evaluations_per_pass", "n_src", "n_tgt"], out.tolist()))

    # ---- (a2)
    def preshape_stats(self, xyz):
        a = np.ascontiguousarray(xyz)
        if a.dtype == np.float32:
            a, dt = _f32(a), F32
        else:
            a, dt = _f64(a), F64
        c = np.empty(3, np.float64)
        r = C.c_double(0)
        self._chk(self.L.kss_preshape_stats(self.h, _p(a), dt, len(a), _p(c), C.byref(r)), "kss_preshape_stats")
        return c, r.value

    def preshape_stats_dev(self, dptr, dtype, n):
        c = np.empty(3, np.float64)
        r = C.c_double(0)
        self._chk(self.L.kss_preshape_stats_dev(self.h, C.c_void_p(int(dptr)), dtype, int(n), _p(c), C.byref(r)), "kss_preshape_stats_dev")
        return c, r.value

    def preshape_stats_pair_dev(self, d_src, ns, d_tgt, nt, dtype):
        cs, ct = np.empty(3, np.float64), np.empty(3, np.float64)
        rs, rt = C.c_double(0), C.c_double(0)
        self._chk(self.L.kss_preshape_stats_pair_dev(self.h, C.c_void_p(int(d_src)), int(ns), C.c_void_p(int(d_tgt)), int(nt), dtype,
                                                     _p(cs), C.byref(rs), _p(ct), C.byref(rt)), "kss_preshape_stats_pair_dev")
        return (cs, rs.value), (ct, rt.value)

    # ---- (a7)
    @staticmethod
    def make_pose(shift, center, scale, angle):
        p = Pose()
        for k in range(3):
            p.shift[k] = float(shift[k]); p.center[k] = float(center[k]); p.angle[k] = float(angle[k])
        p.scale = float(scale)
        return p

    def pose_apply(self, pts, pose):
        a = _f64(pts)
        out = np.empty_like(a)
        self._chk(self.L.kss_pose_apply(self.h, _p(a), len(a), C.byref(pose), _p(out)), "kss_pose_apply")
        return out

    def pose_apply_dev(self, d_in, n, pose, d_out):
        self._chk(self.L.kss_pose_apply_dev(self.h, C.c_void_p(int(d_in)), int(n), C.byref(pose), C.c_void_p(int(d_out))), "kss_pose_apply_dev")

    def transform_apply(self, T, pts):
        a = _f64(pts)
        Tm = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
        out = np.empty_like(a)
        self._chk(self.L.kss_transform_apply(self.h, _p(Tm), _p(a), len(a), _p(out)), "kss_transform_apply")
        return out

    # ---- (a8)
    def nn(self, src, tgt):
        s, t = _f32(src), _f32(tgt)
        idx = np.empty(len(s), np.int32)
        d2 = np.empty(len(s), np.float32)
        self._chk(self.L.kss_nn(self.h, _p(s), len(s), _p(t), len(t), _p(idx), _p(d2)), "kss_nn")
        return idx, d2

    def nn_dev(self, d_src, ns, d_tgt, nt, d_idx, d_d2):
        self._chk(self.L.kss_nn_dev(self.h, C.c_void_p(int(d_src)), int(ns), C.c_void_p(int(d_tgt)), int(nt),
                                    C.c_void_p(int(d_idx)) if d_idx else None, C.c_void_p(int(d_d2)) if d_d2 else None), "kss_nn_dev")

    # ---- (a10)
    def cov(self, src, tgt, idx, max_d2=1.0):
        s, t = _f32(src), _f32(tgt)
        i = np.ascontiguousarray(idx, dtype=np.int32)
        sums = np.empty(NSUMS, np.float64)
        self._chk(self.L.kss_cov(self.h, _p(s), _p(t), _p(i), len(s), len(t), float(max_d2), _p(sums)), "kss_cov")
        return sums

    # ---- (a4)
    def rotation_search(self, src_preshaped, tgt, step):
        s, t = _f64(src_preshaped), _f64(tgt)
        err = np.empty(40 ** 3, np.float64)
        g = C.c_int(0)
        self._chk(self.L.kss_rotation_search(self.h, _p(s), len(s), _p(t), len(t), float(step), _p(err), err.size, C.byref(g)), "kss_rotation_search")
        g = g.value
        return err[:g ** 3].reshape(g, g, g).copy()

    # ---- (a9)
    def icp_params(self, **kw):
        p = IcpParams()
        self.L.kss_icp_default_params(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        return p

    def icp(self, src, tgt, params=None, trace_cap=0, fitness_corr=False):
        s, t = _f32(src), _f32(tgt)
        p = params if params is not None else self.icp_params()
        res = IcpResult()
        tr = None
        fc = None
        if fitness_corr:
            fc = (np.full(len(s), -1, np.int32), np.full(len(s), np.nan, np.float32))
            p.fitness_idx = fc[0].ctypes.data_as(C.POINTER(C.c_int32))
            p.fitness_d2 = fc[1].ctypes.data_as(C.POINTER(C.c_float))
        if trace_cap > 0:
            sums = np.zeros((trace_cap, NSUMS), np.float64)
            tk = np.zeros((trace_cap, 16), np.float32)
            n = C.c_int(0)
            p.trace_sums = sums.ctypes.data_as(C.POINTER(C.c_double))
            p.trace_Tk = tk.ctypes.data_as(C.POINTER(C.c_float))
            p.trace_cap = trace_cap
            p.trace_n = C.pointer(n)
            tr = (sums, tk, n)
        self._chk(self.L.kss_icp(self.h, _p(s), len(s), _p(t), len(t), C.byref(p), C.byref(res)), "kss_icp")
        out = {"T": res.matrix(), "iterations": res.iterations, "converged": bool(res.converged),
               "state": res.state, "fitness": res.fitness, "last_mse": res.last_mse}
        if tr:
            out["trace_sums"] = tr[0][:tr[2].value].copy()
            out["trace_Tk"] = tr[1][:tr[2].value].reshape(-1, 4, 4).copy()
            p.trace_sums = None; p.trace_Tk = None; p.trace_cap = 0; p.trace_n = None
        if fc:
            out["fitness_idx"], out["fitness_d2"] = fc
            p.fitness_idx = None; p.fitness_d2 = None
        return out

    def icp_dev(self, d_src, ns, d_tgt, nt, params):
        res = IcpResult()
        self._chk(self.L.kss_icp_dev(self.h, C.c_void_p(int(d_src)), int(ns), C.c_void_p(int(d_tgt)), int(nt),
                                     C.byref(params), C.byref(res)), "kss_icp_dev")
        return res

    def icp_batch(self, src_all, src_off, tgt_all, tgt_off, params=None):
        s, t = _f32(src_all), _f32(tgt_all)
        so = np.ascontiguousarray(src_off, dtype=np.int64)
        to = np.ascontiguousarray(tgt_off, dtype=np.int64)
        npairs = len(so) - 1
        p = params if params is not None else self.icp_params()
        res = (IcpResult * npairs)()
        self._chk(self.L.kss_icp_batch(self.h, _p(s), _p(so), _p(t), _p(to), npairs, C.byref(p), C.cast(res, C.c_void_p)), "kss_icp_batch")
        return list(res)

    def icp_batch_dev(self, d_src_all, src_off, d_tgt_all, tgt_off, params):
        so = np.ascontiguousarray(src_off, dtype=np.int64)
        to = np.ascontiguousarray(tgt_off, dtype=np.int64)
        npairs = len(so) - 1
        res = (IcpResult * npairs)()
        self._chk(self.L.kss_icp_batch_dev(self.h, C.c_void_p(int(d_src_all)), _p(so), C.c_void_p(int(d_tgt_all)), _p(to),
                                           npairs, C.byref(params), C.cast(res, C.c_void_p)), "kss_icp_batch_dev")
        return res

    def transform_apply_f32(self, T, pts):
        a = _f32(pts)
        Tm = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
        out = np.empty_like(a)
        self._chk(self.L.kss_transform_apply_f32(self.h, _p(Tm), _p(a), len(a), _p(out)), "kss_transform_apply_f32")
        return out

    def downsample_fps(self, pts, m):
        a = _f64(pts)
        out = np.empty((int(m), 3), np.float64)
        idx = np.empty(int(m), np.int32)
        self._chk(self.L.kss_downsample_fps(self.h, _p(a), len(a), int(m), _p(out), _p(idx)), "kss_downsample_fps")
        return out, idx

    def downsample_aivs(self, pts, point_num):
        a = _f64(pts)
        out = np.empty_like(a)
        idx = np.empty(len(a), np.int32)
        k = C.c_int64(0)
        self._chk(self.L.kss_downsample_aivs(self.h, _p(a), len(a), int(point_num), _p(out), len(a), C.byref(k), _p(idx)), "kss_downsample_aivs")
        return out[:k.value].copy(), idx[:k.value].copy()

    def downsample_aivs_pair(self, pts0, point_num0, pts1, point_num1):
        """Both clouds of a registration at once (the second on a worker context): ((points, indices), (points, indices))."""
        a, b = _f64(pts0), _f64(pts1)
        oa, ob = np.empty_like(a), np.empty_like(b)
        ia, ib = np.empty(len(a), np.int32), np.empty(len(b), np.int32)
        ka, kb = C.c_int64(0), C.c_int64(0)
        rc = (C.c_int * 2)(0, 0)
        self._chk(self.L.kss_downsample_aivs_pair(self.h, _p(a), len(a), int(point_num0), _p(oa), len(a), C.byref(ka), _p(ia),
                                                  _p(b), len(b), int(point_num1), _p(ob), len(b), C.byref(kb), _p(ib), rc), "kss_downsample_aivs_pair")
        for k in range(2):
            self._chk(rc[k], "kss_downsample_aivs_pair (cloud %d)" % k)
        return (oa[:ka.value].copy(), ia[:ka.value].copy()), (ob[:kb.value].copy(), ib[:kb.value].copy())

    def normals_orient(self, pts, normals):
        a = _f64(pts); nrm = _f64(normals).copy()
        self._chk(self.L.kss_normals_orient(self.h, _p(a), len(a), _p(nrm)), "kss_normals_orient")
        return nrm

    def register_batch(self, src_all, src_off, tgt_all, tgt_off, sample_cap=2000, accurate=8.0, iters=1000, workers=0, want_align=False):
        s, t = _f64(src_all), _f64(tgt_all)
        so = np.ascontiguousarray(src_off, dtype=np.int64); to = np.ascontiguousarray(tgt_off, dtype=np.int64)
        npairs = len(so) - 1
        res = (RegisterResult * npairs)()
        align = np.empty_like(s) if want_align else None
        self._chk(self.L.kss_register_batch(self.h, _p(s), _p(so), _p(t), _p(to), npairs, int(sample_cap), float(accurate), int(iters),
                                            int(workers), _p(align) if want_align else None, C.cast(res, C.c_void_p)), "kss_register_batch")
        return (list(res), align) if want_align else list(res)

    def downsample_octree(self, pts):
        """(selected point indices in octree depth-first voxel order -- repeats possible --, resolution)."""
        a = _f64(pts)
        idx = np.empty(len(a), np.int32)
        k = C.c_int64(0)
        res = C.c_double(0)
        self._chk(self.L.kss_downsample_octree(self.h, _p(a), len(a), _p(idx), len(a), C.byref(k), C.byref(res)), "kss_downsample_octree")
        return idx[:k.value].copy(), res.value

    def knn(self, query, tgt, k):
        q, t = _f32(query), _f32(tgt)
        idx = np.empty((len(q), int(k)), np.int32)
        d2 = np.empty((len(q), int(k)), np.float32)
        self._chk(self.L.kss_knn(self.h, _p(q), len(q), _p(t), len(t), int(k), _p(idx), _p(d2)), "kss_knn")
        return idx, d2

    def normals(self, pts, k=20):
        a = _f64(pts)
        out = np.empty_like(a)
        self._chk(self.L.kss_normals(self.h, _p(a), len(a), int(k), _p(out)), "kss_normals")
        return out

    # ---- PCR_QM
    def pcr_qm(self, aligned, tmpl):
        a, t = _f64(aligned), _f64(tmpl)
        out = np.empty(3, np.float64)
        self._chk(self.L.kss_pcr_qm(self.h, _p(a), len(a), _p(t), len(t), _p(out)), "kss_pcr_qm")
        return out

    # ---- (a16)
    def register(self, src_sub, tgt_sub, src_full, accurate=8.0, iters=1000):
        s, t, f = _f64(src_sub), _f64(tgt_sub), _f64(src_full)
        align = np.empty_like(f)
        r = RegisterResult()
        self._chk(self.L.kss_register(self.h, _p(s), len(s), _p(t), len(t), _p(f), len(f), float(accurate), int(iters),
                                      _p(align), C.byref(r)), "kss_register")
        return {"pointAlign": align, "scale": r.scale, "angle": np.array(r.angle), "R": np.array(r.R).reshape(3, 3),
                "t": np.array(r.t), "T_icp": np.array(r.T_icp, dtype=np.float32).reshape(4, 4),
                "E_d_init": r.E_d_init, "final_fitness": r.final_fitness, "used_angle_list": bool(r.used_angle_list),
                "angle_index": r.angle_index, "n_angle_list": r.n_angle_list, "icp_iterations": r.icp_iterations,
                "icp_converged": bool(r.icp_converged), "grid": r.grid}
