"""ctypes loader for the CPU oracle (oracle/kss_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product path (kss-icp_amd/).
See oracle/kss_oracle.h for the parity-pin status of each function.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libkss_oracle.so")
    src = os.path.join(_HERE, "kss_oracle.c")
    hdr = os.path.join(_HERE, "kss_oracle.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


class Preshape(C.Structure):
    _fields_ = [("c_src", C.c_double * 3), ("c_tgt", C.c_double * 3), ("shift", C.c_double * 3),
                ("r_src", C.c_double), ("r_tgt", C.c_double), ("scale", C.c_double)]


class IcpParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("max_corr_dist", C.c_double),
                ("transformation_epsilon", C.c_double), ("euclidean_fitness_epsilon", C.c_double),
                ("abs_mse_epsilon", C.c_double), ("min_correspondences", C.c_int),
                ("fixed_iterations", C.c_int), ("fma", C.c_int), ("use_kdtree", C.c_int),
                ("nthreads", C.c_int), ("compute_fitness", C.c_int)]


class IcpResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("iterations", C.c_int), ("converged", C.c_int),
                ("state", C.c_int), ("fitness", C.c_double), ("last_mse", C.c_double),
                ("nn_seconds", C.c_double), ("build_seconds", C.c_double), ("total_seconds", C.c_double)]


class IcpTrace(C.Structure):
    _fields_ = [("cap", C.c_int), ("n", C.c_int), ("sums", C.POINTER(C.c_double)), ("Tk", C.POINTER(C.c_float))]


class KssResult(C.Structure):
    _fields_ = [("scale", C.c_double), ("R0_angle", C.c_double * 3), ("used_angle_list", C.c_int),
                ("angle_index", C.c_int), ("n_angle_list", C.c_int), ("E_d_init", C.c_double),
                ("final_fitness", C.c_double), ("T_icp", C.c_float * 16), ("R", C.c_double * 9),
                ("t", C.c_double * 3), ("icp_iterations", C.c_int), ("icp_converged", C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        # KSS_ORACLE_SO: another build of the same source (bench.py's cpu_baseline compiles a -march=native one on the box)
        alt = os.environ.get("KSS_ORACLE_SO")
        _LIB = C.CDLL(alt if alt and os.path.exists(alt) else build())
        L = _LIB
        L.ko_kdtree_build.restype = C.c_void_p
        L.ko_kdtree_build.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        L.ko_kdtree_free.argtypes = [C.c_void_p]
        L.ko_kdtree_nn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ko_nn_brute.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.ko_preshape_stats.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(Preshape)]
        L.ko_similarity_apply.argtypes = [C.c_void_p, C.c_int64, C.POINTER(Preshape), C.c_void_p]
        L.ko_axis_rotate.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_int64]
        L.ko_pose_apply.argtypes = [C.c_void_p, C.c_int64, C.POINTER(Preshape), C.c_void_p, C.c_void_p]
        L.ko_grid_angles.argtypes = [C.c_double, C.c_void_p, C.c_int]
        L.ko_error_ave.restype = C.c_double
        L.ko_error_ave.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        L.ko_local_min.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ko_rotation_search.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double,
                                         C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.ko_icp_default_params.argtypes = [C.POINTER(IcpParams)]
        L.ko_icp.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(IcpParams),
                             C.POINTER(IcpResult), C.POINTER(IcpTrace)]
        L.ko_rigid_from_sums.argtypes = [C.c_void_p, C.c_void_p]
        L.ko_svd3.argtypes = [C.c_void_p] * 4
        L.ko_mat4_mul.argtypes = [C.c_void_p] * 3
        L.ko_transform_points_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.ko_pcr_qm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        L.ko_kssicp_register.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                         C.c_double, C.c_int, C.c_int, C.c_void_p, C.POINTER(KssResult)]
        L.ko_ply_load.restype = C.c_int64
        L.ko_ply_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_double))]
        L.ko_free.argtypes = [C.c_void_p]
        L.ko_fps.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.ko_knn_brute.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.ko_normals_pcl.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.ko_normals_regular.restype = None
        L.ko_normals_regular.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.ko_octree_downsample.restype = C.c_int64
        L.ko_octree_downsample.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_double)]
        L.ko_aivs.restype = C.c_int64
        L.ko_aivs.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64]
        L.ko_splitmix64.restype = C.c_uint64
        L.ko_splitmix64.argtypes = [C.c_uint64, C.c_uint64]
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 3)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------------------
def preshape_stats(S, T):
    S, T = _f64(S), _f64(T)
    ps = Preshape()
    lib().ko_preshape_stats(_p(S), len(S), _p(T), len(T), C.byref(ps))
    return ps


def similarity_apply(P, ps):
    P = _f64(P)
    out = np.empty_like(P)
    lib().ko_similarity_apply(_p(P), len(P), C.byref(ps), _p(out))
    return out


def axis_rotate(cord, angle, P):
    out = _f64(P).copy()
    lib().ko_axis_rotate(int(cord), float(angle), _p(out), len(out))
    return out


def pose_apply(P, ps, angle):
    P = _f64(P)
    ang = np.ascontiguousarray(angle, dtype=np.float64)
    out = np.empty_like(P)
    lib().ko_pose_apply(_p(P), len(P), C.byref(ps), _p(ang), _p(out))
    return out


def nn_brute(q, t, fma=0):
    q, t = _f32(q), _f32(t)
    idx = np.empty(len(q), np.int32)
    d2 = np.empty(len(q), np.float32)
    lib().ko_nn_brute(_p(q), len(q), _p(t), len(t), int(fma), _p(idx), _p(d2))
    return idx, d2


class KdTree:
    def __init__(self, t, leaf=15):
        self.t = _f32(t)
        self.h = lib().ko_kdtree_build(_p(self.t), len(self.t), leaf)

    def nn(self, q, fma=0, nthreads=1):
        q = _f32(q)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float32)
        lib().ko_kdtree_nn(self.h, _p(q), len(q), int(fma), int(nthreads), _p(idx), _p(d2))
        return idx, d2

    def __del__(self):
        if getattr(self, "h", None):
            lib().ko_kdtree_free(self.h)
            self.h = None


def grid_angles(step):
    buf = np.empty(256, np.float64)
    g = lib().ko_grid_angles(float(step), _p(buf), 256)
    return buf[:g].copy()


def error_ave(cloud, tgt):
    cloud = _f64(cloud)
    tf = _f32(np.asarray(tgt, dtype=np.float64).astype(np.float32))
    return lib().ko_error_ave(_p(cloud), len(cloud), _p(tf), len(tf), None)


def local_min(value, i, j, k, r=2):
    v = np.ascontiguousarray(value, dtype=np.float64)
    g = v.shape[0]
    return bool(lib().ko_local_min(_p(v), g, i, j, k, r))


def rotation_search(Sp, T, step):
    """Sp: pre-shaped source (doubles), T: target (doubles).  Returns dict."""
    Sp, T = _f64(Sp), _f64(T)
    capg = 64
    value = np.empty(capg ** 3, np.float64)
    best = np.empty(3, np.float64)
    alist = np.empty(3 * capg ** 3, np.float64)
    nl = C.c_int(0)
    g = lib().ko_rotation_search(_p(Sp), len(Sp), _p(T), len(T), float(step), _p(value), capg,
                                 _p(best), _p(alist), C.byref(nl), capg ** 3)
    if g < 0:
        raise RuntimeError("ko_rotation_search failed: %d" % g)
    return {"g": g, "value": value[:g ** 3].reshape(g, g, g).copy(), "angle": best,
            "angle_list": alist[:3 * nl.value].reshape(-1, 3).copy()}


def icp_params(**kw):
    p = IcpParams()
    lib().ko_icp_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def icp(src, tgt, params=None, trace_cap=0):
    src, tgt = _f32(src), _f32(tgt)
    p = params if params is not None else icp_params()
    res = IcpResult()
    tr = None
    if trace_cap > 0:
        sums = np.zeros((trace_cap, 20), np.float64)
        tk = np.zeros((trace_cap, 16), np.float32)
        tr = IcpTrace(trace_cap, 0, sums.ctypes.data_as(C.POINTER(C.c_double)), tk.ctypes.data_as(C.POINTER(C.c_float)))
    rc = lib().ko_icp(_p(src), len(src), _p(tgt), len(tgt), C.byref(p), C.byref(res), C.byref(tr) if tr else None)
    if rc != 0:
        raise RuntimeError("ko_icp rc=%d" % rc)
    out = {"T": np.array(res.T, dtype=np.float32).reshape(4, 4), "iterations": res.iterations,
           "converged": bool(res.converged), "state": res.state, "fitness": res.fitness,
           "last_mse": res.last_mse, "nn_seconds": res.nn_seconds, "build_seconds": res.build_seconds,
           "total_seconds": res.total_seconds}
    if tr:
        out["trace_sums"] = sums[:tr.n].copy()
        out["trace_Tk"] = tk[:tr.n].reshape(-1, 4, 4).copy()
    return out


def rigid_from_sums(sums):
    s = np.ascontiguousarray(sums, dtype=np.float64)
    T = np.empty(16, np.float32)
    lib().ko_rigid_from_sums(_p(s), _p(T))
    return T.reshape(4, 4)


def svd3(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    U = np.empty((3, 3)); s = np.empty(3); V = np.empty((3, 3))
    lib().ko_svd3(_p(A), _p(U), _p(s), _p(V))
    return U, s, V


def mat4_mul(A, B):
    A = np.ascontiguousarray(A, dtype=np.float32); B = np.ascontiguousarray(B, dtype=np.float32)
    out = np.empty((4, 4), np.float32)
    lib().ko_mat4_mul(_p(A), _p(B), _p(out))
    return out


def transform_points_f32(T, P):
    T = np.ascontiguousarray(T, dtype=np.float32); P = _f32(P)
    out = np.empty_like(P)
    lib().ko_transform_points_f32(_p(T), _p(P), len(P), _p(out))
    return out


def pcr_qm(aligned, tmpl):
    a, t = _f64(aligned), _f64(tmpl)
    out = np.empty(3, np.float64)
    lib().ko_pcr_qm(_p(a), len(a), _p(t), len(t), _p(out))
    return out


def kssicp_register(Ssub, Tsub, Sfull, accurate=8.0, iters=1000, use_kdtree=1):
    Ssub, Tsub, Sfull = _f64(Ssub), _f64(Tsub), _f64(Sfull)
    align = np.empty_like(Sfull)
    r = KssResult()
    rc = lib().ko_kssicp_register(_p(Ssub), len(Ssub), _p(Tsub), len(Tsub), _p(Sfull), len(Sfull),
                                  float(accurate), int(iters), int(use_kdtree), _p(align), C.byref(r))
    if rc != 0:
        raise RuntimeError("ko_kssicp_register rc=%d" % rc)
    return {"pointAlign": align, "scale": r.scale, "angle": np.array(r.R0_angle),
            "used_angle_list": bool(r.used_angle_list), "angle_index": r.angle_index,
            "n_angle_list": r.n_angle_list, "E_d_init": r.E_d_init, "final_fitness": r.final_fitness,
            "T_icp": np.array(r.T_icp, dtype=np.float32).reshape(4, 4),
            "R": np.array(r.R).reshape(3, 3), "t": np.array(r.t),
            "icp_iterations": r.icp_iterations, "icp_converged": bool(r.icp_converged)}


def ply_load(path):
    pp = C.POINTER(C.c_double)()
    n = lib().ko_ply_load(path.encode(), C.byref(pp))
    if n < 0:
        return int(n), None
    arr = np.ctypeslib.as_array(pp, shape=(n * 3,)).reshape(n, 3).copy() if n > 0 else np.zeros((0, 3))
    lib().ko_free(pp)
    return int(n), arr


def fps(xyz, m):
    a = _f64(xyz)
    idx = np.empty(int(m), np.int32)
    lib().ko_fps(_p(a), len(a), int(m), _p(idx))
    return idx


def knn_brute(q, t, k):
    q, t = _f32(q), _f32(t)
    idx = np.empty((len(q), k), np.int32)
    d2 = np.empty((len(q), k), np.float32)
    lib().ko_knn_brute(_p(q), len(q), _p(t), len(t), int(k), _p(idx), _p(d2))
    return idx, d2


def normals_pcl(pts, k=20):
    a = _f64(pts)
    out = np.empty_like(a)
    lib().ko_normals_pcl(_p(a), len(a), int(k), _p(out))
    return out


def aivs(xyz, point_num):
    """Indices of the AIVS-selected points, in output order."""
    a = _f64(xyz)
    idx = np.empty(len(a), np.int32)
    k = lib().ko_aivs(_p(a), len(a), int(point_num), _p(idx), len(idx))
    if k < 0:
        raise RuntimeError("ko_aivs rc=%d" % k)
    return idx[:k].copy()


def normals_regular(pts, normals):
    """estimateNormal_RegularNormal: re-oriented copy of `normals`."""
    a = _f64(pts); nrm = _f64(normals).copy()
    lib().ko_normals_regular(_p(a), len(a), _p(nrm))
    return nrm


def octree_downsample(xyz):
    """(indices of the selected points in octree depth-first voxel order -- repeats possible --, resolution)."""
    a = _f64(xyz)
    idx = np.empty(len(a), np.int32)
    res = C.c_double(0)
    k = lib().ko_octree_downsample(_p(a), len(a), _p(idx), len(idx), C.byref(res))
    if k < 0:
        raise RuntimeError("ko_octree_downsample rc=%d" % k)
    return idx[:k].copy(), res.value


def splitmix64(seed, counter):
    return int(lib().ko_splitmix64(C.c_uint64(seed), C.c_uint64(counter)))
