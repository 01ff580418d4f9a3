#!/usr/bin/env python3
"""kss_register on the reference's Bunny pair and on synthetic pairs: result bits (sha1 over R, t, scale, indices) and ms per
registration.  Run once per build / environment (KSS_CAND_RESIDENT=0|1, KSS_CAND_FUSED=0|1): the forms must print the same hashes."""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth
ctx = pkg.Context(0)
def digest(r, keys=None):
    h = hashlib.sha1()
    for k in sorted(r):
        if keys is not None and k not in keys: continue
        v = r[k]
        if isinstance(v, np.ndarray): h.update(np.ascontiguousarray(v).tobytes())
        elif isinstance(v, (int, float, np.integer, np.floating)): h.update(np.float64(v).tobytes())
    return h.hexdigest()[:12]
def run(name, s, t, full, reps=10):
    r = ctx.register(s, t, full, 8.0, 1000)
    t0 = time.perf_counter()
    for _ in range(reps): r = ctx.register(s, t, full, 8.0, 1000)
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("%-30s all %s  R,t,T_icp %s  %.3f ms  (candidates %d, used list %d index %d, final ICP %d iterations, E_d_init %.9e)" % (
        name, digest(r), digest(r, ("R", "t", "T_icp", "angle_index", "icp_iterations")), ms, r["n_angle_list"], r["used_angle_list"], r["angle_index"], r["icp_iterations"], r["E_d_init"]), flush=True)
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_data", "registration")
if os.path.isdir(d):
    A = np.loadtxt(os.path.join(d, "Bunny.gird"), skiprows=1); B = np.loadtxt(os.path.join(d, "Bunny.wlop"), skiprows=1)
    m = min(len(A), len(B)) // 2
    s, _ = ctx.downsample_aivs(A, m); t, _ = ctx.downsample_aivs(B, m)
    run("Bunny %d samples" % m, s, t, A)
for (n, seed, sub) in ((2000, 6243, 1000), (2000, 6242, 0), (900, 5, 0), (70, 6, 0)):
    src, tgt = S.make_pair(seed, n, R=S.rot_axis_angle([0.3, 0.2, 1.0], np.deg2rad(30.0)), t=(0.05, -0.02, 0.03), shape="bumpy")
    src = src.astype(np.float64); tgt = tgt.astype(np.float64)
    if sub:
        t_, _ = ctx.downsample_aivs(tgt, sub); s_, _ = ctx.downsample_aivs(src, sub)
    else:
        s_, t_ = src, tgt
    run("synthetic %d -> %d x %d" % (n, len(s_), len(t_)), s_, t_, src)
