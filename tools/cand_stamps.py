#!/usr/bin/env python3
"""Timeline of the candidate-resident kernel (KSS_GRID_STAMPS=1) on the reference's Bunny pair: per candidate, the phases of
workgroup 0 summed over the passes, microseconds.  Diagnostic only."""
import ctypes as C, os, sys, time
import numpy as np
os.environ["KSS_GRID_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth
ctx = pkg.Context(0)
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_data", "registration")
if len(sys.argv) > 1:
    n = int(sys.argv[1])
    A, B = S.make_pair(4242 + n, n, R=S.rot_axis_angle([0.3, 0.2, 1.0], np.deg2rad(30.0)), t=(0.05, -0.02, 0.03), shape="bumpy")
    A = A.astype(np.float64); B = B.astype(np.float64); s, t = A, B
else:
    A = np.loadtxt(os.path.join(d, "Bunny.gird"), skiprows=1); B = np.loadtxt(os.path.join(d, "Bunny.wlop"), skiprows=1)
    m = min(len(A), len(B)) // 2
    s, _ = ctx.downsample_aivs(A, m); t, _ = ctx.downsample_aivs(B, m)
ctx.register(s, t, A, 8.0, 1000)
t0 = time.perf_counter(); r = ctx.register(s, t, A, 8.0, 1000); ms = (time.perf_counter() - t0) * 1e3
L = pkg.load_library()
buf = np.zeros(16 * 4096, np.uint64)
L.kss_debug_grid_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
k = L.kss_debug_grid_stamps(ctx.h, buf.ctypes.data_as(C.c_void_p), buf.size)
st = buf[:k].reshape(-1, 16).astype(np.int64)
print("kss_register %.3f ms (with stamps); candidates %d; stamped records %d" % (ms, r["n_angle_list"], len(st)))
t00 = st[:, 0].min()
for p in range(len(st)):
    v = st[p]; n = max(int(v[7]) - 1, 1)
    print("candidate %2d: passes %3d, lifetime %7.1f us (staged after %.1f); per pass: gate %5.2f sweep %5.2f sums+rows %5.2f tags+wait %5.2f total+publish %5.2f = %5.2f us" % (
        p, v[7], (v[15] - v[0]) / 100.0, (v[1] - v[0]) / 100.0, v[8] / n / 100.0, v[9] / n / 100.0, v[10] / n / 100.0, v[11] / n / 100.0, v[12] / n / 100.0,
        (v[8] + v[9] + v[10] + v[11] + v[12]) / n / 100.0))
print("kernel span %.1f us" % ((st[:, 15].max() - t00) / 100.0))
