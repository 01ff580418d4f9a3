#!/usr/bin/env python3
"""End-to-end KSSICP_Registration path (AIVS down-sample, pre-shape, 729-candidate rotation search, candidate ICP
batch, final transform of the full cloud) on the reference's Bunny pair: GPU vs the oracle on one CPU core."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_data", "registration")
S = np.loadtxt(os.path.join(d, "Bunny.gird"), skiprows=1); T = np.loadtxt(os.path.join(d, "Bunny.wlop"), skiprows=1)
m = min(len(S), len(T)) // 2
ctx = pkg.Context(0)
def gpu():
    s, _ = ctx.downsample_aivs(S, m); t, _ = ctx.downsample_aivs(T, m)
    return ctx.register(s, t, S, 8.0, 1000)
gpu()
t0 = time.perf_counter(); r = gpu(); t1 = time.perf_counter()
ctx.profile_enable(True); ctx.profile_reset(); gpu()
rs = ctx.profile_get(pkg.K_ROT_SEARCH); ctx.profile_enable(False)
k = O.kssicp_register(S[O.aivs(S, m)], T[O.aivs(T, m)], S, 8.0, 1000); t2 = time.perf_counter()
print("Bunny %d x %d -> %d samples: GPU %.2f ms (rotation search kernel %.3f ms), oracle 1 core %.0f ms, max|dR| %.1e, candidates %d" %
      (len(S), len(T), m, (t1 - t0) * 1e3, rs[0] / max(1, rs[1]), (t2 - t1) * 1e3, np.abs(r["R"] - k["R"]).max(), r["n_angle_list"]))
# stage breakdown (wall clock through the C-ABI, second call of each)
def tm(f, reps=5):
    f(); t = time.perf_counter()
    for _ in range(reps): out = f()
    return (time.perf_counter() - t) / reps * 1e3, out
ta, (s, _) = tm(lambda: ctx.downsample_aivs(S, m))
tb, (t, _) = tm(lambda: ctx.downsample_aivs(T, m))
tp, ((s2, _), (t2, _)) = tm(lambda: ctx.downsample_aivs_pair(S, m, T, m))
assert np.array_equal(s, s2) and np.array_equal(t, t2)
print("AIVS of both clouds in one call (second cloud on a worker context): %.2f ms" % tp)
tr, r = tm(lambda: ctx.register(s, t, S, 8.0, 1000))
tn, _ = tm(lambda: ctx.register(s, t, None, 8.0, 1000)) if False else (0.0, None)
print("AIVS source %.2f ms, AIVS target %.2f ms, kss_register %.2f ms (judge ICP %d iterations, E_d_init %.2e, used list %d)" %
      (ta, tb, tr, r["icp_iterations"], r["E_d_init"], r["used_angle_list"]))
os.environ["KSS_TIMING"] = "1"
ctx.register(s, t, S, 8.0, 1000)
