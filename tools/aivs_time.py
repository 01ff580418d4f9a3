import sys, time, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle(); S = pkg.synth
ctx = pkg.Context(0)
for n, m in ((100000, 2000), (1000000, 2000), (20000, 2000)):
    P = S.bumpy(3, n)
    ctx.downsample_aivs(P, m)
    t0 = time.perf_counter(); out, idx = ctx.downsample_aivs(P, m); t1 = time.perf_counter()
    r = O.aivs(P, m); t2 = time.perf_counter()
    print("n=%d m=%d: gpu %.2f ms (incl. %d MB upload), cpu oracle %.1f ms, selected %d, identical %s" % (n, m, (t1 - t0) * 1e3, n * 24 // 1000000, (t2 - t1) * 1e3, len(idx), np.array_equal(idx, r)))
