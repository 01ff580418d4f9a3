#!/usr/bin/env python3
"""One context registers the same pair over and over while other threads keep the GPU busy with unrelated work on contexts of
their own: every registration must give the same bits.  usage: load_repro.py [reps] [load: batch|sweep|register] [threads]
With KSS_DEBUG_CANDS=1 the library prints every candidate's record to stderr; the markers this script writes there
("BASELINE ABOVE", "DIFFERING ABOVE") say which of those dumps belong to the solitary run and to a differing one."""
import hashlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
load = sys.argv[2] if len(sys.argv) > 2 else "register"
ctx = pkg.Context(0)
src, tgt = S.make_pair(424, 6000, R=S.rot_axis_angle([0.3, 0.34, 1.0], np.deg2rad(125.0)), scale=1.06, shape="bumpy")
src = src.astype(np.float64); tgt = tgt.astype(np.float64)
t_, _ = ctx.downsample_aivs(tgt, 2000); s_, _ = ctx.downsample_aivs(src, 2000)
def dig(r):
    h = hashlib.sha1()
    for k in ("R", "t", "T_icp"): h.update(np.ascontiguousarray(r[k]).tobytes())
    h.update(np.float64(r["final_fitness"]).tobytes()); h.update(np.int64(r["angle_index"]).tobytes()); h.update(np.int64(r["icp_iterations"]).tobytes())
    return h.hexdigest()[:12]
base = ctx.register(s_, t_, src, 8.0, 1000); hb = dig(base)
print("ctx", hex(ctx.h if isinstance(ctx.h, int) else ctx.h.value), "base", hb, "candidates", base["n_angle_list"], "index", base["angle_index"], "fitness %.17g" % base["final_fitness"], flush=True)
stop = False
def worker(k):
    c = pkg.Context(0)
    if load == "sweep":
        a, b = S.make_pair(7 + k, 40000)
        while not stop: c.icp(a, b, c.icp_params(max_iterations=3, nn_mode=pkg.NN_BRUTE))
    elif load == "batch":
        pairs = [S.config_c3_pair(50 + i, 3000) for i in range(64)]
        sa = np.concatenate([p[0] for p in pairs]); ta = np.concatenate([p[1] for p in pairs]); off = np.arange(65, dtype=np.int64) * 3000
        while not stop: c.icp_batch(sa, off, ta, off, c.icp_params(max_iterations=10, fixed_iterations=1))
    else:
        a, b = S.make_pair(500 + k, 5000, R=S.rot_axis_angle([0.1, 1.0, 0.2], np.deg2rad(40.0 + 10 * k)), shape="bumpy")
        a = a.astype(np.float64); b = b.astype(np.float64)
        ta, _ = c.downsample_aivs(b, 2000); sa, _ = c.downsample_aivs(a, 2000)
        while not stop: c.register(sa, ta, a, 8.0, 1000)
    c.close()
th = [threading.Thread(target=worker, args=(k,)) for k in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6)]
for t in th: t.start()
bad = 0; t0 = time.time()
for i in range(reps):
    r = ctx.register(s_, t_, src, 8.0, 1000)
    if i == 0: sys.stderr.write('BASELINE ABOVE\n')
    if dig(r) != hb:
        sys.stderr.write('DIFFERING ABOVE\n')
        bad += 1
        print("rep %d differs: index %d iterations %d fitness %.17g max|dR| %.3e" % (i, r["angle_index"], r["icp_iterations"], r["final_fitness"], np.abs(r["R"] - base["R"]).max()), flush=True)
stop = True
for t in th: t.join()
print("load %s: %d registrations, %d differed, %.1f s" % (load, reps, bad, time.time() - t0))
