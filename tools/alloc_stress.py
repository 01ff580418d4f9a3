#!/usr/bin/env python3
"""Registrations on one context while another thread creates and destroys contexts (hipMalloc / hipFree / hipHostMalloc in
flight): the resident engines must not be stalled by HIP calls of other threads.  Prints the slowest registration."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth
ctx = pkg.Context(0)
src, tgt = S.make_pair(6243, 2000, R=S.rot_axis_angle([0.3, 0.2, 1.0], np.deg2rad(30.0)), t=(0.05, -0.02, 0.03), shape="bumpy")
src = src.astype(np.float64); tgt = tgt.astype(np.float64)
t_, _ = ctx.downsample_aivs(tgt, 1000); s_, _ = ctx.downsample_aivs(src, 1000)
ctx.register(s_, t_, src, 8.0, 1000)
stop = False
made = 0
def churn():
    global made
    a, b = S.make_pair(1, 3000)
    while not stop:
        c = pkg.Context(0)
        c.icp(a, b, c.icp_params(max_iterations=3))
        c.close()
        made += 1
th = threading.Thread(target=churn); th.start()
worst = 0.0; n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.perf_counter()
for i in range(n):
    t1 = time.perf_counter(); ctx.register(s_, t_, src, 8.0, 1000); worst = max(worst, time.perf_counter() - t1)
dt = time.perf_counter() - t0
stop = True; th.join()
print("%d registrations while %d contexts came and went: mean %.3f ms, slowest %.3f ms" % (n, made, dt / n * 1e3, worst * 1e3))
