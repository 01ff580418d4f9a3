#!/usr/bin/env python3
"""Rotation-search kernel (all g^3 Euler candidates x n(S') exact NN in one launch) at the reference's sizes:
time per launch from HIP events, pairs/s and the fraction of the FP32 vector peak (8 flop per pair)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle(); S = pkg.synth
ctx = pkg.Context(0)
for n, step in ((1000, 8.0), (1406, 8.0), (2000, 8.0), (2000, 12.0), (2000, 16.0)):
    src, tgt = S.make_pair(3, n, R=S.rot_axis_angle([0, 0, 1], np.deg2rad(30.0)), shape="bumpy")
    s64, t64 = src.astype(np.float64), tgt.astype(np.float64)
    ctx.rotation_search(s64, t64, step)
    ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(10):
        r = ctx.rotation_search(s64, t64, step)
    wall = (time.perf_counter() - t0) / 10
    ms, cnt = ctx.profile_get(pkg.K_ROT_SEARCH); ctx.profile_enable(False)
    gg = r.shape[0]; pairs = gg ** 3 * n * n
    k = ms / cnt * 1e-3
    line = "n=%d step=%g g=%d: kernel %.3f ms (call %.3f ms), %.2e pairs/s, %.1f TFLOP/s = %.3f of 157.3" % (
        n, step, gg, k * 1e3, wall * 1e3, pairs / k, 8 * pairs / k / 1e12, 8 * pairs / k / 1e12 / 157.3)
    if n == 1000:
        t0 = time.perf_counter(); ro = O.rotation_search(s64, t64, step); cpu = time.perf_counter() - t0
        line += "; oracle 1 core %.2f s (%.0fx), max |dE| %.1e" % (cpu, cpu / wall, np.abs(ro["value"] - r).max())
    print(line, flush=True)
