#!/usr/bin/env python3
"""Throughput of full KSS registrations (kss_register_batch) against the worker count: N pairs of ~n-point clouds."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth
ctx = pkg.Context(0)
npairs, n = 64, 10000
cl = []
for i in range(npairs):
    s, t = S.make_pair(400 + i, n, R=S.rot_axis_angle([0.3, 0.1 + 0.01 * i, 1.0], np.deg2rad(5.0 + 5.0 * (i % 30))), scale=1.0 + 0.02 * (i % 7), shape="bumpy")
    cl.append((s.astype(np.float64), t.astype(np.float64)))
src_all = np.concatenate([c[0] for c in cl]); tgt_all = np.concatenate([c[1] for c in cl])
off = np.arange(npairs + 1, dtype=np.int64) * n
ctx.register_batch(src_all[:2 * n], off[:3], tgt_all[:2 * n], off[:3])
base = None
for w in [int(x) for x in sys.argv[1:]] or (1, 2, 4, 8, 16):
    t0 = time.perf_counter(); r = ctx.register_batch(src_all, off, tgt_all, off, workers=w); dt = time.perf_counter() - t0
    base = base or dt
    print("%2d workers: %d registrations of %dx%d in %.1f ms = %.0f registrations/s (x%.2f), mean fitness %.2e" %
          (w, npairs, n, n, dt * 1e3, npairs / dt, base / dt, np.mean([x.final_fitness for x in r])), flush=True)
