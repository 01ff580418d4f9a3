#!/usr/bin/env python3
"""Batched ICP on badly posed pairs (sources start far from their targets): the batched cell-list engine must stay
within reach of the brute-force engine (bounded fallback) and give the same transforms."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package(); S = pkg.synth
ctx = pkg.Context(0)
for shift in (0.0, 0.3, 1.0, 3.0):
    npairs, n = 64, 5000
    src = np.empty((npairs * n, 3), np.float32); tgt = np.empty((npairs * n, 3), np.float32)
    for i in range(npairs):
        s, t = S.make_pair(100 + i, n, R=S.rot_axis_angle([0.2, 1.0, 0.3], np.deg2rad(25.0)), t=(shift, -0.5 * shift, 0.25 * shift), shape="bumpy")
        src[i * n:(i + 1) * n] = s; tgt[i * n:(i + 1) * n] = t
    off = np.arange(npairs + 1, dtype=np.int64) * n
    ds = torch.from_numpy(src).cuda(); dt = torch.from_numpy(tgt).cuda()
    out = {}
    for name, m in (("grid", pkg.NN_GRID), ("brute", pkg.NN_BRUTE)):
        p = ctx.icp_params(max_iterations=10, fixed_iterations=1, max_corr_dist=100.0, nn_mode=m)
        ctx.icp_batch_dev(ds.data_ptr(), off, dt.data_ptr(), off, p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ctx.icp_batch_dev(ds.data_ptr(), off, dt.data_ptr(), off, p)
        torch.cuda.synchronize(); out[name] = (time.perf_counter() - t0, r)
    d = max(np.abs(out["grid"][1][i].matrix() - out["brute"][1][i].matrix()).max() for i in range(npairs))
    print("shift %.1f: cell lists %.2f ms, brute force %.2f ms, max |dT| %.1e" % (shift, out["grid"][0] * 1e3, out["brute"][0] * 1e3, d), flush=True)
