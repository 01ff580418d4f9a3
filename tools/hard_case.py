#!/usr/bin/env python3
"""Cell-list vs brute-force engine on a badly initialised pair (many far queries): total ICP time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package(); S = pkg.synth
n = 100000
for deg, t in ((5.0, (0.0, 0.0, 0.0)), (20.0, (0.1, -0.05, 0.08)), (35.0, (0.4, 0.3, -0.2))):
    src, tgt = S.make_pair(3, n, R=S.rot_axis_angle([0.3, 1.0, 0.2], np.deg2rad(deg)), t=t, shape="bumpy")
    ds = torch.from_numpy(src).cuda(); dt = torch.from_numpy(tgt).cuda()
    ctx = pkg.Context(0)
    out = []
    for mode in (pkg.NN_GRID, pkg.NN_BRUTE):
        p = ctx.icp_params(nn_mode=mode)
        ctx.icp_dev(ds.data_ptr(), n, dt.data_ptr(), n, p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ctx.icp_dev(ds.data_ptr(), n, dt.data_ptr(), n, p)
        torch.cuda.synchronize(); dtm = time.perf_counter() - t0
        out.append((dtm * 1e3, r.iterations, r.fitness, np.array(r.T)))
    print("rot %4.1f deg t=%s: grid %.2f ms (%d it), brute %.2f ms (%d it), max|dT| %.1e, fitness %.2e" %
          (deg, t, out[0][0], out[0][1], out[1][0], out[1][1], np.abs(out[0][3] - out[1][3]).max(), out[0][2]))
    ctx.close()
