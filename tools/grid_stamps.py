#!/usr/bin/env python3
"""Timeline of the fused grid kernel from in-kernel s_memrealtime stamps (KSS_GRID_STAMPS=1), in microseconds
relative to the first workgroup's start.  Diagnostic only: the stamps perturb the kernel slightly."""
import ctypes as C, os, sys
import numpy as np
mode = sys.argv[3] if len(sys.argv) > 3 else "1"      # 2: the gated chain
os.environ["KSS_GRID_STAMPS"] = mode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package(); S = pkg.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
src, tgt = S.make_pair(0, n, R=S.rot_axis_angle([0, 0, 1], np.deg2rad(10.0)))
ds = torch.from_numpy(src).cuda(); dt = torch.from_numpy(tgt).cuda()
ctx = pkg.Context(0)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
p = ctx.icp_params(max_iterations=iters, fixed_iterations=1, compute_fitness=0, nn_mode=pkg.NN_GRID)
ctx.icp_dev(ds.data_ptr(), n, dt.data_ptr(), n, p)
ctx.icp_dev(ds.data_ptr(), n, dt.data_ptr(), n, p)
L = pkg.load_library()
buf = np.zeros(16 * 4096, np.uint64)
L.kss_debug_grid_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
k = L.kss_debug_grid_stamps(ctx.h, buf.ctypes.data_as(C.c_void_p), buf.size)
st = buf[:k].reshape(-1, 16).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
us = lambda a: (a - t0) / 100.0
names = [(0, "start"), (13, "loads issued"), (9, "gate open"), (14, "phase A"), (5, "B enter"), (8, "B answered"), (15, "phase B"), (1, "searched"), (2, "row ready"), (3, "ticketed")]
print("workgroups:", len(st))
for i, nm in names:
    v = us(st[:, i])
    print("%-12s min %7.2f  median %7.2f  max %7.2f us" % (nm, v.min(), np.median(v), v.max()))
last = st[st[:, 4] > 0]
if len(last):
    print("last workgroup: result stored at %.2f us" % us(last[0, 4]))
print("search duration per workgroup: median %.2f max %.2f us" % (np.median(us(st[:, 1]) - us(st[:, 0])), (us(st[:, 1]) - us(st[:, 0])).max()))
nq = min(n, len(st) * 512)
print("distance evaluations per query: %.2f   evaluation slots issued per query (wave-padded): %.2f" % (st[:, 10].sum() / n, st[:, 11].sum() / n))
print("walkers per workgroup: median %d  max %d" % (np.median(st[:, 12]), st[:, 12].max()))
