#!/usr/bin/env python3
"""Timeline of the fused pass in a C3-like batch (KSS_GRID_STAMPS=1): per-workgroup phase durations in microseconds
(in-kernel s_memrealtime stamps of the last launch = the fitness pass unless --iters says otherwise).  Diagnostic only."""
import ctypes as C, os, sys
import numpy as np
os.environ["KSS_GRID_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package(); S = pkg.synth
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 10000
src = np.empty((npairs * n, 3), np.float32); tgt = np.empty((npairs * n, 3), np.float32)
for i in range(npairs):
    s, t = S.config_c3_pair(i, n)
    src[i * n:(i + 1) * n] = s; tgt[i * n:(i + 1) * n] = t
off = np.arange(npairs + 1, dtype=np.int64) * n
ds = torch.from_numpy(src).cuda(); dt = torch.from_numpy(tgt).cuda()
ctx = pkg.Context(0)
p = ctx.icp_params(max_iterations=iters, fixed_iterations=1, compute_fitness=0, nn_mode=pkg.NN_GRID)
ctx.icp_batch_dev(ds.data_ptr(), off, dt.data_ptr(), off, p)
L = pkg.load_library()
buf = np.zeros(16 * 32768, np.uint64)
L.kss_debug_grid_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
k = L.kss_debug_grid_stamps(ctx.h, buf.ctypes.data_as(C.c_void_p), buf.size)
print('entries', k, 'nonzero', int((buf[:max(k,0)] != 0).sum()))
st = buf[:k].reshape(-1, 16).astype(np.int64)
st = st[st[:, 0] > 0]
print("workgroups:", len(st), " kernel span %.1f us" % ((st[:, [0, 1, 2]].max() - st[:, 0].min()) / 100.0))
seq = [(0, "start"), (13, "loads in"), (14, "phase A"), (15, "phase B"), (1, "searched"), (2, "row ready"), (3, "ticketed")]
prev = None
for i, nm in seq:
    ok = st[:, i] > 0
    if prev is not None:
        both = ok & (st[:, prev] > 0)
        d = (st[both, i] - st[both, prev]) / 100.0
        print("%-10s +%6.2f us median  (+%6.2f p90)  since %s" % (nm, np.median(d), np.percentile(d, 90), seq[[s for s, _ in seq].index(prev)][1]))
    prev = i
tot = (st[:, 2] - st[:, 0]) / 100.0
print("workgroup start -> row ready: median %.2f  p90 %.2f us" % (np.median(tot), np.percentile(tot, 90)))
print("walkers per workgroup: median %d" % np.median(st[:, 12]))
