#!/usr/bin/env python3
"""Timeline of the pair-resident kernel (KSS_GRID_STAMPS=1): per-pair phase times summed over the passes, microseconds.
Diagnostic only.   python tools/resident_stamps.py [npairs] [points] [iters]"""
import ctypes as C, os, sys
import numpy as np
os.environ["KSS_GRID_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package(); S = pkg.synth
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
src = np.empty((npairs * n, 3), np.float32); tgt = np.empty((npairs * n, 3), np.float32)
for i in range(npairs):
    s, t = S.config_c3_pair(i, n)
    src[i * n:(i + 1) * n] = s; tgt[i * n:(i + 1) * n] = t
off = np.arange(npairs + 1, dtype=np.int64) * n
ds = torch.from_numpy(src).cuda(); dt = torch.from_numpy(tgt).cuda()
ctx = pkg.Context(0)
p = ctx.icp_params(max_iterations=iters, fixed_iterations=1)
ctx.icp_batch_dev(ds.data_ptr(), off, dt.data_ptr(), off, p)
ctx.icp_batch_dev(ds.data_ptr(), off, dt.data_ptr(), off, p)
L = pkg.load_library()
buf = np.zeros(16 * 65536, np.uint64)
L.kss_debug_grid_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
k = L.kss_debug_grid_stamps(ctx.h, buf.ctypes.data_as(C.c_void_p), buf.size)
st = buf[:k].reshape(-1, 16).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
print("pairs:", len(st), " kernel span %.1f us" % ((st[:, 15].max() - t0) / 100.0))
passes = np.maximum(st[:, 7], 1)
print("passes per pair (median): %d" % np.median(passes))
print("start (us after first): median %.1f  max %.1f" % (np.median(st[:, 0] - t0) / 100.0, (st[:, 0].max() - t0) / 100.0))
print("move in: median %.2f us" % (np.median(st[:, 1] - st[:, 0]) / 100.0))
print("pass 0: phase B (every source searches) median %.2f us; the rest of pass 0 %.2f us" % (np.median(st[:, 13]) / 100.0, np.median(st[:, 6]) / 100.0))
st[:, 10] = st[:, 10] + st[:, 2] + st[:, 3]          # phase B = wait for the slowest wave + serving the queue + the rest
for idx, nm in ((8, "gate wait"), (9, "phase A"), (10, "phase B"), (11, "phase C"), (12, "total+publish")):
    per = st[:, idx] / np.maximum(passes - 1, 1) / 100.0
    print("%-14s per pass: median %6.2f us  p90 %6.2f" % (nm, np.median(per), np.percentile(per, 90)))
per = lambda k: np.median(st[:, k] / np.maximum(passes - 1, 1)) / 100.0
print("inside phase B (wave 0's view), per pass: wait for the slowest wave's phase A %.2f us, serving the queue %.2f us, the rest (barriers, pick-up, further rounds) %.2f us" % (
    per(2), per(3), per(10) - per(2) - per(3)))
print("requests queued per pass (median over pairs): %.0f" % np.median(st[:, 14] / passes))
rq = st[:, 14] / passes
print("requests queued per pass, over pairs: p10 %.0f median %.0f p90 %.0f max %.0f" % (np.percentile(rq, 10), np.median(rq), np.percentile(rq, 90), rq.max()))
life = (st[:, 15] - st[:, 0]) / 100.0
print("pair lifetime: median %.1f us  p90 %.1f  max %.1f" % (np.median(life), np.percentile(life, 90), life.max()))
print("sum of lifetimes / 256 CUs: %.1f us" % (life.sum() / 256.0))
order = np.argsort(st[:, 0])
for lo, hi in ((0, 256), (256, 512), (512, 768), (768, 1024)):
    sel = order[lo:hi]
    if len(sel) == 0: continue
    print("pairs started %4d..%4d: start %.0f..%.0f us, lifetime median %.0f max %.0f, gate wait per pass median %.2f max %.2f, phase B per pass median %.1f" % (
        lo, hi, (st[sel, 0].min() - t0) / 100.0, (st[sel, 0].max() - t0) / 100.0, np.median(life[sel]), life[sel].max(),
        np.median(st[sel, 8] / np.maximum(passes[sel] - 1, 1)) / 100.0, (st[sel, 8] / np.maximum(passes[sel] - 1, 1)).max() / 100.0,
        np.median(st[sel, 10] / np.maximum(passes[sel] - 1, 1)) / 100.0))
# which pairs are the long ones?  (C3's pairs differ in their rotation angle, U(0, 15 deg), and in their random points)
ang = np.array([15.0 * S.u01(4000 + i, 4)[0] for i in range(len(st))])
rq_all = st[:, 14] / passes
for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 12), (12, 15)):
    sel = (ang >= lo) & (ang < hi)
    if sel.any():
        print("angle %2d..%2d deg: %4d pairs, lifetime median %6.0f us, requests per pass median %5.0f" % (lo, hi, sel.sum(), np.median(life[sel]), np.median(rq_all[sel])))
worst = np.argsort(-life)[:5]
for w in worst:
    print("slow pair %d (angle %.2f deg): lifetime %.0f us: gate %.0f A %.0f B %.0f C %.0f pub %.0f pass0-B %.0f pass0-rest %.0f" % (w, ang[w], life[w], st[w, 8] / 100.0, st[w, 9] / 100.0, st[w, 10] / 100.0, st[w, 11] / 100.0, st[w, 12] / 100.0, st[w, 13] / 100.0, st[w, 6] / 100.0))
