#!/usr/bin/env python3
"""kss_register_batch with 1 vs many workers: which records differ, and how (a diagnostic of the concurrency of resident launches)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth; ctx = pkg.Context(0)
npairs, n = 48, 6000
cl = []
for i in range(npairs):
    s, t = S.make_pair(400 + i, n, R=S.rot_axis_angle([0.3, 0.1 + 0.01 * i, 1.0], np.deg2rad(5.0 + 5.0 * (i % 30))), scale=1.0 + 0.02 * (i % 7), shape="bumpy")
    cl.append((s.astype(np.float64), t.astype(np.float64)))
src_all = np.concatenate([c[0] for c in cl]); tgt_all = np.concatenate([c[1] for c in cl])
off = np.arange(npairs + 1, dtype=np.int64) * n
def rec(x): return dict(R=np.array(x.R), t=np.array(x.t), it=x.icp_iterations, idx=x.angle_index, fit=x.final_fitness, nl=x.n_angle_list, used=x.used_angle_list, E=x.E_d_init, scale=x.scale, ang=np.array(x.angle))
base = [rec(x) for x in ctx.register_batch(src_all, off, tgt_all, off, workers=1)]
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    for w in (8, 16):
        r = [rec(x) for x in ctx.register_batch(src_all, off, tgt_all, off, workers=w)]
        for i in range(npairs):
            d = [k for k in base[i] if not np.array_equal(base[i][k], r[i][k])]
            if d:
                bad += 1
                print("rep %d workers %d pair %d differs in %s: base it %d idx %d fit %.17g E %.17g nl %d | got it %d idx %d fit %.17g E %.17g nl %d; max|dR| %.3e" % (
                    rep, w, i, d, base[i]["it"], base[i]["idx"], base[i]["fit"], base[i]["E"], base[i]["nl"], r[i]["it"], r[i]["idx"], r[i]["fit"], r[i]["E"], r[i]["nl"], np.abs(base[i]["R"] - r[i]["R"]).max()), flush=True)
print("differences:", bad)
