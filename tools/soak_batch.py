#!/usr/bin/env python3
"""Soak test of "a pair gives the same bits alone or inside a batch": random ragged batches (shapes, sizes, poses --
including badly posed pairs that end in the cell search's fallback), every record of kss_icp_batch on the cell-list
engine must equal kss_icp of that pair alone, bit for bit (T, iterations, state, fitness, last MSE).
usage: python tools/soak_batch.py [batches] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
ctx = pkg.Context(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
S = pkg.synth
os.environ["KSS_GRID_NOSWITCH"] = "1"      # stay on the cell lists even when many lanes fall back


def cloud(kind, n, seed):
    if kind == "bumpy":
        return S.bumpy(seed, n)
    if kind == "sphere":
        return S.sphere(seed, n)
    v = rng.normal(size=(n, 3)) * (np.array([2.0, 0.5, 0.1]) if kind == "aniso" else 1.0)
    return v


bad = same = fell = 0
t0 = time.time()
for b in range(nb):
    npairs = int(rng.choice([2, 5, 17, 33, 70])) if os.environ.get("KSS_SOAK_BIG") is None else int(rng.choice([520, 600, 777]))   # (KSS_SOAK_BIG=1: more pairs than twice the compute units: split batches, kss_engine.hip)
    pairs = []
    for i in range(npairs):
        nt = int(rng.integers(1100, 6000)); ns = int(rng.integers(600, nt))
        kind = rng.choice(["bumpy", "sphere", "aniso", "gauss"])
        t = cloud(kind, nt, 10000 * b + i)
        ang = np.deg2rad(float(rng.choice([2.0, 8.0, 20.0, 45.0])))
        R = S.rot_axis_angle(rng.normal(size=3), ang)
        shift = rng.normal(size=3) * float(rng.choice([0.0, 0.02, 0.3]))
        s = t[rng.permutation(nt)[:ns]] @ R.T + shift + rng.normal(size=(ns, 3)) * 1e-3
        pairs.append((s.astype(np.float32), t.astype(np.float32)))
    src_all = np.concatenate([p[0] for p in pairs]); tgt_all = np.concatenate([p[1] for p in pairs])
    so = np.concatenate([[0], np.cumsum([len(p[0]) for p in pairs])]); to = np.concatenate([[0], np.cumsum([len(p[1]) for p in pairs])])
    fixed = bool(rng.integers(0, 2))
    kw = dict(max_iterations=int(rng.integers(3, 12)), fixed_iterations=1) if fixed else dict(max_iterations=60)
    kw["max_corr_dist"] = float(rng.choice([1.0, 100.0]))
    res = ctx.icp_batch(src_all, so, tgt_all, to, ctx.icp_params(nn_mode=pkg.NN_GRID, **kw))
    for i, (s, t) in enumerate(pairs):
        if npairs > 100 and i % 7: continue      # (a sample of the big batches)
        one = ctx.icp(s, t, ctx.icp_params(nn_mode=pkg.NN_GRID, **kw))
        ok = (np.array_equal(res[i].matrix(), one["T"]) and res[i].iterations == one["iterations"] and res[i].state == one["state"]
              and res[i].fitness == one["fitness"] and res[i].last_mse == one["last_mse"])
        same += ok
        if not ok:
            bad += 1
            print("MISMATCH batch %d pair %d (%d x %d) kw %s: iters %d vs %d, max|dT| %.3e, fitness %.17g vs %.17g" %
                  (b, i, len(s), len(t), kw, res[i].iterations, one["iterations"], np.abs(res[i].matrix() - one["T"]).max(), res[i].fitness, one["fitness"]), flush=True)
print("batches %d, pair records compared %d, identical %d, mismatches %d, %.1f s" % (nb, same + bad, same, bad, time.time() - t0))
sys.exit(1 if bad else 0)
