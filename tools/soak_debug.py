#!/usr/bin/env python3
"""Replays one case of tools/soak_grid.py (same RNG stream) and prints both engines' per-iteration traces."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
ctx = pkg.Context(0)
want, seed = int(sys.argv[1]), int(sys.argv[2])
src_txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "soak_grid.py")).read()
ns_ = {}
pre = src_txt[src_txt.index("def cloud("):src_txt.index("bad = 0")]
rng = np.random.default_rng(seed)
exec(pre, {"np": np, "rng": rng}, ns_)
cloud, rot = ns_["cloud"], ns_["rot"]
for case in range(want + 1):
    kind = rng.choice(["sphere", "aniso", "clusters", "plane", "lattice", "cube"])
    nt = int(rng.choice([1500, 5000, 20000, 60000, 150000]))
    ns = int(rng.choice([1100, 4000, 30000, 100000]))
    tgt = cloud(kind, nt)
    pick = rng.integers(0, len(tgt), ns)
    ang = float(rng.choice([0.0, 0.02, 0.2, 1.0]))
    R = rot(rng.normal(size=3), ang)
    t = rng.normal(size=3) * float(rng.choice([0.0, 0.01, 0.3]))
    jit = float(rng.choice([0.0, 1e-4, 1e-2]))
    src = (tgt[pick].astype(np.float64) @ R.T + t + rng.normal(size=(ns, 3)) * jit).astype(np.float32)
    runs = []
    for iters in (int(rng.integers(1, 4)), int(rng.integers(4, 12))):
        runs.append((iters, float(rng.choice([1.0, 0.05, 10.0]))))
    if case != want:
        continue
    print("case", case, kind, "ns", ns, "nt", len(tgt), "ang", ang, "t", t, "jit", jit, "runs", runs)
    for iters, mcd in runs:
        kw = dict(max_iterations=iters, fixed_iterations=1, max_corr_dist=mcd)
        a = ctx.icp(src, tgt, ctx.icp_params(nn_mode=pkg.NN_GRID, **kw), trace_cap=16, fitness_corr=True)
        b = ctx.icp(src, tgt, ctx.icp_params(nn_mode=pkg.NN_BRUTE, **kw), trace_cap=16, fitness_corr=True)
        o = O.icp(src, tgt, O.icp_params(max_iterations=iters, fixed_iterations=1, max_corr_dist=mcd), trace_cap=16)
        print(" iters", iters, "max_corr", mcd, "state", a["state"], b["state"], "its", a["iterations"], b["iterations"], o["iterations"])
        for k in range(len(a["trace_sums"])):
            sa, sb, so = a["trace_sums"][k], b["trace_sums"][k], o["trace_sums"][k] if k < len(o["trace_sums"]) else None
            print("  it %d: n grid %d brute %d oracle %s | rel diff sums grid-brute %.2e | dTk %.2e" % (
                k, sa[0], sb[0], int(so[0]) if so is not None else "-", np.abs(sa[:19] - sb[:19]).max() / max(1.0, np.abs(sb[:19]).max()),
                np.abs(a["trace_Tk"][k] - b["trace_Tk"][k]).max()))
        print("  final |dT| grid-brute %.3e, grid-oracle %.3e, brute-oracle %.3e" % (np.abs(a["T"] - b["T"]).max(), np.abs(a["T"] - o["T"]).max(), np.abs(b["T"] - o["T"]).max()))
        print("  sums it0 grid :", a["trace_sums"][0][:8]); print("  sums it0 brute:", b["trace_sums"][0][:8])
