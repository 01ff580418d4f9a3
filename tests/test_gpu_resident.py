"""GPU tests (-m gpu) of the pair-resident batch engine (kss-icp_amd/csrc/kss_resident.hip): one workgroup per pair for the
whole registration -- the batch KSS_ICP.hpp:102-118 runs one candidate at a time, configs C3 / C5.

Bars: resident == launch-per-pass engine == one-pair path BIT FOR BIT (T, iterations, state, fitness, last_mse, per-source
correspondences of the fitness pass): the engines share the exact search and one summation order; the oracle within the
existing 1e-5 / 1e-9.  Edge cases: a 1-point source, a NaN source, sources far outside the target (the in-LDS sweep), lattice
targets (ties settled by the original index), duplicate targets, a pair too large for a CU (the whole batch then runs on the
other engine), an unanswered workgroup (the call falls back), a gate record seen torn."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CODE = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth; ctx = pkg.Context(0)
rng = np.random.default_rng(5)
pairs = []
for i in range(40):
    nt = int(rng.integers(700, 9000)); ns = int(rng.integers(300, 9000))
    R = S.rot_axis_angle(rng.normal(size=3), np.deg2rad(float(rng.uniform(1.0, 14.0))))
    pairs.append(S.make_pair(900 + i, nt, R=R, t=tuple(rng.normal(scale=0.01, size=3)), shape="bumpy" if i %% 3 else "sphere", n_src=ns))
pairs[3] = (pairs[3][0][:1].copy(), pairs[3][1])                          # one source point
s = pairs[5][0].copy(); s[7] = np.nan; s[100, 1] = np.inf; pairs[5] = (s, pairs[5][1])   # non-finite sources match nothing
s = pairs[8][0].copy(); s[:40] += np.float32(3.0); pairs[8] = (s, pairs[8][1])      # far outside the target: the sweep
pairs[11] = (pairs[11][0] + np.float32(0.35), pairs[11][1])               # displaced: many shells
gx = np.stack(np.meshgrid(*[np.arange(12, dtype=np.float32) / 12 - 0.5] * 3, indexing="ij"), -1).reshape(-1, 3)
pairs[13] = ((gx[::2] + np.float32(1.0 / 24)).copy(), gx.copy())          # lattice targets, sources at cell centres: 8-way ties
t = pairs[17][1]; pairs[17] = (pairs[17][0], np.concatenate([t, t[:500], t[:200]]))   # duplicate targets
src_all = np.concatenate([p[0] for p in pairs]); tgt_all = np.concatenate([p[1] for p in pairs])
so = np.concatenate([[0], np.cumsum([len(p[0]) for p in pairs])]).astype(np.int64)
to = np.concatenate([[0], np.cumsum([len(p[1]) for p in pairs])]).astype(np.int64)
out = []
for kw in (dict(), dict(max_iterations=7, fixed_iterations=1), dict(max_iterations=25, fixed_iterations=1, compute_fitness=0)):
    res = ctx.icp_batch(src_all, so, tgt_all, to, ctx.icp_params(nn_mode=pkg.NN_GRID, **kw))
    out.append([[list(r.T), r.iterations, r.state, r.converged, r.fitness, r.last_mse] for r in res])
print("RESULT" + json.dumps(out))
""" 


def _run(extra_env, timeout=600):
    code = _CODE % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **extra_env))
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0][6:]), r.stderr


def test_resident_equals_launch_per_pass_engine_on_a_ragged_batch():
    """40 ragged pairs with every edge case the reference's inputs can hold, three parameter sets: the resident engine, the
    launch-per-pass engine (KSS_RESIDENT=0) and the resident engine with other skip margins give the same records bit for
    bit; so do a run whose gate record is once seen torn and a run in which one workgroup is never answered in time (the
    call then starts over on the other engine and says so)."""
    base, err = _run({})
    assert "launch-per-pass" not in err
    off, _ = _run({"KSS_RESIDENT": "0"})
    assert off == base
    for env in ({"KSS_SKIN": "-1"}, {"KSS_SKIN": "1.0"}, {"KSS_HOST_THREADS": "1"}, {"KSS_HOST_THREADS": "5"}, {"KSS_TEST_TORN_RES_GATE": "9"}):
        got, err = _run(env)
        assert got == base, env
        assert "launch-per-pass" not in err, env
    got, err = _run({"KSS_TEST_RES_STALL": "30", "KSS_GATE_POLLS": "3000"})
    assert got == base
    assert "launch-per-pass" in err


def test_resident_batch_equals_one_pair_path_and_oracle(ctx, pkg, O):
    """Each record of a resident batch equals the registration of that pair alone (single-pair cell-list engine), bit for bit,
    and the oracle's PCL-style ICP within the usual bars; the fitness pass's per-source correspondences equal the exact NN of
    the transformed cloud."""
    S = pkg.synth
    pairs = [S.config_c3_pair(40 + i, 2500 + 611 * (i % 7)) for i in range(24)]
    src_all = np.concatenate([p[0] for p in pairs]); tgt_all = np.concatenate([p[1] for p in pairs])
    so = np.concatenate([[0], np.cumsum([len(p[0]) for p in pairs])]).astype(np.int64)
    to = np.concatenate([[0], np.cumsum([len(p[1]) for p in pairs])]).astype(np.int64)
    for kw in (dict(), dict(max_iterations=12, fixed_iterations=1)):
        res = ctx.icp_batch(src_all, so, tgt_all, to, ctx.icp_params(nn_mode=pkg.NN_GRID, **kw))
        for i, (s, t) in enumerate(pairs):
            one = ctx.icp(s, t, ctx.icp_params(nn_mode=pkg.NN_GRID, **kw))
            assert res[i].pair_id == i and res[i].iterations == one["iterations"] and res[i].state == one["state"], i
            assert np.array_equal(res[i].matrix(), one["T"]) and res[i].fitness == one["fitness"] and res[i].last_mse == one["last_mse"], i
        for i in (0, 7, 23):
            r = O.icp(*pairs[i], O.icp_params(**kw))
            assert res[i].iterations == r["iterations"] and res[i].state == r["state"]
            assert np.abs(res[i].matrix() - r["T"]).max() < 1e-5 and abs(res[i].fitness - r["fitness"]) < 1e-9
    # the optional outputs of a batch call: pair 0's per-iteration trace and the correspondences getFitnessScore() sums over
    import ctypes as C
    p = ctx.icp_params(nn_mode=pkg.NN_GRID)
    n0 = len(pairs[0][0])
    fi = np.full(n0, -1, np.int32); fd = np.full(n0, np.nan, np.float32)
    sums = np.zeros((64, pkg.NSUMS), np.float64); tk = np.zeros((64, 16), np.float32); n = C.c_int(0)
    p.fitness_idx = fi.ctypes.data_as(C.POINTER(C.c_int32)); p.fitness_d2 = fd.ctypes.data_as(C.POINTER(C.c_float))
    p.trace_sums = sums.ctypes.data_as(C.POINTER(C.c_double)); p.trace_Tk = tk.ctypes.data_as(C.POINTER(C.c_float)); p.trace_cap = 64; p.trace_n = C.pointer(n)
    res = ctx.icp_batch(src_all, so, tgt_all, to, p)
    one = ctx.icp(pairs[0][0], pairs[0][1], ctx.icp_params(nn_mode=pkg.NN_GRID), trace_cap=64, fitness_corr=True)
    assert n.value == len(one["trace_sums"]) == res[0].iterations
    assert np.array_equal(sums[:n.value], one["trace_sums"]) and np.array_equal(tk[:n.value].reshape(-1, 4, 4), one["trace_Tk"])
    assert np.array_equal(fi, one["fitness_idx"]) and np.array_equal(fd, one["fitness_d2"])
    moved = ctx.transform_apply_f32(res[0].matrix(), pairs[0][0])
    oi, od = O.nn_brute(moved, pairs[0][1])
    assert np.array_equal(fi, oi) and np.array_equal(fd, od)


def test_a_pair_too_large_for_a_cu_sends_the_batch_to_the_other_engine(ctx, pkg):
    """ns > 10240 (or a target that does not fit the LDS) is not resident material: same API, same bits, other engine."""
    S = pkg.synth
    pairs = [S.config_c3_pair(70 + i, n) for i, n in enumerate((3000, 12000, 2500, 4000))]
    src_all = np.concatenate([p[0] for p in pairs]); tgt_all = np.concatenate([p[1] for p in pairs])
    so = np.concatenate([[0], np.cumsum([len(p[0]) for p in pairs])]).astype(np.int64)
    to = np.concatenate([[0], np.cumsum([len(p[1]) for p in pairs])]).astype(np.int64)
    res = ctx.icp_batch(src_all, so, tgt_all, to, ctx.icp_params(nn_mode=pkg.NN_GRID))
    for i, (s, t) in enumerate(pairs):
        one = ctx.icp(s, t, ctx.icp_params(nn_mode=pkg.NN_GRID))
        assert np.array_equal(res[i].matrix(), one["T"]) and res[i].fitness == one["fitness"] and res[i].iterations == one["iterations"]


_SPLIT_CODE = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth; ctx = pkg.Context(0)
rng = np.random.default_rng(17)
pairs = []
for i in range(640):
    nt = int(rng.integers(300, 2200)); ns = int(rng.integers(200, 2200))
    R = S.rot_axis_angle(rng.normal(size=3), np.deg2rad(float(rng.uniform(0.5, 16.0))))
    pairs.append(S.make_pair(1900 + i, nt, R=R, t=tuple(rng.normal(scale=0.01, size=3)), shape="bumpy" if i %% 3 else "sphere", n_src=min(ns, nt)))
pairs[7] = (pairs[7][0][:1].copy(), pairs[7][1])                          # one source point
s = pairs[9][0].copy(); s[3] = np.nan; pairs[9] = (s, pairs[9][1])        # a non-finite source
pairs[11] = (pairs[11][0] + np.float32(40.0), pairs[11][1])               # nothing within reach of max_corr_dist: no correspondences
pairs[13] = (pairs[13][1][:500].copy(), pairs[13][1])                     # already aligned: converges at once
src_all = np.concatenate([p[0] for p in pairs]); tgt_all = np.concatenate([p[1] for p in pairs])
so = np.concatenate([[0], np.cumsum([len(p[0]) for p in pairs])]).astype(np.int64)
to = np.concatenate([[0], np.cumsum([len(p[1]) for p in pairs])]).astype(np.int64)
out = []
for kw in (dict(), dict(max_iterations=9, fixed_iterations=1), dict(max_iterations=12, compute_fitness=0, max_corr_dist=0.5)):
    res = ctx.icp_batch(src_all, so, tgt_all, to, ctx.icp_params(nn_mode=pkg.NN_GRID, **kw))
    out.append([[list(r.T), r.iterations, r.state, r.converged, r.fitness, r.last_mse] for r in res])
print("RESULT" + json.dumps(out))
"""


def test_split_batch_equals_one_launch():
    """More pairs than the device runs at once (640 ragged pairs on 256 compute units): the batch runs as two launches, every
    pair's first pass, then the rest longest-first, the pairs' registers resting in memory in between.  Same records as ONE
    launch (KSS_RESIDENT_SPLIT=0), as a split after passes 1 and 3, and as the launch-per-pass engine, bit for bit -- with pairs
    that end before, at and after the split, pairs without correspondences, and the fitness pass on either side of it."""
    def run(env):
        r = subprocess.run([sys.executable, "-c", _SPLIT_CODE % ROOT], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout + r.stderr
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0][6:]), r.stderr
    base, err = run({})
    assert "launch-per-pass" not in err
    for env in ({"KSS_RESIDENT_SPLIT": "0"}, {"KSS_RESIDENT_SPLIT": "2"}, {"KSS_RESIDENT_SPLIT": "4"}, {"KSS_RESIDENT_SPLIT": "4", "KSS_RESIDENT_SPLIT_KEY": "mse"},
                {"KSS_RESIDENT": "0"}):
        got, err = run(env)
        assert got == base, env
        assert "launch-per-pass" not in err, env
    # an answer held back in the first launch (the 300th) and in the second (the 2500th): the call starts over on the other engine
    for stall in ("300", "2500"):
        got, err = run({"KSS_TEST_RES_STALL": stall, "KSS_GATE_POLLS": "3000"})
        assert got == base, stall
        assert "launch-per-pass" in err, stall
