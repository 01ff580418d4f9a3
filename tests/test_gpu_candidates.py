"""GPU tests (-m gpu) of the candidate batch of KSSICP_Registration (KSS_ICP.hpp:86-131) on its resident engine
(kss-icp_amd/csrc/kss_kernels.hip: cand_resident_kernel): the judge ICP (:92-93) and the candidate ICPs of the angle list
(:102-118) run as ONE launch whose workgroups stay on the chip for the whole registration; when the judge ends at or below
the threshold of :99 the candidates are told to stop.

Bar: every form of the path gives the SAME registration bit for bit -- resident launch, one launch per pass
(KSS_CAND_RESIDENT=0), judge and candidates as two sequential calls (KSS_REGISTER_SPEC=0), several tiles per workgroup and a
capacity too small for the batch (KSS_CAND_CAP), one host thread, a gate record seen torn, a workgroup nobody answers (the
call starts over on the launch-per-pass form and says so).  The two-launch form of round 2 (KSS_CAND_FUSED=0) adds the same
correspondences in another order: equal within 1e-6.  The oracle comparison of the same entry point is
tests/test_gpu_parity.py::test_register_matches_oracle_on_reference_pairs and tests/test_gpu_refdata.py."""
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
out = []
def reg(s, t, full):
    r = ctx.register(s, t, full, 8.0, 1000)
    out.append({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in r.items()})
R30 = S.rot_axis_angle([0.3, 0.2, 1.0], np.deg2rad(30.0))
src, tgt = S.make_pair(6243, 2000, R=R30, t=(0.05, -0.02, 0.03), shape="bumpy")      # judge fails: the candidates decide
src = src.astype(np.float64); tgt = tgt.astype(np.float64)
t_, _ = ctx.downsample_aivs(tgt, 1000); s_, _ = ctx.downsample_aivs(src, 1000)
reg(s_, t_, src)
src, tgt = S.make_pair(5, 900, R=R30, t=(0.05, -0.02, 0.03), shape="bumpy")          # judge good enough: the candidates are stopped
reg(src.astype(np.float64), tgt.astype(np.float64), src.astype(np.float64))
src, tgt = S.make_pair(77, 517, R=S.rot_axis_angle([1.0, 0.1, 0.2], np.deg2rad(70.0)), t=(0.1, 0.0, 0.0), shape="bumpy", n_src=333)   # ragged tiles
reg(src.astype(np.float64), tgt.astype(np.float64), src.astype(np.float64))
src, tgt = S.make_pair(6, 70, R=R30, shape="bumpy")                                  # three tiles per candidate
reg(src.astype(np.float64), tgt.astype(np.float64), src.astype(np.float64))
src, tgt = S.make_pair(9, 1999, R=S.rot_axis_angle([0.1, 1.0, 0.3], np.deg2rad(120.0)), shape="bumpy", n_src=33)   # few sources, many targets
reg(src.astype(np.float64), tgt.astype(np.float64), src.astype(np.float64))
print("RESULT" + json.dumps(out))
"""


def _run(extra_env, timeout=600):
    r = subprocess.run([sys.executable, "-c", _CODE % ROOT], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **extra_env))
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0][6:]), r.stderr


def test_every_form_of_the_candidate_batch_gives_the_same_registration():
    base, err = _run({})
    assert "launch-per-pass" not in err
    assert [r["used_angle_list"] for r in base][:2] == [1, 0]      # (both branches of KSS_ICP.hpp:99 are in the set)
    for env in ({"KSS_CAND_RESIDENT": "0"}, {"KSS_REGISTER_SPEC": "0"}, {"KSS_REGISTER_SPEC": "0", "KSS_CAND_RESIDENT": "0"},
                {"KSS_CAND_CAP": "300"}, {"KSS_CAND_CAP": "130"}, {"KSS_CAND_CAP": "40"}, {"KSS_HOST_THREADS": "1"},
                {"KSS_CAND_PAIRS_PER_THREAD": "4"}, {"KSS_TEST_TORN_RES_GATE": "7"}, {"KSS_GATE_BAR": "0"}):
        got, err = _run(env)
        assert got == base, env
        assert "launch-per-pass" not in err, env
    # a workgroup that is not answered in time leaves; the call starts over on the launch-per-pass form
    got, err = _run({"KSS_TEST_RES_STALL": "12", "KSS_GATE_POLLS": "3000"})
    assert got == base
    assert "launch-per-pass" in err
    # round 2's sweep + reduce: the same correspondences, added in another order
    old, _ = _run({"KSS_CAND_FUSED": "0", "KSS_CAND_RESIDENT": "0", "KSS_REGISTER_SPEC": "0"})
    for a, b in zip(old, base):
        assert a["angle_index"] == b["angle_index"] and a["used_angle_list"] == b["used_angle_list"] and a["n_angle_list"] == b["n_angle_list"]
        assert a["icp_iterations"] == b["icp_iterations"]
        assert np.abs(np.array(a["R"]) - np.array(b["R"])).max() < 1e-6 and np.abs(np.array(a["t"]) - np.array(b["t"])).max() < 1e-6


def test_many_registrations_in_a_row_leave_nothing_behind(ctx, pkg):
    """Stopped candidates may leave a pass half done (rows of some of a candidate's workgroups only): rows carry the number of
    their launch and pass, so nothing of it can be taken for a later registration's.  Alternating registrations whose candidates are stopped, run to the end, stopped ... on ONE
    context must each equal the same registration on a fresh context."""
    S = pkg.synth
    R30 = S.rot_axis_angle([0.3, 0.2, 1.0], np.deg2rad(30.0))
    cases = []
    for seed, n, sub in ((5, 900, 0), (6243, 2000, 1000), (6, 70, 0), (6244, 1600, 700), (5, 900, 0), (13, 400, 0)):
        src, tgt = S.make_pair(seed, n, R=R30, t=(0.05, -0.02, 0.03), shape="bumpy")
        src = src.astype(np.float64); tgt = tgt.astype(np.float64)
        if sub:   # two different samplings of the surface: the judge's fitness stays above the threshold
            tgt = ctx.downsample_aivs(tgt, sub)[0]; src = ctx.downsample_aivs(src, sub)[0]
        cases.append((src, tgt))
    seq = [ctx.register(s, t, s, 8.0, 1000) for s, t in cases]
    assert len({r["used_angle_list"] for r in seq}) == 2           # (both kinds occur)
    for (s, t), r in zip(cases, seq):
        c2 = pkg.Context(0)
        f = c2.register(s, t, s, 8.0, 1000)
        c2.close()
        for k in ("R", "t", "T_icp", "pointAlign"):
            assert np.array_equal(np.asarray(r[k]), np.asarray(f[k])), k
        assert r["icp_iterations"] == f["icp_iterations"] and r["E_d_init"] == f["E_d_init"] and r["final_fitness"] == f["final_fitness"]


def test_many_workers_never_stall_each_other():
    """kss_register_batch with more workers than the device can keep resident launches for: every launch reserves its
    workgroups out of the device's capacity (the rest take more tiles per workgroup or the launch-per-pass form), no serving
    thread calls HIP while it has pairs to answer, and a kernel that is gone is recognised by its exit flags.  Round 3 saw a
    registration in ~200 lose a second to a workgroup nobody answered before that: the message must not appear, and the
    records must equal the one-worker run."""
    code = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth; ctx = pkg.Context(0)
npairs, n = 48, 6000
cl = []
for i in range(npairs):
    s, t = S.make_pair(400 + i, n, R=S.rot_axis_angle([0.3, 0.1 + 0.01 * i, 1.0], np.deg2rad(5.0 + 5.0 * (i %% 30))), scale=1.0 + 0.02 * (i %% 7), shape="bumpy")
    cl.append((s.astype(np.float64), t.astype(np.float64)))
src_all = np.concatenate([c[0] for c in cl]); tgt_all = np.concatenate([c[1] for c in cl])
off = np.arange(npairs + 1, dtype=np.int64) * n
out = []
for w in (1, 8, 16, 8, 16, 12, 16):
    r = ctx.register_batch(src_all, off, tgt_all, off, workers=w)
    out.append([[list(x.R), list(x.t), x.icp_iterations, x.angle_index, x.final_fitness] for x in r])
print("RESULT" + json.dumps(out))
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "left before every pair" not in r.stderr, r.stderr
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0][6:])
    for k in range(1, len(res)):
        assert res[k] == res[0], k


def test_registrations_under_load_equal_the_solitary_one():
    """One context registers the same pair 300 times while five threads register other pairs on contexts of their own
    (tools/load_repro.py).  Round 3: with rows handed over by per-candidate ticket counters, one registration in ten differed
    from the solitary result under exactly this load; rows are now handed over by launch-and-pass tags.
    Every registration must give the solitary run's bits."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "load_repro.py"), "300", "register", "5"], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "300 registrations, 0 differed" in r.stdout, r.stdout[-2000:]
    assert "left before every pair" not in r.stderr, r.stderr[-2000:]
