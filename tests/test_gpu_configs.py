"""GPU parity tests (-m gpu) on the BASELINE.json workloads the other files do not reach at size:

  C3  1024 x (10k x 10k) ModelNet40-scale pairs in ONE kss_icp_batch_dev call -- the > 32-pair and >= 64-pair code
      paths each C5 rank also runs (KSS_ICP.hpp:102-118 is the reference's own batch: the candidate ICPs);
  C4  one 1M x 1M pair, 2x scale + 60 deg: device pre-shape (initRegistrationKSS.hpp:144-220), device pose
      application, exact NN at 1M against the kd-tree oracle, fixed ICP iterations (KSS_ICP.hpp:133-183);
  the thresholds of the batched engine (33 and 64 pairs) and a ragged ~80-pair batch with the host pool on.

Bars: batch == one-pair path BIT FOR BIT per engine (T, iterations, state, fitness: the engines share one summation
order by construction); engines against each other and against the oracle within the existing 1e-5 (T) / 1e-9 (fitness)
bars, iteration counts and convergence states identical."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch_arrays(pairs):
    src_all = np.concatenate([p[0] for p in pairs]); tgt_all = np.concatenate([p[1] for p in pairs])
    so = np.concatenate([[0], np.cumsum([len(p[0]) for p in pairs])]).astype(np.int64)
    to = np.concatenate([[0], np.cumsum([len(p[1]) for p in pairs])]).astype(np.int64)
    return src_all, so, tgt_all, to


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _same_record(b, one):
    return (np.array_equal(np.array(b.T), np.array(one.T)) and b.iterations == one.iterations and b.state == one.state
            and b.converged == one.converged and b.fitness == one.fitness and b.last_mse == one.last_mse)


@pytest.fixture(scope="module")
def c3(pkg):
    """The C3 workload: 1024 pairs of 10k points, random axis, angle U(0, 15 deg), resident on the device."""
    import torch
    pairs = [pkg.synth.config_c3_pair(i, 10000) for i in range(1024)]
    src_all, so, tgt_all, to = _batch_arrays(pairs)
    return {"pairs": pairs, "so": so, "to": to, "d_src": _dev(torch, src_all), "d_tgt": _dev(torch, tgt_all)}


@pytest.mark.parametrize("mode", ["auto", "grid", "brute"])
def test_c3_batch_of_1024_equals_one_pair_path(ctx, pkg, c3, mode):
    """Every record of the 1024-pair batch equals the registration of that pair alone, bit for bit, on each engine."""
    nn = {"auto": pkg.NN_AUTO, "grid": pkg.NN_GRID, "brute": pkg.NN_BRUTE}[mode]
    for kw in (dict(), dict(max_iterations=20, fixed_iterations=1)):       # PCL convergence tests on / the bench's 20 fixed
        p = ctx.icp_params(nn_mode=nn, **kw)
        res = ctx.icp_batch_dev(c3["d_src"].data_ptr(), c3["so"], c3["d_tgt"].data_ptr(), c3["to"], p)
        bad = []
        step = 1 if (mode != "brute" and not kw) else 8                     # every pair once per engine family, a sample otherwise
        for i in range(0, 1024, step):
            one = ctx.icp_dev(c3["d_src"].data_ptr() + 12 * int(c3["so"][i]), 10000,
                              c3["d_tgt"].data_ptr() + 12 * int(c3["to"][i]), 10000, p)
            assert res[i].pair_id == i
            if not _same_record(res[i], one):
                bad.append((i, float(np.abs(np.array(res[i].T) - np.array(one.T)).max()), res[i].iterations, one.iterations))
        assert not bad, (mode, kw, len(bad), bad[:5])
        if kw:
            assert all(r.iterations == 20 and r.state == 1 for r in res)


def test_c3_engines_and_oracle_agree(ctx, O, pkg, c3):
    """Cell-list batch vs brute-force batch on all 1024 pairs, and a 40-pair sample against the oracle's PCL-style ICP."""
    g = ctx.icp_batch_dev(c3["d_src"].data_ptr(), c3["so"], c3["d_tgt"].data_ptr(), c3["to"], ctx.icp_params(nn_mode=pkg.NN_GRID))
    b = ctx.icp_batch_dev(c3["d_src"].data_ptr(), c3["so"], c3["d_tgt"].data_ptr(), c3["to"], ctx.icp_params(nn_mode=pkg.NN_BRUTE))
    for i in range(1024):
        assert g[i].iterations == b[i].iterations and g[i].state == b[i].state and g[i].converged == b[i].converged, i
        assert np.abs(np.array(g[i].T) - np.array(b[i].T)).max() < 1e-6, i
        assert abs(g[i].fitness - b[i].fitness) <= 1e-12 * max(1.0, b[i].fitness), i
    op = O.icp_params(nthreads=8)
    for i in list(range(0, 1024, 32)) + [1, 2, 3, 5, 8, 13, 1021, 1023]:
        s, t = c3["pairs"][i]
        r = O.icp(s, t, op)
        assert g[i].iterations == r["iterations"] and g[i].state == r["state"] and bool(g[i].converged) == r["converged"], i
        assert np.abs(g[i].matrix() - r["T"]).max() < 1e-5, i                # north star: 1e-4 rotation / 1e-3 translation
        assert abs(g[i].fitness - r["fitness"]) < 1e-9, i


@pytest.mark.parametrize("npairs", [32, 33, 63, 64])
def test_batch_size_thresholds(ctx, O, pkg, npairs):
    """32 / 33 pairs: last batch awaited by spinning on the published sums / first one past it; 63 / 64: the per-pair
    3x3 solves move from the calling thread to the host pool.  Same records on either side of each threshold."""
    S = pkg.synth
    pairs = [S.config_c3_pair(500 + i, 2000 + 37 * (i % 5)) for i in range(npairs)]
    src_all, so, tgt_all, to = _batch_arrays(pairs)
    for mode in (pkg.NN_GRID, pkg.NN_BRUTE):
        res = ctx.icp_batch(src_all, so, tgt_all, to, ctx.icp_params(nn_mode=mode))
        for i, (s, t) in enumerate(pairs):
            one = ctx.icp(s, t, ctx.icp_params(nn_mode=mode))
            assert res[i].pair_id == i and res[i].iterations == one["iterations"] and res[i].state == one["state"], (npairs, mode, i)
            assert np.array_equal(res[i].matrix(), one["T"]) and res[i].fitness == one["fitness"], (npairs, mode, i)
        for i in (0, npairs // 2, npairs - 1):
            r = O.icp(*pairs[i])
            assert res[i].iterations == r["iterations"] and res[i].state == r["state"]
            assert np.abs(res[i].matrix() - r["T"]).max() < 1e-5 and abs(res[i].fitness - r["fitness"]) < 1e-9


def test_ragged_batch_of_80_pairs_with_host_pool(ctx, pkg, O):
    """~80 ragged pairs (a few hundred to 2k points): a 1-point source, a displaced pair that needs the fallback; from 64
    pairs up the per-pair 3x3 solves run on the host pool (min(cores, 8) threads unless KSS_HOST_THREADS says otherwise;
    32 pairs per thread at least).  batch == one by one == oracle."""
    S = pkg.synth
    c = ctx
    rng = np.random.default_rng(11)
    pairs = []
    for i in range(80):
        nt = int(rng.integers(600, 2000)); ns = int(rng.integers(300, nt))
        R = S.rot_axis_angle(rng.normal(size=3), np.deg2rad(float(rng.uniform(1.0, 12.0))))
        pairs.append(S.make_pair(700 + i, nt, R=R, t=tuple(rng.normal(scale=0.01, size=3)), shape="bumpy", n_src=ns))
    pairs[17] = (pairs[3][0][:1].copy(), pairs[17][1])                       # one source point: not enough correspondences
    pairs[41] = (pairs[41][0] + np.float32(0.4), pairs[41][1])               # displaced: shells run out, brute-force fallback
    src_all, so, tgt_all, to = _batch_arrays(pairs)
    for mode in (pkg.NN_GRID, pkg.NN_BRUTE):
        res = c.icp_batch(src_all, so, tgt_all, to, c.icp_params(nn_mode=mode))
        for i, (s, t) in enumerate(pairs):
            one = c.icp(s, t, c.icp_params(nn_mode=pkg.NN_BRUTE if len(s) < 3 else mode))
            assert res[i].iterations == one["iterations"] and res[i].state == one["state"], (mode, i)
            assert np.abs(res[i].matrix() - one["T"]).max() < 1e-6 and abs(res[i].fitness - one["fitness"]) <= 1e-12 * max(1.0, one["fitness"]), (mode, i)
        assert res[17].state == 5 and res[17].iterations == 0
        for i in range(0, 80, 7):
            r = O.icp(*pairs[i])
            assert res[i].iterations == r["iterations"] and res[i].state == r["state"], (mode, i)
            assert np.abs(res[i].matrix() - r["T"]).max() < 1e-5 and abs(res[i].fitness - r["fitness"]) < 1e-9, (mode, i)


def test_non_finite_source_in_a_batch_is_contained(ctx, pkg):
    """A NaN source point matches nothing on EITHER engine of the batched path (the integer key compare of the cell-list
    fallback must not let NaN bits win): both engines register the remaining points identically."""
    S = pkg.synth
    pairs = [S.make_pair(900 + i, 3000, R=S.rot_axis_angle([0, 0.2, 1], np.deg2rad(6.0)), shape="bumpy") for i in range(4)]
    bad = pairs[2][0].copy(); bad[7] = np.nan; bad[100, 1] = np.nan
    clean = np.delete(pairs[2][0], [7, 100], axis=0)
    pairs[2] = (bad, pairs[2][1])
    src_all, so, tgt_all, to = _batch_arrays(pairs)
    kw = dict(max_iterations=6, fixed_iterations=1, compute_fitness=0)
    ref = ctx.icp(clean, pairs[2][1], ctx.icp_params(nn_mode=pkg.NN_BRUTE, **kw))
    for mode in (pkg.NN_GRID, pkg.NN_BRUTE):
        res = ctx.icp_batch(src_all, so, tgt_all, to, ctx.icp_params(nn_mode=mode, **kw))
        assert np.isfinite(res[2].matrix()).all() and res[2].iterations == 6, mode
        assert np.abs(res[2].matrix() - ref["T"]).max() < 1e-6, mode


# ---- C4 -----------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c4(pkg):
    src, tgt = pkg.synth.config_c4(1000000)
    return src, tgt


def test_c4_preshape_pose_and_nn_at_1m(ctx, O, pkg, c4):
    """C4 = 1M x 1M, source = 2 x R(60 deg about (1,1,1)) x target + t.  Device pre-shape statistics of both clouds vs the
    oracle (1e-11), scale 0.5; S' = device pose application (bit-exact f64); exact NN of S' in T at 1M vs the kd-tree
    oracle, bit for bit, on the cell list; both clouds in one call == one call per cloud."""
    import torch
    src, tgt = c4
    d_src = _dev(torch, src); d_tgt = _dev(torch, tgt)
    cS, rS = ctx.preshape_stats_dev(d_src.data_ptr(), pkg.binding.F32, len(src))
    cT, rT = ctx.preshape_stats_dev(d_tgt.data_ptr(), pkg.binding.F32, len(tgt))
    s64, t64 = src.astype(np.float64), tgt.astype(np.float64)
    ps = O.preshape_stats(s64, t64)
    rel = lambda a, b: np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max())
    assert rel(cS, ps.c_src) < 1e-11 and rel(cT, ps.c_tgt) < 1e-11 and rel(rS, ps.r_src) < 1e-11 and rel(rT, ps.r_tgt) < 1e-11
    scale = rT / rS
    assert abs(scale - 0.5) < 2e-3 and rel(scale, ps.scale) < 1e-11
    (c2S, r2S), (c2T, r2T) = ctx.preshape_stats_pair_dev(d_src.data_ptr(), len(src), d_tgt.data_ptr(), len(tgt), pkg.binding.F32)
    assert np.array_equal(c2S, cS) and r2S == rS and np.array_equal(c2T, cT) and r2T == rT
    # S' on the device, in f64 as the reference holds its clouds (vector<vector<double>>): shift, scale about c_T
    d_s64 = _dev(torch, s64)
    d_sp = torch.empty_like(d_s64)
    shift = [cT[k] - cS[k] for k in range(3)]
    pose = ctx.make_pose(shift, cT, scale, [0.0, 0.0, 0.0])
    ctx.pose_apply_dev(d_s64.data_ptr(), len(src), pose, d_sp.data_ptr())
    ctx.synchronize()
    ops = O.preshape_stats(s64, t64)
    ops.c_src[:] = list(cS); ops.c_tgt[:] = list(cT); ops.shift[:] = shift; ops.scale = scale      # the device's statistics
    assert np.array_equal(d_sp.cpu().numpy(), O.similarity_apply(s64, ops))
    # NN of the narrowed S' (initRegistrationKSS.hpp:440-442) in T, 1M x 1M
    d_spf = d_sp.to(torch.float32).contiguous()
    d_idx = torch.empty(len(src), dtype=torch.int32, device="cuda:0"); d_d2 = torch.empty(len(src), dtype=torch.float32, device="cuda:0")
    ctx.set_nn_mode(pkg.NN_GRID)
    try:
        ctx.nn_dev(d_spf.data_ptr(), len(src), d_tgt.data_ptr(), len(tgt), d_idx.data_ptr(), d_d2.data_ptr())
        ctx.synchronize()
    finally:
        ctx.set_nn_mode(pkg.NN_AUTO)
    oi, od = O.KdTree(tgt).nn(d_spf.cpu().numpy(), nthreads=16)
    assert np.array_equal(d_idx.cpu().numpy(), oi)
    assert np.array_equal(d_d2.cpu().numpy().view(np.uint32), od.view(np.uint32))


def test_c4_icp_iterations_at_1m(ctx, O, pkg, c4):
    """Three fixed ICP iterations + the fitness pass on a 1M x 1M pair inside ICP's basin (the C4 target, rotated by
    8 deg: after the KSS search the reference hands ICP such a pair) against the oracle's kd-tree ICP."""
    S = pkg.synth
    _, tgt = c4
    src, tgt = S.make_pair(0, 1000000, R=S.rot_axis_angle([1, 1, 1], np.deg2rad(8.0)), t=(0.01, -0.005, 0.02))
    got = ctx.icp(src, tgt, ctx.icp_params(max_iterations=3, fixed_iterations=1), trace_cap=4)
    ref = O.icp(src, tgt, O.icp_params(max_iterations=3, fixed_iterations=1, nthreads=16), trace_cap=4)
    assert got["iterations"] == 3 and np.array_equal(got["trace_sums"][:, 0], ref["trace_sums"][:, 0])
    assert np.allclose(got["trace_sums"], ref["trace_sums"], rtol=1e-9, atol=1e-12)
    assert np.abs(got["T"] - ref["T"]).max() < 1e-5 and abs(got["fitness"] - ref["fitness"]) < 1e-10


def test_batch_builds_with_wide_and_narrow_counters(ctx, pkg):
    """The per-pair cell-list build in LDS: 16-bit counters when no pair has 65536 points (grids up to 73k cells stay in
    LDS), 32-bit otherwise, the global-atomic path beyond.  Three batches that take the three routes; every record must
    equal the one-pair path bit for bit."""
    S = pkg.synth
    def batch(sizes, seed):
        pairs = [S.make_pair(seed + i, n, R=S.rot_axis_angle([0.3, 0.5, 1.0], np.deg2rad(4.0 + i)), t=(0.01 * i, 0.0, 0.005), shape="bumpy", n_src=n - 37 * (i + 1))
                 for i, n in enumerate(sizes)]
        src_all, so, tgt_all, to = _batch_arrays(pairs)
        p = ctx.icp_params(nn_mode=pkg.NN_GRID, max_iterations=6, fixed_iterations=1)
        res = ctx.icp_batch(src_all, so, tgt_all, to, p)
        for i, (s, t) in enumerate(pairs):
            one = ctx.icp(s, t, p)
            assert np.array_equal(res[i].matrix(), one["T"]) and res[i].fitness == one["fitness"] and res[i].last_mse == one["last_mse"], (sizes, i)
    batch([15000, 9000, 14000, 3000], 900)          # ~48k cells: LDS only with 16-bit counters
    batch([6000, 70000, 5000], 910)                 # a pair of >= 65536 points: 32-bit counters are not enough either -> global path
    batch([2500, 4000, 8000, 1200, 600], 920)       # small grids: LDS, 16-bit


def test_c5_leg_of_the_bench_on_one_rank_over_rccl():
    """BASELINE config 5 (8192 pairs over 8 GPUs + RCCL gather) cannot run on a one-GPU box; its per-rank code can: with
    KSS_BENCH_FORCE_DIST=1 bench.py creates a real RCCL process group of one rank and runs the C5 leg -- this rank's shard in one
    kss_icp_batch_dev call, ONE all-gather of the 96-byte records -- exactly as every rank of an 8-GPU job would."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"KSS_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--brute-steps", "0", "--no-cpu-baseline",
                        "--legs", "c5", "--c5-pairs", "96"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c5 = d["secondary"]["c5"]
    assert c5["n_gpus"] == 1 and c5["pairs_total"] == 96 and c5["result_iterations"] == 20
    assert c5["gathered_records_equal_local"] is True and c5["gathered_pair_ids_in_order"] is True
    assert c5["registrations_per_sec"] > 0 and c5["gather_us_max_over_ranks"] > 0
