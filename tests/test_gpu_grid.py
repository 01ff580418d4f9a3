"""GPU tests of the exact cell-list NN (KSS_NN_GRID): bit-identical to the brute-force sweep and to the
oracle on every input class, including the ones that force the brute-force list fallback."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def gctx(ctx, pkg):
    ctx.set_nn_mode(pkg.NN_GRID)
    yield ctx
    ctx.set_nn_mode(pkg.NN_AUTO)


def _check(gctx, O, s, t):
    idx, d2 = gctx.nn(s, t)
    if len(t) * len(s) <= 4e8:
        oi, od = O.nn_brute(s, t)
    else:
        oi, od = O.KdTree(t).nn(s, nthreads=8)
    assert np.array_equal(idx, oi)
    assert np.array_equal(d2.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("ns,nt", [(1, 1), (3, 2), (1000, 5), (257, 4097), (5000, 20000), (20000, 5000)])
def test_grid_nn_random_clouds(gctx, O, ns, nt):
    rng = np.random.default_rng(ns + 3 * nt)
    _check(gctx, O, rng.normal(size=(ns, 3)).astype(np.float32), rng.normal(size=(nt, 3)).astype(np.float32))


def test_grid_nn_surface_and_far_queries(gctx, O, pkg):
    S = pkg.synth
    tgt = S.bumpy(1, 30000).astype(np.float32)
    near = (S.bumpy(2, 8000) + S.normal(3, 24000).reshape(-1, 3) * 1e-3).astype(np.float32)
    _check(gctx, O, near, tgt)
    # rotated by 25 degrees: many queries need several shells, some fall back to the brute-force list
    R = S.rot_axis_angle([1, 1, 0], np.deg2rad(25.0))
    _check(gctx, O, (S.bumpy(2, 8000) @ R.T).astype(np.float32), tgt)
    # far outside the bbox (every query unresolved by the cell search) and straddling it
    _check(gctx, O, (S.bumpy(2, 3000) + np.array([7.0, -3.0, 2.0])).astype(np.float32), tgt)
    _check(gctx, O, (S.bumpy(2, 3000) * 3.0).astype(np.float32), tgt)
    # large coordinates: rounding slack scales with the magnitude
    off = np.array([1000.0, -2000.0, 500.0])
    _check(gctx, O, (S.bumpy(2, 3000) * 1.01 + off).astype(np.float32), (S.bumpy(1, 30000) + off).astype(np.float32))


def test_grid_nn_degenerate_targets(gctx, O):
    rng = np.random.default_rng(4)
    q = rng.normal(size=(2000, 3)).astype(np.float32)
    plane = rng.normal(size=(6000, 3)).astype(np.float32); plane[:, 2] = 0.25          # zero extent in z
    _check(gctx, O, q, plane)
    line = np.zeros((5000, 3), np.float32); line[:, 0] = np.linspace(-1, 1, 5000)      # zero extent in y and z
    _check(gctx, O, q, line)
    same = np.ones((4100, 3), np.float32) * 0.5                                         # all targets coincide
    _check(gctx, O, q, same)
    lattice = np.stack(np.meshgrid(*[np.arange(17, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    _check(gctx, O, lattice[::2] + np.float32(0.5), lattice)                            # exact ties everywhere
    dup = rng.normal(size=(5000, 3)).astype(np.float32); dup[2500:] = dup[:2500]        # every point twice
    _check(gctx, O, dup[::3], dup)


def test_grid_nn_full_size(gctx, O, pkg):
    src, tgt = pkg.synth.config_c2(100000)
    _check(gctx, O, src, tgt)


def test_grid_icp_bitwise_equals_brute(ctx, pkg):
    S = pkg.synth
    src, tgt = S.make_pair(31, 20000, R=S.rot_axis_angle([0.3, 0.2, 1.0], np.deg2rad(12.0)), t=(0.02, 0.0, -0.01), shape="bumpy")
    a = ctx.icp(src, tgt, ctx.icp_params(max_iterations=12, fixed_iterations=1, nn_mode=pkg.NN_BRUTE), trace_cap=16)
    b = ctx.icp(src, tgt, ctx.icp_params(max_iterations=12, fixed_iterations=1, nn_mode=pkg.NN_GRID), trace_cap=16)
    # same NN bits; the f64 sums are grouped differently (fused persistent workgroups vs 256*R chunks)
    assert np.array_equal(a["trace_sums"][:, 0], b["trace_sums"][:, 0])
    assert np.allclose(a["trace_sums"], b["trace_sums"], rtol=1e-12, atol=1e-12)
    assert np.abs(a["T"] - b["T"]).max() < 1e-6 and abs(a["fitness"] - b["fitness"]) < 1e-12
    c = ctx.icp(src, tgt, ctx.icp_params(nn_mode=pkg.NN_GRID))
    d = ctx.icp(src, tgt, ctx.icp_params(nn_mode=pkg.NN_BRUTE))
    assert c["iterations"] == d["iterations"] and np.abs(c["T"] - d["T"]).max() < 1e-6
    # run to run the grid path is bitwise reproducible (row-ordered final reduction)
    c2 = ctx.icp(src, tgt, ctx.icp_params(nn_mode=pkg.NN_GRID), trace_cap=64)
    c3 = ctx.icp(src, tgt, ctx.icp_params(nn_mode=pkg.NN_GRID), trace_cap=64)
    assert np.array_equal(c2["trace_sums"], c3["trace_sums"]) and np.array_equal(c2["T"], c3["T"])


def _tie_lattice_pair(n_side=24):
    """Targets on an integer lattice, sources on cell faces / edges / centres of that lattice (2, 4 and 8 targets at
    exactly the same distance) and slightly off them: exact ties across cells, the case the previous-winner bound must
    not prune (equal distance must still resolve to the lowest index)."""
    g = np.arange(n_side, dtype=np.float32)
    tgt = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) * np.float32(0.125)
    rng = np.random.default_rng(5)
    tgt = tgt[rng.permutation(len(tgt))]                     # index order unrelated to position
    base = tgt[rng.integers(0, len(tgt), 6000)]
    off = rng.integers(0, 2, size=(6000, 3)).astype(np.float32) * np.float32(0.0625)   # 0 or half a lattice step per axis
    src = base + off
    src[::3] += rng.normal(scale=1e-3, size=(2000, 3)).astype(np.float32)
    return src.astype(np.float32), tgt.astype(np.float32)


@pytest.mark.parametrize("case", ["bumpy", "ties", "duplicates", "c2"])
def test_grid_warm_started_search_is_bit_exact(ctx, O, pkg, case):
    """After the first iteration the cell search starts from each source's previous winner and skips the cells that
    bound excludes.  The per-source (index, d2) of the last pass must still equal a brute-force search bit for bit."""
    S = pkg.synth
    if case == "bumpy":
        src, tgt = S.make_pair(41, 30000, R=S.rot_axis_angle([0.2, 1.0, 0.1], np.deg2rad(9.0)), t=(0.01, -0.02, 0.0), shape="bumpy")
    elif case == "ties":
        src, tgt = _tie_lattice_pair()
    elif case == "duplicates":
        src, tgt = S.make_pair(42, 12000, R=S.rot_axis_angle([0, 0, 1], np.deg2rad(4.0)), shape="bumpy")
        tgt = np.concatenate([tgt, tgt[::2], tgt[::5]]).astype(np.float32)      # repeated points: equal distances, lower index wins
    else:
        src, tgt = S.config_c2(100000)
    tree = O.KdTree(tgt) if len(src) * len(tgt) > 4e8 else None
    for iters in (1, 2, 7):
        for mode in (pkg.NN_GRID, pkg.NN_BRUTE):
            r = ctx.icp(src, tgt, ctx.icp_params(max_iterations=iters, fixed_iterations=1, nn_mode=mode), fitness_corr=True)
            moved = O.transform_points_f32(r["T"], src)      # the final Matrix4f applied in float, as the kernels do
            oi, od = tree.nn(moved, nthreads=8) if tree else O.nn_brute(moved, tgt)
            assert np.array_equal(r["fitness_idx"], oi), (case, iters, mode, int((r["fitness_idx"] != oi).sum()))
            assert np.array_equal(r["fitness_d2"].view(np.uint32), od.view(np.uint32)), (case, iters, mode)


def test_grid_icp_with_unresolved_sources(ctx, O, pkg):
    """A source far from the target: the cell search gives up on most points in the first iterations, the
    brute-force list pass resolves them, later iterations stay inside the grid."""
    S = pkg.synth
    src, tgt = S.make_pair(32, 8000, R=S.rot_axis_angle([0, 1, 0], np.deg2rad(5.0)), t=(0.45, -0.3, 0.2), shape="bumpy")
    g = ctx.icp(src, tgt, ctx.icp_params(nn_mode=pkg.NN_GRID), trace_cap=200)
    r = O.icp(src, tgt, trace_cap=200)
    assert g["iterations"] == r["iterations"] and g["state"] == r["state"]
    assert np.array_equal(g["trace_sums"][:, 0], r["trace_sums"][:, 0])
    assert np.allclose(g["trace_sums"], r["trace_sums"], rtol=1e-9, atol=1e-12)
    assert np.abs(g["T"] - r["T"]).max() < 1e-5 and abs(g["fitness"] - r["fitness"]) < 1e-9


def test_grid_batch_matches_brute_batch_and_oracle(ctx, O, pkg):
    """Batched cell lists (one per pair): ragged pairs, a degenerate pair, far sources -- against the
    brute-force batch and the oracle."""
    S = pkg.synth
    specs = [(3000, 2500), (1200, 4000), (5000, 5000), (1, 2100), (2600, 2048), (700, 9000)]
    pairs = []
    for i, (ns, nt) in enumerate(specs):
        R = S.rot_axis_angle([0.2 * i, 1.0, 0.3], np.deg2rad(4.0 + 2.0 * i))
        s, t = S.make_pair(40 + i, nt, R=R, t=(0.01 * i, -0.02, 0.015), shape="bumpy", n_src=ns)
        pairs.append((s, t))
    pairs[4] = (pairs[4][0] + np.float32(0.35), pairs[4][1])      # displaced: needs several shells
    src_all = np.concatenate([p[0] for p in pairs]); tgt_all = np.concatenate([p[1] for p in pairs])
    so = np.cumsum([0] + [len(p[0]) for p in pairs]); to = np.cumsum([0] + [len(p[1]) for p in pairs])
    g = ctx.icp_batch(src_all, so, tgt_all, to, ctx.icp_params(nn_mode=pkg.NN_GRID))
    b = ctx.icp_batch(src_all, so, tgt_all, to, ctx.icp_params(nn_mode=pkg.NN_BRUTE))
    for i, (s, t) in enumerate(pairs):
        assert g[i].pair_id == i and g[i].iterations == b[i].iterations and g[i].state == b[i].state
        assert np.abs(g[i].matrix() - b[i].matrix()).max() < 1e-6
        assert abs(g[i].fitness - b[i].fitness) <= 1e-12 * max(1.0, abs(b[i].fitness))
        r = O.icp(s, t)
        assert g[i].iterations == r["iterations"] and g[i].state == r["state"]
        assert np.abs(g[i].matrix() - r["T"]).max() < 1e-5
    assert g[3].state == 5      # one source point: not enough correspondences
    # fixed-iteration mode, correspondences counted: bit-identical NN means identical counts every iteration
    g2 = ctx.icp_batch(src_all, so, tgt_all, to, ctx.icp_params(nn_mode=pkg.NN_GRID, max_iterations=4, fixed_iterations=1))
    b2 = ctx.icp_batch(src_all, so, tgt_all, to, ctx.icp_params(nn_mode=pkg.NN_BRUTE, max_iterations=4, fixed_iterations=1))
    for i in range(len(pairs)):
        assert np.abs(g2[i].matrix() - b2[i].matrix()).max() < 1e-6 and abs(g2[i].last_mse - b2[i].last_mse) < 1e-12


def test_non_finite_coordinates_are_contained(ctx, pkg):
    """Non-finite input must not hang or fault the search: a target with an infinite coordinate is rejected by the
    cell-list build, a NaN source point matches nothing (it is dropped from the sums) and the rest registers normally."""
    S = pkg.synth
    src, tgt = S.make_pair(61, 6000, R=S.rot_axis_angle([0, 0, 1], np.deg2rad(5.0)), shape="bumpy")
    bad_t = tgt.copy(); bad_t[17, 1] = np.inf
    with pytest.raises(pkg.KssError) as e:
        ctx.icp(src, bad_t, ctx.icp_params(nn_mode=pkg.NN_GRID, max_iterations=3))
    assert e.value.status == -1
    bad_s = src.copy(); bad_s[5] = np.nan
    ref = ctx.icp(np.delete(src, 5, axis=0), tgt, ctx.icp_params(nn_mode=pkg.NN_BRUTE, max_iterations=5, fixed_iterations=1), trace_cap=8)
    for mode in (pkg.NN_GRID, pkg.NN_BRUTE):
        r = ctx.icp(bad_s, tgt, ctx.icp_params(nn_mode=mode, max_iterations=5, fixed_iterations=1, compute_fitness=0), trace_cap=8)
        assert np.isfinite(r["T"]).all() and r["iterations"] == 5
        assert np.array_equal(r["trace_sums"][:, 0], ref["trace_sums"][:, 0])          # 5999 correspondences every pass
        assert np.abs(r["T"] - ref["T"]).max() < 1e-6


def test_batch_of_badly_posed_pairs_switches_engine_without_changing_results(ctx, pkg):
    """Sources that start far from their targets end in the batched kernel's in-wave brute-force fallback; the host then
    moves the call to the brute-force engine.  With the switch, without it (KSS_GRID_NOSWITCH) and on the brute-force
    engine from the start the transforms must agree."""
    import os
    S = pkg.synth
    npairs, n = 6, 3000
    src = np.empty((npairs * n, 3), np.float32); tgt = np.empty((npairs * n, 3), np.float32)
    for i in range(npairs):
        shift = 0.0 if i == 0 else 2.0          # one well-posed pair among badly posed ones
        s, t = S.make_pair(70 + i, n, R=S.rot_axis_angle([0.1, 1.0, 0.4], np.deg2rad(20.0)), t=(shift, -0.4 * shift, 0.3 * shift), shape="bumpy")
        src[i * n:(i + 1) * n] = s; tgt[i * n:(i + 1) * n] = t
    off = np.arange(npairs + 1, dtype=np.int64) * n
    kw = dict(max_iterations=12, fixed_iterations=1, max_corr_dist=100.0)
    a = ctx.icp_batch(src, off, tgt, off, ctx.icp_params(nn_mode=pkg.NN_GRID, **kw))
    os.environ["KSS_GRID_NOSWITCH"] = "1"
    try:
        b = ctx.icp_batch(src, off, tgt, off, ctx.icp_params(nn_mode=pkg.NN_GRID, **kw))
    finally:
        del os.environ["KSS_GRID_NOSWITCH"]
    c = ctx.icp_batch(src, off, tgt, off, ctx.icp_params(nn_mode=pkg.NN_BRUTE, **kw))
    for i in range(npairs):
        assert a[i].iterations == b[i].iterations == c[i].iterations == 12
        assert np.abs(a[i].matrix() - c[i].matrix()).max() < 1e-5 and np.abs(b[i].matrix() - c[i].matrix()).max() < 1e-5
        assert abs(a[i].fitness - c[i].fitness) < 1e-9 * max(1.0, c[i].fitness) and abs(b[i].fitness - c[i].fitness) < 1e-9 * max(1.0, c[i].fitness)


def test_soak_tools_short_run():
    """tools/soak_grid.py (the cell-list engine's per-source correspondences vs the brute-force engine over random shapes, sizes,
    poses and iteration counts: skip test, walks, fallbacks) and tools/soak_batch.py (every record of random ragged batches vs
    the one-pair path, bit for bit): a short run of each with fixed seeds; the tools exit non-zero on any mismatch."""
    import os, subprocess, sys
    from conftest import ROOT
    for tool, args in (("soak_grid.py", ["60", "2024"]), ("soak_batch.py", ["12", "2025"])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, tool + "\n" + r.stdout[-2000:] + r.stderr[-2000:]
        assert "Memory access fault" not in r.stdout + r.stderr


def test_canonical_wave_trees_on_the_device():
    """kss_device.hpp: wave_tree16 / wave_tree2 (permlane swaps + DPP) against a plain __shfl_xor restatement of the same
    binary tree, bit for bit (tools/tree_check.hip, built by __graft_entry__.build()).  Every engine sums through these trees,
    so a wrong tree makes all engines wrong ALIKE and no engine-against-engine test sees it -- an inline-asm form of the swaps
    once lost gfx950's wait state after a permlane swap and put every sum off by ~1e-8 relative."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "tree_check")
    assert os.path.exists(exe), "tools/tree_check not built: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout + r.stderr
