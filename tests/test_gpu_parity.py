"""GPU parity tests (-m gpu): every call goes through the C-ABI of libkssicp.so and is compared with
the CPU oracle on the same seeded inputs, with the committed golden vectors, and -- at full benchmark
sizes -- through size-independent properties.

Bars: bit-exact for indices and f32 squared distances (same no-fma arithmetic); f64 reductions within
1e-12 relative (summation order differs: tree vs serial); ICP transform within BASELINE.json's north-star
tolerance (1e-4 rotation, 1e-3 translation) -- in practice ~1e-6."""
import os

import numpy as np
import pytest

from conftest import ROOT, GOLDEN

pytestmark = pytest.mark.gpu

REL = 1e-12


def _close(a, b, rel=REL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= rel * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b))))


# ---- (a8) NN --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ns,nt", [(1, 1), (1, 300), (257, 1), (1000, 1000), (2000, 2048), (5000, 777), (10000, 10000)])
def test_nn_bit_exact(ctx, O, ns, nt):
    rng = np.random.default_rng(ns * 7 + nt)
    s = rng.normal(size=(ns, 3)).astype(np.float32)
    t = rng.normal(size=(nt, 3)).astype(np.float32)
    idx, d2 = ctx.nn(s, t)
    oi, od = O.nn_brute(s, t)
    assert np.array_equal(idx, oi)
    assert np.array_equal(d2.view(np.uint32), od.view(np.uint32))


def test_nn_ties_take_lowest_index(ctx, O):
    rng = np.random.default_rng(9)
    t = rng.normal(size=(3000, 3)).astype(np.float32)
    t[1500] = t[10]; t[2999] = t[10]; t[700] = t[699]      # duplicates across sub-tiles, tiles and splits
    s = np.concatenate([t[[10, 699, 2999]], rng.normal(size=(500, 3)).astype(np.float32)])
    idx, d2 = ctx.nn(s, t)
    assert idx[0] == 10 and idx[1] == 699 and idx[2] == 10 and d2[0] == 0
    oi, od = O.nn_brute(s, t)
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)
    # a lattice has massive exact ties
    g = np.stack(np.meshgrid(*[np.arange(12, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    q = g[::3] + np.float32(0.5)
    idx, d2 = ctx.nn(q, g)
    oi, od = O.nn_brute(q, g)
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)


def test_nn_golden_vector(ctx):
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    idx, d2 = ctx.nn(g["g4_src"], g["g4_tgt"])
    assert np.array_equal(idx, g["g4_nn_idx"]) and np.array_equal(d2, g["g4_nn_d2"])


def test_nn_full_size_properties(ctx, pkg, O):
    """100k x 100k (config C2 size): self-query is the identity with zero distance; a shuffled +
    jittered copy maps back to the permutation; the kd-tree oracle agrees bit for bit."""
    src, tgt = pkg.synth.config_c2(100000)
    idx, d2 = ctx.nn(tgt, tgt)
    assert np.array_equal(idx, np.arange(len(tgt), dtype=np.int32)) and not d2.any()
    idx, d2 = ctx.nn(src, tgt)
    oi, od = O.KdTree(tgt).nn(src, nthreads=8)
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)


def test_nn_rejects_bad_arguments(ctx, pkg):
    with pytest.raises(pkg.KssError) as e:
        ctx.nn(np.zeros((0, 3), np.float32), np.zeros((5, 3), np.float32))
    assert e.value.status == -1
    with pytest.raises(pkg.KssError):
        ctx.nn(np.zeros((5, 3), np.float32), np.zeros((0, 3), np.float32))


# ---- (a10) covariance sums ----------------------------------------------------------------------------------
def test_cov_matches_oracle_sums(ctx, O):
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    s, t = g["g4_src"], g["g4_tgt"]
    sums = ctx.cov(s, t, g["g4_nn_idx"], 1.0)
    assert _close(sums, g["g4_trace_sums"][0])          # first ICP iteration of the golden trace
    assert sums[0] == g["g4_trace_sums"][0][0]
    # max distance gate: only correspondences with d2 <= max_d2 are kept, [17],[18] always sum all
    thr = float(np.median(g["g4_nn_d2"]))
    sums2 = ctx.cov(s, t, g["g4_nn_idx"], thr)
    assert sums2[0] == np.sum(g["g4_nn_d2"].astype(np.float64) <= thr)
    assert _close(sums2[17], sums[17]) and _close(sums2[18], sums[18])
    # linearity: sums of a concatenation = sum of sums
    a = ctx.cov(s[:700], t, g["g4_nn_idx"][:700], 1.0)
    b = ctx.cov(s[700:], t, g["g4_nn_idx"][700:], 1.0)
    assert _close(a + b, sums, 1e-11)


# ---- (a2) pre-shape --------------------------------------------------------------------------------------------
def test_preshape_matches_oracle(ctx, O):
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    src64, tgt64 = g["g1_src"].astype(np.float64), g["g1_tgt"].astype(np.float64)
    cS, rS = ctx.preshape_stats(src64)
    cT, rT = ctx.preshape_stats(tgt64)
    st = g["g1_stats"]
    assert _close(cS, st[0:3]) and _close(cT, st[3:6]) and _close(rS, st[9]) and _close(rT, st[10])
    assert _close(rT / rS, st[11])
    cS32, rS32 = ctx.preshape_stats(g["g1_src"])         # f32 input widened on load
    assert _close(cS32, st[0:3]) and _close(rS32, st[9])


def test_preshape_large_and_ragged(ctx, O, pkg):
    for n in (1, 255, 257, 100003):
        p = (pkg.synth.bumpy(n, n) * 3.0 + np.array([5.0, -2.0, 0.5]))
        c, r = ctx.preshape_stats(p)
        ps = O.preshape_stats(p, p)
        assert _close(c, ps.c_src, 1e-11) and _close(r, ps.r_src, 1e-11)


def test_preshape_f32_vector_path(ctx, O, pkg):
    """f32 clouds >= 1024 points take the float4 streaming kernels: every tail length (3n mod 4, n mod 3)."""
    for n in (1024, 1025, 1026, 1027, 4099, 100003):
        p = (pkg.synth.bumpy(n + 1, n) * 2.0 + np.array([3.0, -1.0, 0.25])).astype(np.float32)
        c, r = ctx.preshape_stats(p)
        ps = O.preshape_stats(p.astype(np.float64), p.astype(np.float64))
        assert _close(c, ps.c_src, 1e-11) and _close(r, ps.r_src, 1e-11)


# ---- (a3,a7) pose application ----------------------------------------------------------------------------------------
def test_pose_apply_bit_exact(ctx, O):
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    src64 = g["g1_src"].astype(np.float64)
    st = g["g1_stats"]
    pose = ctx.make_pose(st[6:9], st[3:6], st[11], [0.7875, 5.5125, 3.15])
    out = ctx.pose_apply(src64, pose)
    assert np.array_equal(out, g["g3_pose"])            # same f64 operations, same order, no fma
    pose0 = ctx.make_pose(st[6:9], st[3:6], st[11], [0, 0, 0])
    assert np.array_equal(ctx.pose_apply(src64, pose0), g["g1_preshaped"])


def test_transform_apply_matches_reference_expression(ctx):
    rng = np.random.default_rng(1)
    P = rng.normal(size=(1000, 3))
    T = np.eye(4, dtype=np.float32); T[:3, :3] = rng.normal(size=(3, 3)).astype(np.float32); T[:3, 3] = [0.1, 0.2, 0.3]
    out = ctx.transform_apply(T, P)
    Td = T.astype(np.float64)
    exp = np.stack([((Td[r, 0] * P[:, 0] + Td[r, 1] * P[:, 1]) + Td[r, 2] * P[:, 2]) + Td[r, 3] for r in range(3)], 1)
    assert np.array_equal(out, exp)                      # KSS_ICP.hpp:224-230 evaluated left to right


# ---- (a4) rotation search ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("step", [6, 8])
def test_rotation_search_matches_oracle(ctx, pkg, step):
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    err = ctx.rotation_search(g["g1_preshaped"], g["g1_tgt"].astype(np.float64), step)
    ref = g["g2_value_%d" % step]
    assert err.shape == ref.shape
    assert _close(err, ref, 1e-12)
    best, alist = pkg.rotation_candidates(err, step)
    assert np.array_equal(best, g["g2_angle_%d" % step])
    assert np.array_equal(alist, g["g2_list_%d" % step])


@pytest.mark.parametrize("step", [12, 16])
def test_rotation_search_steps_12_and_16_on_the_device(ctx, O, pkg, step):
    """initRegistrationKSS.hpp:245-296 at the other step counts the reference uses: 12 (KSSICP_Registration_Additional,
    KSS_ICP.hpp:371-379: g = 12) and 16 (g = 17, from the double accumulation of 6.3 / 16).  Error volume to 1e-12, identical
    arg-min and angleList, with every candidates-per-lane width of rot_search_kernel (KSS_ROT_S: read at each launch)."""
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    Sp, T = g["g1_preshaped"], g["g1_tgt"].astype(np.float64)
    ref = O.rotation_search(Sp, T, step)
    assert ref["g"] == {12: 12, 16: 17}[step]
    old = os.environ.get("KSS_ROT_S")
    try:
        for width in (None, "1", "2", "4"):
            if width is None:
                os.environ.pop("KSS_ROT_S", None)
            else:
                os.environ["KSS_ROT_S"] = width
            err = ctx.rotation_search(Sp, T, step)
            assert err.shape == ref["value"].shape, width
            assert _close(err, ref["value"], 1e-12), width
            assert np.unravel_index(np.argmin(err), err.shape) == np.unravel_index(np.argmin(ref["value"]), err.shape), width
            best, alist = pkg.rotation_candidates(err, step)
            assert np.array_equal(best, ref["angle"]), width
            assert np.array_equal(alist, ref["angle_list"]), width
    finally:
        if old is None:
            os.environ.pop("KSS_ROT_S", None)
        else:
            os.environ["KSS_ROT_S"] = old


# ---- (a9) ICP ---------------------------------------------------------------------------------------------------------
def test_icp_trace_matches_oracle(ctx, O):
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    r = ctx.icp(g["g4_src"], g["g4_tgt"], trace_cap=64)
    assert r["iterations"] == int(g["g4_iters"][0]) and int(r["converged"]) == int(g["g4_iters"][1])
    assert r["state"] == int(g["g4_iters"][2])
    n = len(g["g4_trace_sums"])
    assert len(r["trace_sums"]) == n
    assert np.array_equal(r["trace_sums"][:, 0], g["g4_trace_sums"][:, 0])      # correspondence counts
    assert _close(r["trace_sums"], g["g4_trace_sums"], 1e-9)
    assert np.abs(r["trace_Tk"] - g["g4_trace_Tk"]).max() < 1e-6
    assert np.abs(r["T"][:3, :3] - g["g4_T"][:3, :3]).max() < 1e-4              # north-star tolerance
    assert np.abs(r["T"][:3, 3] - g["g4_T"][:3, 3]).max() < 1e-3
    assert np.abs(r["T"] - g["g4_T"]).max() < 5e-6                              # what we actually reach
    assert abs(r["fitness"] - g["g4_fitness"][0]) < 1e-9


@pytest.mark.parametrize("ns,nt", [(3, 50), (900, 1300), (4000, 2500)])
def test_icp_matches_oracle_ragged(ctx, O, pkg, ns, nt):
    S = pkg.synth
    src, tgt = S.make_pair(ns + nt, nt, R=S.rot_axis_angle([0.1, 0.9, 0.2], np.deg2rad(9.0)), t=(0.01, 0.02, -0.01),
                           shape="bumpy", n_src=ns)
    got = ctx.icp(src, tgt)
    ref = O.icp(src, tgt)
    assert got["iterations"] == ref["iterations"] and got["converged"] == ref["converged"] and got["state"] == ref["state"]
    assert np.abs(got["T"] - ref["T"]).max() < 1e-5
    assert abs(got["fitness"] - ref["fitness"]) < 1e-9


def test_icp_fixed_iterations_and_fma_mode(ctx, O, pkg):
    S = pkg.synth
    src, tgt = S.make_pair(21, 3000, R=S.rot_axis_angle([0, 0, 1], np.deg2rad(10.0)))
    p = ctx.icp_params(max_iterations=7, fixed_iterations=1)
    got = ctx.icp(src, tgt, p)
    ref = O.icp(src, tgt, O.icp_params(max_iterations=7, fixed_iterations=1))
    assert got["iterations"] == 7 and ref["iterations"] == 7 and got["state"] == 1
    assert np.abs(got["T"] - ref["T"]).max() < 1e-5
    p = ctx.icp_params(max_iterations=7, fixed_iterations=1, nn_fma=1)
    got = ctx.icp(src, tgt, p)
    ref = O.icp(src, tgt, O.icp_params(max_iterations=7, fixed_iterations=1, fma=1))
    assert np.abs(got["T"] - ref["T"]).max() < 1e-5


def test_icp_not_enough_correspondences(ctx, O):
    src = np.array([[10, 0, 0], [0, 10, 0], [0, 0, 10], [10, 10, 10]], np.float32)      # all farther than 1.0
    tgt = np.zeros((8, 3), np.float32) + np.arange(8, dtype=np.float32)[:, None] * 0.01
    got = ctx.icp(src, tgt)
    ref = O.icp(src, tgt)
    assert not got["converged"] and got["state"] == 5 and got["iterations"] == 0
    assert ref["state"] == 5 and not ref["converged"]
    assert np.array_equal(got["T"], np.eye(4, dtype=np.float32))
    assert abs(got["fitness"] - ref["fitness"]) < 1e-9 * ref["fitness"]


def test_icp_tuning_knobs_do_not_change_results(ctx, pkg):
    S = pkg.synth
    src, tgt = S.make_pair(5, 6000, R=S.rot_axis_angle([0, 1, 1], np.deg2rad(7.0)), shape="bumpy")
    base = ctx.icp(src, tgt, ctx.icp_params(max_iterations=5, fixed_iterations=1, nn_mode=pkg.NN_BRUTE), trace_cap=8)
    for spt, split in [(1, 1), (2, 3), (4, 7), (8, 2)]:
        r = ctx.icp(src, tgt, ctx.icp_params(max_iterations=5, fixed_iterations=1, nn_sources_per_thread=spt,
                                             nn_target_splits=split, nn_mode=pkg.NN_BRUTE), trace_cap=8)
        assert np.array_equal(r["trace_sums"], base["trace_sums"])      # bitwise: exact NN + fixed-order sums
        assert np.array_equal(r["T"], base["T"])


def test_icp_batch_equals_singles(ctx, pkg):
    S = pkg.synth
    pairs = [S.config_c3_pair(i, n) for i, n in enumerate([1500, 800, 2300, 1, 1024])]
    pairs[3] = (pairs[0][0][:1], pairs[3][1])         # a 1-point source: < 3 correspondences
    src_all = np.concatenate([p[0] for p in pairs]); tgt_all = np.concatenate([p[1] for p in pairs])
    so = np.cumsum([0] + [len(p[0]) for p in pairs]); to = np.cumsum([0] + [len(p[1]) for p in pairs])
    res = ctx.icp_batch(src_all, so, tgt_all, to)
    for i, (s, t) in enumerate(pairs):
        one = ctx.icp(s, t)
        assert res[i].pair_id == i
        assert res[i].iterations == one["iterations"] and bool(res[i].converged) == one["converged"]
        assert np.array_equal(res[i].matrix(), one["T"])
        assert res[i].fitness == one["fitness"]
    assert res[3].state == 5


def test_icp_config_c2_shape(ctx, O, pkg):
    """Config C2 inputs (100k x 100k sphere, R_z(10 deg)), a few fixed iterations, vs the kd-tree oracle."""
    src, tgt = pkg.synth.config_c2(100000)
    got = ctx.icp(src, tgt, ctx.icp_params(max_iterations=3, fixed_iterations=1), trace_cap=4)
    ref = O.icp(src, tgt, O.icp_params(max_iterations=3, fixed_iterations=1, nthreads=8), trace_cap=4)
    assert np.array_equal(got["trace_sums"][:, 0], ref["trace_sums"][:, 0])
    assert _close(got["trace_sums"], ref["trace_sums"], 1e-9)
    assert np.abs(got["T"] - ref["T"]).max() < 1e-5
    assert abs(got["fitness"] - ref["fitness"]) < 1e-10


# ---- PCR_QM and the orchestrated registration ------------------------------------------------------------------------------
def test_pcr_qm_matches_oracle(ctx, O, ref_pairs):
    S, T = ref_pairs[("registration", "Horse")]
    got = ctx.pcr_qm(S, T)
    ref = O.pcr_qm(S, T)
    assert _close(got, ref, 1e-12)


@pytest.mark.parametrize("key", [("registration", "Bunny"), ("registration", "Horse"), ("registration_scale", "Bunny")])
def test_register_matches_oracle_on_reference_pairs(ctx, O, ref_pairs, key):
    S, T = ref_pairs[key]
    got = ctx.register(S, T, S, 8.0, 1000)
    ref = O.kssicp_register(S, T, S, 8.0, 1000)
    assert got["grid"] == 9 and got["n_angle_list"] == ref["n_angle_list"]
    assert got["used_angle_list"] == ref["used_angle_list"] and got["angle_index"] == ref["angle_index"]
    assert np.array_equal(got["angle"], ref["angle"])
    assert _close(got["scale"], ref["scale"])
    assert np.abs(got["R"] - ref["R"]).max() < 1e-4 and np.abs(got["t"] - ref["t"]).max() < 1e-3    # north star
    assert np.abs(got["R"] - ref["R"]).max() < 1e-5
    assert abs(got["E_d_init"] - ref["E_d_init"]) < 1e-8 and abs(got["final_fitness"] - ref["final_fitness"]) < 1e-8
    assert np.abs(got["pointAlign"] - ref["pointAlign"]).max() < 1e-4


# ---- C-ABI argument checking and the RCCL record gather -----------------------------------------------------------
def test_abi_rejects_bad_arguments(ctx, pkg):
    import ctypes as C
    L = pkg.load_library()
    a = np.zeros((4, 3), np.float32)
    out = np.zeros(8, np.float64)
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    assert L.kss_nn(ctx.h, None, 4, vp(a), 4, None, None) == -1                  # null cloud
    assert L.kss_nn(ctx.h, vp(a), -1, vp(a), 4, None, None) == -1                # negative size
    assert L.kss_preshape_stats(ctx.h, vp(a), 7, 4, vp(out), C.byref(C.c_double())) == -1      # bad dtype
    assert L.kss_preshape_stats(ctx.h, vp(a), 0, 0, vp(out), C.byref(C.c_double())) == -1      # empty cloud
    idx = np.array([0, 1, 2, 9], np.int32)
    assert L.kss_cov(ctx.h, vp(a), vp(a), vp(idx), 4, 4, 1.0, vp(np.zeros(20))) == -1          # index out of range
    g = C.c_int(0)
    d = np.zeros((4, 3), np.float64)
    assert L.kss_rotation_search(ctx.h, vp(d), 4, vp(d), 4, 8.0, vp(np.zeros(10)), 10, C.byref(g)) == -5   # err buffer too small
    assert L.kss_ctx_set_nn_mode(ctx.h, 7) == -1
    assert b"" != L.kss_last_error(ctx.h)
    assert L.kss_ctx_create(99, C.byref(C.c_void_p())) == -1                      # no such device
    # maximum sizes: more points than 32-bit indexing allows are refused before anything is touched (device-pointer entry
    # points: the pointers below are never dereferenced)
    res = pkg.IcpResult()
    p = ctx.icp_params()
    assert L.kss_nn_dev(ctx.h, vp(a), 3 * 10**9, vp(a), 10, None, None) == -1
    assert L.kss_icp_dev(ctx.h, vp(a), 10, vp(a), 2**31 + 5, C.byref(p), C.byref(res)) == -1
    assert b"too large" in L.kss_last_error(ctx.h)
    assert L.kss_knn_dev(ctx.h, vp(a), 3 * 10**9, vp(a), 10, 4, vp(a), vp(a)) == -1
    assert L.kss_rigid_from_sums(vp(np.zeros(20)), vp(np.zeros(16, np.float32))) == -1         # zero correspondences


def _single_rank_comm():
    """A real ncclComm_t with one rank, made through librccl directly (the same library dlopen("librccl.so") finds)."""
    import ctypes as C
    import torch  # noqa: F401  (loads torch's librccl so that dlopen("librccl.so") resolves to the same library)
    rccl = None
    for name in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
        try:
            rccl = C.CDLL(name)
            break
        except OSError:
            continue
    if rccl is None:
        import os, torch as _t
        rccl = C.CDLL(os.path.join(os.path.dirname(_t.__file__), "lib", "librccl.so"))

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    return rccl, comm


def test_gather_results_over_rccl_single_rank(ctx, pkg):
    """kss_gather_results with a real ncclComm_t (one rank: the collective degenerates to a copy, but the dlopen of
    librccl, the symbol lookup, the staging and the stream handling are the ones N ranks use)."""
    import ctypes as C
    rccl, comm = _single_rank_comm()
    n = 5
    local = (pkg.IcpResult * n)()
    for i in range(n):
        local[i].pair_id = i; local[i].fitness = 0.25 * i; local[i].iterations = 10 + i
        for k in range(16):
            local[i].T[k] = float(i * 16 + k)
    allr = (pkg.IcpResult * n)()
    L = pkg.load_library()
    rc = L.kss_gather_results(ctx.h, comm, 1, C.cast(local, C.c_void_p), n, C.cast(allr, C.c_void_p))
    assert rc == 0, L.kss_last_error(ctx.h)
    for i in range(n):
        assert allr[i].pair_id == i and allr[i].fitness == 0.25 * i and allr[i].iterations == 10 + i and allr[i].T[7] == float(i * 16 + 7)
    rccl.ncclCommDestroy(comm)


def test_split_source_icp_two_ranks_on_one_gpu(pkg):
    """SURVEY 8e, the single-pair exchange step: source rows split over ranks, target replicated, the 20 sums summed
    over the ranks after every pass.  Two contexts on this GPU play the two ranks (one thread each); the all-reduce
    callback is a barrier-and-add between the threads, standing in for RCCL.  Both ranks must end with the same
    transform, and it must match the unsharded registration."""
    import ctypes as C
    import threading
    S = pkg.synth
    src, tgt = S.make_pair(51, 40000, R=S.rot_axis_angle([0.2, 0.1, 1.0], np.deg2rad(7.0)), t=(0.01, 0.02, -0.01), shape="bumpy")
    world = 2
    ctxs = [pkg.Context(0) for _ in range(world)]
    bar = threading.Barrier(world)
    slots = [None] * world

    def make_cb(rank):
        def cb(_user, values, n):
            try:
                slots[rank] = np.ctypeslib.as_array(values, shape=(int(n),)).copy()
                bar.wait(timeout=60)
                tot = slots[0] + slots[1]                 # fixed order: both ranks get the same bits
                bar.wait(timeout=60)
                np.ctypeslib.as_array(values, shape=(int(n),))[:] = tot
                return 0
            except Exception:
                return -1
        return pkg.binding.ALLREDUCE_FN(cb)

    out = [None] * world

    def run(rank):
        lo, hi = pkg.shard.shard_range(len(src), world, rank)
        cb = make_cb(rank)
        out[rank] = pkg.shard.icp_split_source(ctxs[rank], pkg.binding, src[lo:hi], tgt, cb, nn_mode=pkg.NN_GRID)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert all(o is not None for o in out)
    ref = ctxs[0].icp(src, tgt, ctxs[0].icp_params(nn_mode=pkg.NN_GRID))
    assert np.array_equal(out[0]["T"], out[1]["T"]) and out[0]["iterations"] == out[1]["iterations"] and out[0]["fitness"] == out[1]["fitness"]
    assert out[0]["iterations"] == ref["iterations"] and out[0]["converged"] == ref["converged"]
    assert np.abs(out[0]["T"] - ref["T"]).max() < 2e-6 and abs(out[0]["fitness"] - ref["fitness"]) < 1e-9
    for c in ctxs:
        c.close()


def test_split_source_icp_over_rccl_single_rank(ctx, pkg):
    """The RCCL-backed callback (kss_rccl_allreduce_sum) with a real one-rank communicator: ncclAllReduce of the sums
    on the context's stream every iteration; one rank's sum is itself, so the result equals the plain registration."""
    import ctypes as C
    rccl, comm = _single_rank_comm()
    S = pkg.synth
    src, tgt = S.make_pair(52, 20000, R=S.rot_axis_angle([0, 0, 1], np.deg2rad(6.0)), shape="bumpy")
    L = pkg.load_library()
    link = pkg.binding.RcclLink(ctx.h, comm)
    p = ctx.icp_params()
    p.allreduce = C.cast(L.kss_rccl_allreduce_sum, pkg.binding.ALLREDUCE_FN)
    p.allreduce_user = C.cast(C.pointer(link), C.c_void_p)
    a = ctx.icp(src, tgt, p)
    b = ctx.icp(src, tgt, ctx.icp_params())
    assert np.array_equal(a["T"], b["T"]) and a["iterations"] == b["iterations"] and a["fitness"] == b["fitness"]
    rccl.ncclCommDestroy(comm)


def test_register_batch_equals_one_by_one(ctx, pkg, ref_pairs):
    """kss_register_batch: many full registrations (down-sample, pre-shape, search, candidate ICPs) spread over worker
    contexts on the same GPU.  Every pair must come out exactly as the one-pair path gives it, whatever the worker
    count; ragged sizes, a scaled pair and the reference's own pairs."""
    S = pkg.synth
    clouds = []
    for i, (n, scale, deg) in enumerate(((2400, 1.0, 25.0), (3100, 2.0, 60.0), (1800, 0.5, 140.0), (2600, 1.0, 10.0), (1200, 1.3, 200.0))):
        s, t = S.make_pair(300 + i, n, R=S.rot_axis_angle([0.4, 0.2 + 0.1 * i, 1.0], np.deg2rad(deg)), scale=scale, t=(0.1 * i, -0.05, 0.02), shape="bumpy", n_src=n - 150)
        clouds.append((s.astype(np.float64), t.astype(np.float64)))
    for key in (("registration", "Bunny"), ("registration", "Horse")):
        clouds.append(ref_pairs[key])
    src_all = np.concatenate([c[0] for c in clouds]); tgt_all = np.concatenate([c[1] for c in clouds])
    so = np.concatenate([[0], np.cumsum([len(c[0]) for c in clouds])]); to = np.concatenate([[0], np.cumsum([len(c[1]) for c in clouds])])
    one = []
    for s, t in clouds:
        m = min(min(len(s), len(t)) // 2, 2000)
        ss, _ = ctx.downsample_aivs(s, m); tt, _ = ctx.downsample_aivs(t, m)
        one.append(ctx.register(ss, tt, s, 8.0, 1000))
    for workers in (1, 3, 8):
        res, align = ctx.register_batch(src_all, so, tgt_all, to, sample_cap=2000, accurate=8.0, iters=1000, workers=workers, want_align=True)
        for i, (r, o) in enumerate(zip(res, one)):
            assert r.scale == o["scale"] and list(r.angle) == list(o["angle"]) and r.final_fitness == o["final_fitness"], (workers, i)
            assert np.array_equal(np.array(r.R).reshape(3, 3), o["R"]) and np.array_equal(np.array(r.t), o["t"])
            assert r.icp_iterations == o["icp_iterations"] and r.n_angle_list == o["n_angle_list"] and r.angle_index == o["angle_index"]
            assert np.array_equal(align[so[i]:so[i + 1]], o["pointAlign"])
    with pytest.raises(pkg.KssError):
        ctx.register_batch(src_all, np.array([0, 0, len(src_all)]), tgt_all, np.array([0, 10, len(tgt_all)]))      # an empty source


def test_gated_launches_give_the_same_registration(ctx, pkg):
    """The mechanisms that only move work around must not move a bit: gated launches (KSS_GATED=0 off), host stores into
    device memory (KSS_GATE_BAR=0 off: the re-publishing gate), chained launches (KSS_CHAIN=0: one launch per pass) and
    the skip test of the fused pass (KSS_SKIN=-1: every source searches in every pass; 0.05 / 1.0: other skins); nor may a
    lost answer (the waiting kernel gives up, the pass is launched again).  Same
    registration, bit for bit, including runs that converge early (the waiting kernel is cancelled), fixed-length
    runs, a pair that needs the brute-force fallback, and a 100k pair (196 workgroups: the chained form of C2)."""
    import subprocess, sys, json
    code = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth; ctx = pkg.Context(0)
out = []
for seed, n, deg, t in ((81, 30000, 8.0, (0.01, 0.0, 0.0)), (82, 12000, 25.0, (0.3, -0.2, 0.1)), (83, 100000, 10.0, (0.0, 0.0, 0.0))):
    src, tgt = S.make_pair(seed, n, R=S.rot_axis_angle([0.2, 0.1, 1.0], np.deg2rad(deg)), t=t, shape="bumpy")
    for kw in (dict(), dict(max_iterations=9, fixed_iterations=1), dict(max_iterations=30, fixed_iterations=1)):
        r = ctx.icp(src, tgt, ctx.icp_params(nn_mode=pkg.NN_GRID, **kw))
        out.append([r["T"].tolist(), r["iterations"], r["fitness"], r["last_mse"]])
print("RESULT" + json.dumps(out))
""" % ROOT
    res = {}
    variants = {"default": {}, "ungated": {"KSS_GATED": "0"}, "no_bar": {"KSS_GATE_BAR": "0"}, "unchained": {"KSS_CHAIN": "0"},
                "no_skip": {"KSS_SKIN": "-1"}, "thin_skin": {"KSS_SKIN": "0.05"}, "thick_skin": {"KSS_SKIN": "1.0"},
                # an answer that never arrives (a stalled host thread): the waiting kernel's bounded poll runs out, it leaves
                # without having touched anything, and the pass is launched again as a plain one
                "lost_answer": {"KSS_TEST_DROP_GATE": "7", "KSS_GATE_POLLS": "20000"},
                "lost_answer_unchained": {"KSS_TEST_DROP_GATE": "5", "KSS_GATE_POLLS": "20000", "KSS_CHAIN": "0"},
                # a 16-byte granule seen TORN: the host's gate record first arrives with a granule whose words
                # do not fit its check word; a result slot is first stored with bits that do not fit its check word.  The
                # receiver polls again instead of taking it for data
                "torn_gate": {"KSS_TEST_TORN_GATE": "6"}, "torn_gate_unchained": {"KSS_TEST_TORN_GATE": "4", "KSS_CHAIN": "0"},
                "torn_slot": {"KSS_TEST_TORN_SLOT": "4"}, "torn_slot_late": {"KSS_TEST_TORN_SLOT": "23"},
                # the set-up's two forms: the target's bounding box by copy + synchronisation instead of host-memory granules
                "box_by_copy": {"KSS_BBOX_HOST": "0"}}
    for name, extra in variants.items():
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, name + r.stdout + r.stderr
        res[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0][6:])
        assert ("launched again" in r.stderr) == name.startswith("lost_answer"), name   # (the hook did fire, and only there)
    for name in variants:
        assert res[name] == res["default"], name


def test_lost_answer_with_more_rows_than_compute_units(pkg):
    """A 150k-point pair has 293 rows for 256 compute units: when the host's answer to a gated launch is lost, the first
    workgroups leave at the gate but later ones may still run.  The recovery re-arms every zero-at-rest counter
    before the plain relaunch; the registration must come out as without the loss, bit for bit."""
    import subprocess, sys, json
    code = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth; ctx = pkg.Context(0)
src, tgt = S.make_pair(91, 150000, R=S.rot_axis_angle([0.2, 0.1, 1.0], np.deg2rad(7.0)), t=(0.01, 0.0, -0.01), shape="bumpy")
out = []
for kw in (dict(max_iterations=8, fixed_iterations=1), dict()):
    r = ctx.icp(src, tgt, ctx.icp_params(nn_mode=pkg.NN_GRID, **kw))
    out.append([r["T"].tolist(), r["iterations"], r["fitness"], r["last_mse"]])
print("RESULT" + json.dumps(out))
""" % ROOT
    res = {}
    for name, extra in {"default": {}, "lost": {"KSS_TEST_DROP_GATE": "3", "KSS_GATE_POLLS": "20000"},
                        "lost_late": {"KSS_TEST_DROP_GATE": "9", "KSS_GATE_POLLS": "20000"}}.items():
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, **extra))
        assert r.returncode == 0, name + r.stdout + r.stderr
        res[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0][6:])
        assert ("launched again" in r.stderr) == (name != "default"), name
    assert res["lost"] == res["default"] and res["lost_late"] == res["default"]


def test_large_pair_rows_tagged_or_ticketed_same_bits(pkg):
    """Pairs with more rows than compute units (300k points: 586 rows) hand their rows over as tagged granules to a polling
    reducer since round 3 (before: store + ticket, the last workgroup adds 586 rows); the two forms, gated or not, must give
    the same registration bit for bit -- the additions are the same in the same order."""
    import subprocess, sys, json
    code = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth; ctx = pkg.Context(0)
src, tgt = S.make_pair(93, 300000, R=S.rot_axis_angle([0.2, 0.1, 1.0], np.deg2rad(6.0)), t=(0.01, 0.0, -0.01), shape="bumpy")
out = []
for kw in (dict(max_iterations=7, fixed_iterations=1), dict()):
    r = ctx.icp(src, tgt, ctx.icp_params(nn_mode=pkg.NN_GRID, **kw))
    out.append([r["T"].tolist(), r["iterations"], r["fitness"], r["last_mse"]])
print("RESULT" + json.dumps(out))
""" % ROOT
    res = {}
    for name, extra in {"default": {}, "ticket": {"KSS_TAGGED_ROWS_MAX": "256"}, "ticket_all": {"KSS_TAGGED_ROWS": "0"},
                        "ungated": {"KSS_GATED": "0"}, "ungated_ticket": {"KSS_GATED": "0", "KSS_TAGGED_ROWS_MAX": "256"},
                        "library_scan": {"KSS_SCAN_LIB": "1"},   # (the cell table's scan by rocPRIM instead of kss_grid.hip's own: same starts)
                        "box_by_copy": {"KSS_BBOX_HOST": "0"}}.items():
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, **extra))
        assert r.returncode == 0, name + r.stdout + r.stderr
        res[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0][6:])
    for name in res:
        assert res[name] == res["default"], name


def test_a_new_context_never_takes_an_old_contexts_rows(pkg):
    """Rows handed over as {bits, launch number} granules are accepted on the number alone: a context that gets
    a destroyed context's workspace block back must not take its rows for its own.  Launch numbers are unique per process
    and the row buffer is zeroed when (re)allocated: same-shaped registrations on successive contexts equal the same
    registration on a context of their own."""
    S = pkg.synth
    pairs = [S.make_pair(600 + k, 60000, R=S.rot_axis_angle([0.3, 1.0, 0.2], np.deg2rad(5.0 + k)), t=(0.01 * k, 0.0, 0.0), shape="bumpy") for k in range(3)]
    p_kw = dict(nn_mode=pkg.NN_GRID, max_iterations=12, fixed_iterations=1)
    chained = []
    for src, tgt in pairs:                   # A registers and dies, B gets its blocks back, ...
        c = pkg.Context(0)
        chained.append(c.icp(src, tgt, c.icp_params(**p_kw)))
        c.close()
    keep = pkg.Context(0)                    # (a live context in between changes what the allocator hands out)
    for (src, tgt), r in zip(pairs, chained):
        c = pkg.Context(0)
        one = c.icp(src, tgt, c.icp_params(**p_kw))
        c.close()
        assert np.array_equal(r["T"], one["T"]) and r["fitness"] == one["fitness"] and r["last_mse"] == one["last_mse"]
    keep.close()


def test_growing_registrations_on_one_context(pkg):
    """Work buffers that are zero at rest (cell counters, tickets, the fallback list's length) across registrations of
    growing and shrinking size on ONE context -- the sequence that once left the tail of a re-allocated counter buffer
    uninitialised (incidents).  KSS_COUNTS_CHECK makes the library verify the counters before every build;
    every result must equal the same registration on a fresh context."""
    S = pkg.synth
    os.environ["KSS_COUNTS_CHECK"] = "1"
    try:
        ctx = pkg.Context(0)
        sizes = [(700, 1500), (2400, 4300), (900, 1100), (5200, 9000), (3000, 3100), (12000, 20000), (600, 700), (30000, 41000)]
        shapes = ["bumpy", "sphere", "bumpy", "sphere", "bumpy", "bumpy", "sphere", "bumpy"]
        got = []
        for k, ((ns, nt), shape) in enumerate(zip(sizes, shapes)):
            src, tgt = S.make_pair(500 + k, nt, R=S.rot_axis_angle([0.3, 1.0, 0.2], np.deg2rad(6.0 + 3 * k)), t=(0.02 * k, 0.0, -0.01), shape=shape, n_src=ns)
            got.append((src, tgt, ctx.icp(src, tgt, ctx.icp_params(nn_mode=pkg.NN_GRID))))
        ctx.close()
        for src, tgt, r in got:
            fresh = pkg.Context(0)
            one = fresh.icp(src, tgt, fresh.icp_params(nn_mode=pkg.NN_GRID))
            fresh.close()
            assert np.array_equal(r["T"], one["T"]) and r["iterations"] == one["iterations"] and r["fitness"] == one["fitness"]
    finally:
        del os.environ["KSS_COUNTS_CHECK"]
