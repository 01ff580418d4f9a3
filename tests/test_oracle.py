"""CPU tests of the oracle: pinned against the reference's own data fixtures (tests/golden/ref_data,
copied data files of PS_AIS_Simplification/data/registration*/) and against committed golden vectors.

Pin status: first-party arithmetic is pinned by the fixtures' known rotations (transfer.txt); the ICP
restatement of PCL 1.8.1 is "parity unpinned" numerically (PCL is absent from the reference tree and
this image; the reference holds no numeric golden output for it)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _axis_R(axis, ang):
    c, s = np.cos(ang), np.sin(ang)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]),
            "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]


def _transfer_table():
    """PS_AIS_Simplification/data/registration/transfer.txt: '<name> <axis>: <angle>'."""
    out = {}
    for line in open(os.path.join(GOLDEN, "ref_data", "registration", "transfer.txt")):
        line = line.strip()
        if not line:
            continue
        name, rest = line.split(None, 1)
        axis, ang = rest.replace(" ", "").split(":")
        out[name] = (axis, float(ang))
    return out


def test_transfer_table_parses():
    t = _transfer_table()
    assert t["Bunny"] == ("x", 1.1) and t["ant"] == ("x", 1.56) and t["Horse"] == ("y", 1.1)


ALL_TEN = ["Angel", "Armadillo", "Bunny", "Cat", "Dog", "Girl", "Horse", "ant", "hand", "woodMan"]
SCALE_13 = ["Angel", "Armadillo", "Buddha", "Bunny", "Cat", "Dog", "Dragon", "Girl", "ant", "centuar", "giraffe", "hand", "woodMan"]
# PS_AIS_Simplification/data/registration/ICP.txt:1-13 ("Hand", "WoodMan" there; file names are hand, woodMan)
ICP_TXT = {"Armadillo": True, "Bunny": True, "Cat": True, "Girl": True, "hand": True, "Horse": True, "woodMan": True,
           "Angel": False, "ant": False, "Dog": False}


def _ref_npz(sub):
    return np.load(os.path.join(GOLDEN, "ref_data", sub + "_all.npz"))


def _down(O, P, m):
    return P[O.aivs(P, m)]


def _kss_pipeline(O, S, T):
    """KSSICP_init + KSSICP_Registration (KSS_ICP.hpp:53-131): pNumber = min(n)/2 capped at 2000, AIVS both clouds
    (target first), registration on the samples, pointAlign at full resolution."""
    m = min(min(len(S), len(T)) // 2, 2000)
    Ts = _down(O, T, m); Ss = _down(O, S, m)
    return O.kssicp_register(Ss, Ts, S, 8.0, 1000)


def test_ref_fixtures_hold_all_pairs():
    d = _ref_npz("registration"); e = _ref_npz("registration_scale")
    assert list(d["names"]) == ALL_TEN and list(e["names"]) == SCALE_13
    assert set(_transfer_table()) == set(ALL_TEN) == set(ICP_TXT)
    for sub, name in (("registration", "Bunny"), ("registration", "Horse"), ("registration", "ant"), ("registration_scale", "Bunny")):
        z = _ref_npz(sub)      # the text copies kept for the CLI tests are the same data
        for ext in ("gird", "wlop"):
            t = np.loadtxt(os.path.join(GOLDEN, "ref_data", sub, name + "." + ext), skiprows=1)
            assert np.array_equal(t, z[name + "_" + ext])


@pytest.mark.parametrize("name", ALL_TEN)
def test_kssicp_recovers_reference_rotation(O, name):
    """PIN (first-party path + down-sampler + ICP end to end), all ten pairs of data/registration: the .gird cloud is the
    model rotated about the world origin by transfer.txt's angle (transferPC.hpp:66-98); the whole KSSICP pipeline
    (AIVS x2, pre-shape, 729-candidate search, candidate ICPs) must undo it.  Source and target are different
    resamplings of the surface (grid vs WLOP), so the tolerance is the sampling noise, not float rounding.
    Observed max |dR| over the ten: 0.0007 (ant) ... 0.0095 (Bunny)."""
    d = _ref_npz("registration")
    S, T = d[name + "_gird"], d[name + "_wlop"]
    axis, ang = _transfer_table()[name]
    r = _kss_pipeline(O, S, T)
    assert np.abs(r["R"] - _axis_R(axis, -ang)).max() < 1.2e-2
    assert abs(r["scale"] - 1.0) < 7e-2          # different resamplings: mean radius differs by a few percent (Dog: 1.062)
    assert np.abs(r["t"]).max() < 2e-2
    assert r["final_fitness"] < 2e-3
    qm = O.pcr_qm(r["pointAlign"], T)
    assert qm[0] < 2e-3 and abs(qm[1] - np.sqrt(qm[0])) < 1e-15


@pytest.mark.parametrize("name", ["Bunny", "Horse", "ant"])
def test_kssicp_on_presampled_pairs(O, ref_pairs, name):
    """The round-1 form of the pin (registration on the clouds as they are, no AIVS) kept for the three text fixtures."""
    S, T = ref_pairs[("registration", name)]
    axis, ang = _transfer_table()[name]
    r = O.kssicp_register(S, T, S, 8.0, 1000)
    assert np.abs(r["R"] - _axis_R(axis, -ang)).max() < 8e-3
    assert abs(r["scale"] - 1.0) < 2e-2 and np.abs(r["t"]).max() < 1e-2 and r["final_fitness"] < 1e-3


@pytest.mark.parametrize("name", SCALE_13)
def test_scale_pairs_register_and_preshape_is_idempotent(O, name):
    """All 13 pairs of data/registration_scale (source additionally scaled about its centroid and translated by
    (dis, dis, dis): transferPC.hpp:100-138).  (i) pre-shape is idempotent on size and centroid; (ii) the pipeline
    registers every pair: fitness and PCR_QM MSE of two resamplings of one surface, a proper rotation, a scale between
    0.25 and 4 (observed 0.67 .. 3.48; the rates and angles the authors used are recorded nowhere in the reference, so
    there is no ground-truth transform to compare with: the pin is "registers", not "recovers")."""
    d = _ref_npz("registration_scale")
    S, T = d[name + "_gird"], d[name + "_wlop"]
    ps = O.preshape_stats(S, T)
    sp = O.similarity_apply(S, ps)
    ps2 = O.preshape_stats(sp, T)
    assert abs(ps2.scale - 1.0) < 1e-12 and np.abs(np.array(ps2.shift)).max() < 1e-12 * max(1.0, np.abs(T).max())
    r = _kss_pipeline(O, S, T)
    assert r["final_fitness"] < 1.2e-3
    assert 0.25 < r["scale"] < 4.0
    R = r["R"]
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-5 and abs(np.linalg.det(R) - 1.0) < 1e-5
    qm = O.pcr_qm(r["pointAlign"], T)
    assert qm[0] < 1.2e-3


def test_scale_pair_preshape(O, ref_pairs):
    """registration_scale/Bunny: source also scaled + translated (transferPC.hpp:100-138)."""
    S, T = ref_pairs[("registration_scale", "Bunny")]
    ps = O.preshape_stats(S, T)
    assert 0.5 < ps.scale < 0.9                  # source was enlarged ~1.5x (survey: 1.21 vs 0.81)


def _plain_icp_ok(O, src, tgt, Rexp):
    r = O.icp(src, tgt)                          # the reference's settings: KSS_ICP.hpp:155-162
    return bool(np.abs(r["T"][:3, :3] - Rexp).max() < 0.05), r


def test_plain_icp_reproduces_reference_icp_txt(O):
    """PIN of the PCL-1.8.1 ICP restatement at the only evidence the reference holds for it: ICP.txt lists which of the
    ten models plain ICP registers (7) and which it does not (3).  TransferPC_ReturnPoints returns {wlop, gird}
    (transferPC.hpp:125-133), i.e. source = the WLOP cloud, target = the rotated grid cloud; run that way
    (pcl::IterativeClosestPoint settings of KSS_ICP.hpp:155-162) the restatement reproduces the list 10 / 10.
    Run the other way round (gird -> wlop, the direction KSS-ICP's own rotation recovery uses above) it agrees on 6:
    Cat and hand fail, Angel and ant register -- recorded here and in DESIGN.md section 3, not tuned."""
    d = _ref_npz("registration")
    tr = _transfer_table()
    fwd, rev = {}, {}
    for name in ALL_TEN:
        S, T = d[name + "_gird"], d[name + "_wlop"]
        axis, ang = tr[name]
        rev[name], _ = _plain_icp_ok(O, T, S, _axis_R(axis, ang))      # wlop -> gird: ICP must FIND the rotation
        fwd[name], _ = _plain_icp_ok(O, S, T, _axis_R(axis, -ang))     # gird -> wlop: ICP must UNDO it
    assert rev == ICP_TXT
    assert fwd == {"Angel": True, "Armadillo": True, "Bunny": True, "Cat": False, "Dog": False, "Girl": True, "Horse": True,
                   "ant": True, "hand": False, "woodMan": True}
    assert sum(fwd[n] == ICP_TXT[n] for n in ALL_TEN) == 6


def test_grid_trip_counts(O):
    # double accumulation -> step 8 gives 9 samples incl. 6.3
    assert len(O.grid_angles(8)) == 9
    assert len(O.grid_angles(6)) == 6
    assert len(O.grid_angles(12)) == 12
    assert len(O.grid_angles(16)) == 17
    a = O.grid_angles(8)
    assert a[0] == 0.0 and abs(a[-1] - 6.3) < 1e-12 and a[-1] < 6.3


def test_kdtree_matches_brute_force(O):
    rng = np.random.default_rng(3)
    t = rng.normal(size=(5000, 3)).astype(np.float32)
    t[77] = t[5]; t[4000] = t[5]      # exact duplicates: ties -> lowest index
    q = rng.normal(size=(3000, 3)).astype(np.float32)
    q[0] = t[5]
    for fma in (0, 1):
        ib, db = O.nn_brute(q, t, fma)
        ik, dk = O.KdTree(t).nn(q, fma)
        assert (ib == ik).all() and (db == dk).all()
    assert ib[0] == 5
    ik8, dk8 = O.KdTree(t).nn(q, 0, nthreads=4)
    ib, db = O.nn_brute(q, t, 0)
    assert (ik8 == ib).all()


def test_svd_and_rigid_fit(O):
    rng = np.random.default_rng(5)
    for _ in range(20):
        A = rng.normal(size=(3, 3))
        U, s, V = O.svd3(A)
        assert np.abs(U @ np.diag(s) @ V.T - A).max() < 1e-12
        assert np.allclose(s, np.linalg.svd(A)[1], atol=1e-12)
    # rigid fit recovers a known motion from exact correspondences
    P = rng.normal(size=(500, 3))
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec([0.3, -0.2, 0.5]).as_matrix()
    t = np.array([0.1, -0.3, 0.2])
    Q = P @ R.T + t
    sums = np.zeros(20)
    sums[0] = len(P); sums[1:4] = P.sum(0); sums[4:7] = Q.sum(0)
    sums[7:16] = (P[:, :, None] * Q[:, None, :]).sum(0).reshape(9)
    T = O.rigid_from_sums(sums)
    assert np.abs(T[:3, :3] - R).max() < 1e-6 and np.abs(T[:3, 3] - t).max() < 1e-6
    # reflection guard: planar data must still give det(R) = +1
    P2 = P.copy(); P2[:, 2] = 0
    Q2 = P2 @ R.T + t
    sums[1:4] = P2.sum(0); sums[4:7] = Q2.sum(0); sums[7:16] = (P2[:, :, None] * Q2[:, None, :]).sum(0).reshape(9)
    T2 = O.rigid_from_sums(sums).astype(np.float64)
    assert abs(np.linalg.det(T2[:3, :3]) - 1.0) < 1e-5


def test_local_min_is_clamped_and_non_strict(O):
    v = np.ones((5, 5, 5))
    assert O.local_min(v, 0, 0, 0) and O.local_min(v, 2, 2, 2)     # plateau: every cell passes
    v[4, 4, 4] = 0.5
    assert O.local_min(v, 4, 4, 4)
    assert not O.local_min(v, 2, 2, 2)      # within +-2 of the strictly smaller corner
    assert O.local_min(v, 0, 0, 0)          # window is clamped, not wrapped: the far corner is unseen


def test_golden_vectors_reproduce(O, pkg):
    """Committed vectors (tests/golden/make_golden.py) pin the oracle against drift."""
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    src64, tgt64 = g["g1_src"].astype(np.float64), g["g1_tgt"].astype(np.float64)
    ps = O.preshape_stats(src64, tgt64)
    got = np.array(list(ps.c_src) + list(ps.c_tgt) + list(ps.shift) + [ps.r_src, ps.r_tgt, ps.scale])
    assert np.array_equal(got, g["g1_stats"])
    assert abs(ps.scale - 0.5) < 2e-2       # the pair was built with a 2x source
    sp = O.similarity_apply(src64, ps)
    assert np.array_equal(sp, g["g1_preshaped"])
    r = O.rotation_search(sp, tgt64, 6)
    assert np.array_equal(r["value"], g["g2_value_6"]) and np.array_equal(r["angle_list"], g["g2_list_6"])
    idx, d2 = O.nn_brute(g["g4_src"], g["g4_tgt"])
    assert np.array_equal(idx, g["g4_nn_idx"]) and np.array_equal(d2, g["g4_nn_d2"])
    ri = O.icp(g["g4_src"], g["g4_tgt"], trace_cap=64)
    assert np.array_equal(ri["T"], g["g4_T"]) and ri["iterations"] == int(g["g4_iters"][0])
    assert np.array_equal(ri["trace_sums"], g["g4_trace_sums"])


def test_ply_loader_rules(O, pkg, tmp_path):
    pts = pkg.synth.sphere(1, 50).astype(np.float32)
    p = str(tmp_path / "a.ply")
    pkg.synth.write_ply(p, pts)
    n, got = O.ply_load(p)
    assert n == 50 and np.array_equal(got.astype(np.float32), pts)      # parsed as float, widened
    # no 'element face' line: the reference loops forever; the restatement reports an error
    bad = str(tmp_path / "b.ply")
    open(bad, "w").write("ply\nformat ascii 1.0\nelement vertex 1\nend_header\n0 0 0\n")
    assert O.ply_load(bad)[0] < 0
    assert O.ply_load(str(tmp_path / "c.xyz"))[0] < 0                   # extension test


def test_octree_downsample_voxel_bookkeeping(O, pkg):
    """ko_octree_downsample against an independent numpy replay of PCL's bounding-cube growth and voxel keys: same
    number of occupied voxels, every selected point is a real NN of a voxel centre, insertion order matters."""
    P = (pkg.synth.bumpy(31, 1500) * np.array([1.0, 0.5, 2.0]) + 0.7).astype(np.float64)
    idx, res = O.octree_downsample(P)
    pf = P.astype(np.float32)
    d2 = ((pf[:1000, None, :].astype(np.float64) - pf[None, :, :].astype(np.float64)) ** 2).sum(-1)
    second = np.sort(d2, axis=1)[:, 1]                                   # kn = 2: the nearest OTHER point
    assert abs(res - np.float32(np.sqrt(second).mean())) < 1e-6 * res    # (float vs double distance arithmetic)
    eps = float(np.finfo(np.float32).eps)
    r = float(res)
    mn = pf[0].astype(np.float64) - r                                    # first point: +- r/2, centred in a depth-1 cube
    depth = 1
    mx = mn + 2 * r
    for p in pf.astype(np.float64):
        while True:
            up = p >= mx
            if not (up.any() or (p < mn).any()):
                break
            mn = np.where(up, mn, mn - (1 << depth) * r)
            depth += 1
            mx = mn + (1 << depth) * r - eps
    keys = np.floor((pf.astype(np.float64) - mn) / r).astype(np.int64)
    assert len({tuple(k) for k in keys}) == len(idx)
    cen = ((keys.astype(np.float64) + 0.5) * r + mn).astype(np.float32)
    sel = set(idx.tolist())
    for c in cen[::97]:                                                  # spot check: the NN of a centre was selected
        d = ((pf - c) ** 2).sum(1)
        assert int(np.argmin(d)) in sel
    idx2, _ = O.octree_downsample(P[::-1].copy())
    assert len(idx2) > 0.5 * len(idx)                                    # a different first point shifts the lattice, not the scale
    with pytest.raises(RuntimeError):
        O.octree_downsample(P[:999])


def test_normals_regular_orients_consistently(O, pkg):
    """estimateNormal_RegularNormal restatement: randomly flipped normals come back agreeing with their neighbours,
    the seed (point 0) keeps its sign, a second pass changes nothing."""
    P = pkg.synth.bumpy(33, 4000).astype(np.float64)
    n0 = O.normals_pcl(P, 20)
    rng = np.random.default_rng(1)
    flipped = n0 * rng.choice([-1.0, 1.0], size=(len(P), 1))
    r = O.normals_regular(P, flipped)
    assert np.array_equal(r[0], flipped[0])
    assert np.array_equal(np.abs(r), np.abs(flipped))                     # only signs change
    idx, _ = O.knn_brute(P.astype(np.float32), P.astype(np.float32), 8)
    assert np.mean((r[:, None, :] * r[idx[:, 1:]]).sum(-1) > 0) > 0.99
    assert np.array_equal(O.normals_regular(P, r), r)


def test_golden_tool_vectors_reproduce(O):
    """tests/golden/oracle_vectors_tools.npz (make_golden.py): AIVS, octree, k-NN and normal orientation of the oracle
    are pinned against drift on the G4 target cloud."""
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    f = np.load(os.path.join(GOLDEN, "oracle_vectors_tools.npz"))
    t4 = g["g4_tgt"]; t64 = t4.astype(np.float64)
    assert np.array_equal(O.aivs(t64, 700), f["g5_aivs_700"])
    oi, ores = O.octree_downsample(t64)
    assert np.array_equal(oi, f["g6_octree_idx"]) and ores == f["g6_octree_res"][0]
    ki, kd = O.knn_brute(t4[:64], t4, 13)
    assert np.array_equal(ki, f["g7_knn13_idx"]) and np.array_equal(kd, f["g7_knn13_d2"])
    n0 = O.normals_pcl(t64, 20)
    assert np.array_equal(n0, f["g8_normals"])
    flips = np.where((np.arange(len(t64)) * 2654435761 % 7) < 3, -1.0, 1.0)[:, None]
    assert np.array_equal(O.normals_regular(t64, n0 * flips), f["g8_oriented"])
