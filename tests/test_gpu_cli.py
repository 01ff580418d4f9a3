"""GPU tests of the C++ host mirror (include/*.hpp) and the CLI: the reference's class surface driven the
way Main_KSS_ICP.cpp drives it, compared with the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "kss-icp_amd", "cli")


def _build():
    subprocess.check_call(["make", "-C", CLI, "-s"])


def _vals(out, tag):
    line = [l for l in out.splitlines() if l.startswith(tag + " ")][0]
    return [float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", line[len(tag):])]


def test_downsample_and_transform_f32(ctx, O, pkg):
    P = pkg.synth.bumpy(3, 5000) * 0.7 + np.array([0.3, -0.1, 0.2])
    out, idx = ctx.downsample_fps(P, 300)
    ref = O.fps(P, 300)
    assert np.array_equal(idx, ref) and np.array_equal(out, P[ref])
    assert len(set(idx.tolist())) == 300 and idx[0] == 0
    out1, idx1 = ctx.downsample_fps(P, 1)
    assert idx1[0] == 0
    outn, idxn = ctx.downsample_fps(P[:50], 50)
    assert sorted(idxn.tolist()) == list(range(50))
    with pytest.raises(pkg.KssError):
        ctx.downsample_fps(P[:10], 11)
    T = np.eye(4, dtype=np.float32); T[:3, :3] = pkg.synth.rot_axis_angle([1, 2, 3], 0.4).astype(np.float32); T[:3, 3] = [0.5, 0.25, -1]
    p32 = P.astype(np.float32)
    assert np.array_equal(ctx.transform_apply_f32(T, p32), O.transform_points_f32(T, p32))


def test_mirror_classes_against_oracle(O, ref_pairs):
    _build()
    d = os.path.join(GOLDEN, "ref_data", "registration")
    r = subprocess.run([os.path.join(CLI, "mirror_check"), os.path.join(d, "Horse.gird"), os.path.join(d, "Horse.wlop")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    S, T = ref_pairs[("registration", "Horse")]
    ps = O.preshape_stats(S, T)
    pre = _vals(out, "PRESHAPE")
    assert np.allclose(pre[0:3], ps.c_tgt, rtol=1e-12, atol=1e-13) and np.allclose(pre[3:6], ps.shift, rtol=1e-12, atol=1e-13)
    assert abs(pre[6] - ps.scale) < 1e-12
    rs = O.rotation_search(O.similarity_apply(S, ps), T, 6)
    ang = _vals(out, "ANGLE")
    assert ang[0:3] == list(rs["angle"]) and int(ang[3]) == len(rs["angle_list"]) and int(ang[4]) == 6
    P = O.pose_apply(S, ps, rs["angle"])
    assert np.allclose(_vals(out, "POSE0"), P[0], rtol=0, atol=1e-14)
    ri = O.icp(P.astype(np.float32), T.astype(np.float32))
    assert abs(_vals(out, "JUDGE")[0] - ri["fitness"]) < 1e-8
    icp2 = _vals(out, "ICP2")
    assert abs(icp2[0] - ri["fitness"]) < 1e-8
    Td = ri["T"].astype(np.float64)
    exp0 = Td[:3, :3] @ S[0] + Td[:3, 3]        # member pointSource is the ORIGINAL source here (:224)
    assert np.allclose(icp2[2:5], exp0, atol=1e-4)     # icp2[1] is the "0" of the ALIGN0 tag
    # members no front-end path reaches
    Pf = P.astype(np.float32)
    full = _vals(out, "ICPFULL")
    assert abs(full[0] - ri["fitness"]) < 1e-8 and int(full[1]) == len(S)
    al = O.transform_points_f32(ri["T"], Pf).astype(np.float64)      # PCL's output cloud: final Matrix4f applied in float (:169-180)
    assert np.allclose(full[2:5], al[0], atol=1e-4) and np.allclose(full[5:8], al[-1], atol=1e-4)
    v = _vals(out, "ANGLV")
    assert int(v[0]) == len(S) and np.allclose(v[1:4], al[0], atol=1e-4) and np.allclose(v[4:7], al[7], atol=1e-4)
    assert abs(_vals(out, "ANGL")[0] - ri["fitness"]) < 1e-8
    ra = O.pose_apply(S, ps, [0.7875, 5.5125, 3.15])                # initRegistration_Rotation_Angle (:95-109)
    got = _vals(out, "ROTANG")
    assert np.allclose(got[0:3], ra[0], rtol=0, atol=1e-14) and np.allclose(got[3:6], ra[11], rtol=0, atol=1e-14)
    xm = ps.c_tgt[0]                                                 # initRegistration_Rotation_Axis (:111-140): x_middle_S on ALL axes
    rx = O.axis_rotate(2, 0.6, S - xm) + xm
    got = _vals(out, "ROTAXIS")
    assert np.allclose(got[0:3], rx[0], rtol=0, atol=1e-14) and np.allclose(got[3:6], rx[11], rtol=0, atol=1e-14)
    assert np.allclose(_vals(out, "ROTAXISBAD"), S[0] - xm, rtol=0, atol=1e-15) and "error! illegal rotation" in out
    assert np.allclose(_vals(out, "QM"), O.pcr_qm(P, T), rtol=1e-10)
    reg = _vals(out, "REG")
    # KSSICP_Registration = AIVS down-sample (pNumber = min(n)/2, KSS_ICP.hpp:57-81) + kss_register: same pipeline on the oracle
    m = min(len(S), len(T)) // 2
    ko = O.kssicp_register(S[O.aivs(S, m)], T[O.aivs(T, m)], S, 6.0, 1000)
    assert abs(reg[0] - ko["scale"]) < 1e-12 and abs(reg[1] - ko["final_fitness"]) < 1e-8 and int(reg[2]) == len(S)
    nv = _vals(out, "NORMALS")
    n0 = O.normals_pcl(T, 20)
    nr = O.normals_regular(T, n0)
    assert [int(v) for v in nv[:4]] == [len(T), len(T), 1, len(T)]
    assert np.abs(np.array(nv[4:7]) - n0[0]).max() < 1e-5 and np.abs(np.array(nv[7:10]) - nr[5]).max() < 1e-5
    oc = _vals(out, "OCTREE")
    oi, ores = O.octree_downsample(T)
    assert int(oc[0]) == len(oi) and oc[1] == ores and int(oc[2]) == len(oi) and int(oc[3]) == len(oi) and np.array_equal(oc[4:7], T[oi[0]])
    assert "Down-sampling Point:%d" % len(oi) in out


def test_cli_config_c1(pkg, O, tmp_path):
    """Config C1: two 2k-point uniform-sphere PLYs, 30 degree rotation, through the CLI front-end."""
    _build()
    S = pkg.synth
    src, tgt = S.make_pair(0, 2000, R=S.rot_axis_angle([0, 0, 1], np.deg2rad(30.0)))
    ps, pt, po = str(tmp_path / "s.ply"), str(tmp_path / "t.ply"), str(tmp_path / "o.xyz")
    S.write_ply(ps, src); S.write_ply(pt, tgt)
    r = subprocess.run([os.path.join(CLI, "kss_icp"), ps, pt, po], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "registration finished." in r.stdout and "Registration Measure:MSE:" in r.stdout
    mse = float(re.search(r"Registration Measure:MSE: (\S+)", r.stdout).group(1))
    # a sphere is rotation-degenerate: compare the fit quality with the same pipeline on the oracle
    s64, t64 = src.astype(np.float64), tgt.astype(np.float64)
    ko = O.kssicp_register(s64[O.aivs(s64, 1000)], t64[O.aivs(t64, 1000)], s64, 8.0, 1000)
    assert abs(mse - O.pcr_qm(ko["pointAlign"], t64)[0]) < 1e-6 * mse + 1e-9
    lines = open(po).read().split("\n")
    assert int(lines[0]) == 2000 and len(lines[1].split()) == 3
    # missing file: reference-style diagnostics, non-zero exit
    r = subprocess.run([os.path.join(CLI, "kss_icp"), str(tmp_path / "nope.ply"), pt], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "failed" in r.stdout


def _write_ply_binary(path, pts, with_normals=False):
    """binary_little_endian PLY (an input format the CLI accepts beyond the reference's ASCII-only loader)."""
    pts = np.asarray(pts, dtype="<f4")
    props = "property float x\nproperty float y\nproperty float z\n"
    rec = pts
    if with_normals:
        props = "property uchar flag\n" + props + "property double nx\n"
        dt = np.dtype([("flag", "u1"), ("xyz", "<f4", 3), ("nx", "<f8")])
        rec = np.zeros(len(pts), dt); rec["xyz"] = pts; rec["flag"] = 7; rec["nx"] = 0.5
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n%selement face 0\nend_header\n" % (len(pts), props)).encode())
        f.write(rec.tobytes())


def test_cli_input_formats_and_batch_front_end(pkg, O, tmp_path):
    """The same pair as ASCII PLY, binary PLY (plain and with extra properties) and the reference's count + rows text
    format must register identically; the batch front-end (the reference's commented Main_KSS_List loop) runs the
    reference's own data/registration pairs from their .gird / .wlop files."""
    _build()
    S = pkg.synth
    src, tgt = S.make_pair(3, 1500, R=S.rot_axis_angle([0.3, 0.1, 1.0], np.deg2rad(20.0)), shape="bumpy")
    outs = []
    for kind in ("ascii", "binary", "binary_extra", "text"):
        ps, pt = str(tmp_path / ("s_%s" % kind)), str(tmp_path / ("t_%s" % kind))
        if kind == "ascii":
            ps += ".ply"; pt += ".ply"; S.write_ply(ps, src); S.write_ply(pt, tgt)
        elif kind.startswith("binary"):
            ps += ".ply"; pt += ".ply"
            _write_ply_binary(ps, src, kind == "binary_extra"); _write_ply_binary(pt, tgt, kind == "binary_extra")
        else:
            ps += ".xyz"; pt += ".wlop"
            for p, c in ((ps, src), (pt, tgt)):
                with open(p, "w") as f:
                    f.write("%d\n" % len(c)); f.write("".join("%.9g %.9g %.9g\n" % tuple(r) for r in c))
        r = subprocess.run([os.path.join(CLI, "kss_icp"), ps, pt, str(tmp_path / ("o_%s.xyz" % kind))], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append([l for l in r.stdout.splitlines() if l.startswith(("scale:", "R:", "t:", "Registration Measure"))])
    # float32 coordinates survive the PLY flavours exactly (all parsed as float, then widened); the text format is read
    # as double like the reference's loadPoints, so its 9-digit decimals are not the floats' exact values: close only
    assert len(outs[0]) == 4 and outs[1] == outs[0] and outs[2] == outs[0], outs
    num = lambda ls: np.array([float(x) for l in ls[:3] for x in l.split()[1:]])
    assert np.abs(num(outs[3]) - num(outs[0])).max() < 1e-5, outs
    # batch front-end over the reference's own pairs
    d = os.path.join(GOLDEN, "ref_data", "registration")
    r = subprocess.run([os.path.join(CLI, "kss_icp_list"), d, str(tmp_path), "Bunny", "Horse"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for name in ("Bunny", "Horse"):
        assert re.search(r"^%s:time: " % name, r.stdout, re.M) and re.search(r"^%s:MSE: " % name, r.stdout, re.M)
        mse = float(re.search(r"^%s:MSE: (\S+)" % name, r.stdout, re.M).group(1))
        Sg = np.loadtxt(os.path.join(d, name + ".gird"), skiprows=1); Tw = np.loadtxt(os.path.join(d, name + ".wlop"), skiprows=1)
        m = min(len(Sg), len(Tw)) // 2
        ko = O.kssicp_register(Sg[O.aivs(Sg, m)], Tw[O.aivs(Tw, m)], Sg, 8.0, 1000)      # the same pipeline on the oracle
        assert abs(mse - O.pcr_qm(ko["pointAlign"], Tw)[0]) < 1e-4 * mse                      # (printed with 6 digits)
        a = open(str(tmp_path / (name + "Align.xyz"))).read().split("\n")
        t = open(str(tmp_path / (name + "Target.xyz"))).read().split("\n")
        assert int(a[0]) == len(np.loadtxt(os.path.join(d, name + ".gird"), skiprows=1)) and int(t[0]) == len(np.loadtxt(os.path.join(d, name + ".wlop"), skiprows=1))
    r = subprocess.run([os.path.join(CLI, "kss_icp_list"), d, str(tmp_path), "NoSuchObject"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no source/target file" in r.stdout


def test_aivs_matches_oracle(ctx, O, pkg, ref_pairs):
    """AIVS down-sampler (voxel grid + 8-colour per-voxel FPS + accurate cut): identical selection, in identical
    order, as the oracle's line-by-line restatement of the reference."""
    S = pkg.synth
    cases = [(S.bumpy(5, 2000), 1000), (S.bumpy(6, 5000), 2000), (S.bumpy(7, 20000), 2000), (S.sphere(8, 12345), 700),
             (S.bumpy(9, 100000), 2000), (S.bumpy(10, 64), 30), (S.bumpy(11, 3000) * np.array([1.0, 0.31, 2.7]) + 5.0, 999)]
    for key in (("registration", "Bunny"), ("registration", "Horse"), ("registration_scale", "Bunny")):
        s, t = ref_pairs[key]
        m = min(len(s), len(t)) // 2
        cases += [(s, m), (t, m)]
    for P, m in cases:
        out, idx = ctx.downsample_aivs(P, m)
        ref = O.aivs(P, m)
        assert np.array_equal(idx, ref), (len(P), m, len(idx), len(ref))
        assert np.array_equal(out, np.asarray(P, dtype=np.float64)[ref])
        assert len(set(idx.tolist())) == len(idx)
    # both clouds of a registration in one call (the second on a worker context, concurrently): the same selections
    for (P0, m0), (P1, m1) in zip(cases[0::2], cases[1::2]):
        (o0, i0), (o1, i1) = ctx.downsample_aivs_pair(P0, m0, P1, m1)
        assert np.array_equal(i0, O.aivs(P0, m0)) and np.array_equal(i1, O.aivs(P1, m1)), (len(P0), len(P1))
        assert np.array_equal(o0, np.asarray(P0, dtype=np.float64)[i0]) and np.array_equal(o1, np.asarray(P1, dtype=np.float64)[i1])
    # clouds the reference cannot voxelise are rejected, not mis-sampled
    flat = S.bumpy(12, 500).copy(); flat[:, 2] = 0.25
    with pytest.raises(pkg.KssError) as e2:
        ctx.downsample_aivs_pair(S.bumpy(5, 2000), 1000, flat, 100)      # (the second cloud's status is reported as its own)
    assert e2.value.status == -1
    with pytest.raises(pkg.KssError) as e:
        ctx.downsample_aivs(flat, 100)
    assert e.value.status == -1
    with pytest.raises(RuntimeError):
        O.aivs(flat, 100)


def test_octree_downsampler_matches_oracle(ctx, O, pkg, ref_pairs):
    """Method_Octree.hpp: kNN-derived resolution, PCL's bounding-cube growth in insertion order, occupied voxels in
    depth-first order, nearest cloud point per voxel centre -- identical selection, in identical order, as the oracle's
    restatement (kn = 2 below 80000 points, 7 and 14 above)."""
    S = pkg.synth
    rng = np.random.default_rng(3)
    cases = [S.bumpy(21, 1000), S.bumpy(22, 7000) * np.array([1.0, 0.4, 2.5]) - 3.0, S.sphere(23, 30000),
             S.bumpy(24, 90000), S.bumpy(25, 170000) + 10.0, rng.normal(size=(5000, 3)), ref_pairs[("registration", "Bunny")][1]]
    cases.append(cases[1][rng.permutation(7000)])        # insertion order changes the cube PCL grows
    for P in cases:
        P = np.asarray(P, dtype=np.float64)
        idx, res = ctx.downsample_octree(P)
        ref, rres = O.octree_downsample(P)
        assert res == rres and res > 0, (len(P), res, rres)
        assert np.array_equal(idx, ref), (len(P), len(idx), len(ref))
        assert 0 < len(idx) <= len(P) and idx.min() >= 0 and idx.max() < len(P)
    with pytest.raises(pkg.KssError):
        ctx.downsample_octree(S.bumpy(26, 999))           # the reference reads 1000 points unconditionally
    with pytest.raises(pkg.KssError):
        ctx.downsample_octree(np.ones((1500, 3)))          # coincident points: zero resolution (PCL asserts)
    with pytest.raises(RuntimeError):
        O.octree_downsample(np.ones((1500, 3)))
    dup = np.concatenate([S.bumpy(27, 1200)] * 2)        # every point twice: kn = 2 finds the twin at distance 0 for all
    with pytest.raises(pkg.KssError):
        ctx.downsample_octree(dup)


def test_tools_against_committed_golden_vectors(ctx):
    """The section-8f tools against tests/golden/oracle_vectors_tools.npz (no oracle in the loop): AIVS selection, octree
    selection + resolution, k-NN and normal orientation exact, normals to float round-off."""
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    f = np.load(os.path.join(GOLDEN, "oracle_vectors_tools.npz"))
    t4 = g["g4_tgt"]; t64 = t4.astype(np.float64)
    assert np.array_equal(ctx.downsample_aivs(t64, 700)[1], f["g5_aivs_700"])
    oi, ores = ctx.downsample_octree(t64)
    assert np.array_equal(oi, f["g6_octree_idx"]) and ores == f["g6_octree_res"][0]
    ki, kd = ctx.knn(t4[:64], t4, 13)
    assert np.array_equal(ki, f["g7_knn13_idx"]) and np.array_equal(kd.view(np.uint32), f["g7_knn13_d2"].view(np.uint32))
    n0 = ctx.normals(t64, 20)
    assert np.median(np.abs(n0 - f["g8_normals"]).max(axis=1)) < 1e-6
    flips = np.where((np.arange(len(t64)) * 2654435761 % 7) < 3, -1.0, 1.0)[:, None]
    assert np.array_equal(ctx.normals_orient(t64, f["g8_normals"] * flips), f["g8_oriented"])


def test_normal_orientation_matches_oracle(ctx, O, pkg, ref_pairs):
    """estimateNormal_RegularNormal: propagation over the 8-NN graph from point 0.  Randomly flipped normals come back
    consistently oriented and identical to the oracle's restatement (same graph: the device k-NN is bit-exact)."""
    rng = np.random.default_rng(5)
    for P in (pkg.synth.bumpy(14, 8000), ref_pairs[("registration", "Horse")][0], pkg.synth.sphere(15, 3)[:3] , pkg.synth.bumpy(16, 5)):
        P = np.asarray(P, dtype=np.float64)
        k = min(20, len(P))
        n0 = O.normals_pcl(P, k) if len(P) >= 3 else np.tile([0.0, 0.0, 1.0], (len(P), 1))
        flipped = n0 * rng.choice([-1.0, 1.0], size=(len(P), 1))
        got = ctx.normals_orient(P, flipped)
        ref = O.normals_regular(P, flipped)
        assert np.array_equal(got, ref)
        if len(P) > 100:
            idx, _ = O.knn_brute(P.astype(np.float32), P.astype(np.float32), 8)
            assert np.mean((got[:, None, :] * got[idx[:, 1:]]).sum(-1) > 0) > 0.95      # neighbours agree after the pass (sparse clouds keep a few sharp creases)
            assert np.mean((flipped[:, None, :] * flipped[idx[:, 1:]]).sum(-1) > 0) < 0.6


def test_knn_and_normals_match_oracle(ctx, O, pkg, ref_pairs):
    """Exact k-NN (ascending (d2, index)) bit for bit; PCL-style normals within float round-off of the oracle's
    restatement (device atan2f/cosf/sinf differ from glibc's in the last ulps)."""
    rng = np.random.default_rng(2)
    t = rng.normal(size=(3000, 3)).astype(np.float32)
    t[100] = t[7]; t[2000] = t[7]                      # duplicates: ties keep the lower index first
    q = np.concatenate([t[[7, 100]], rng.normal(size=(500, 3)).astype(np.float32)])
    for k in (1, 3, 13, 20, 32, 35, 64):
        idx, d2 = ctx.knn(q, t, k)
        oi, od = O.knn_brute(q, t, k)
        assert np.array_equal(idx, oi) and np.array_equal(d2.view(np.uint32), od.view(np.uint32))
    big = rng.normal(size=(70000, 3)).astype(np.float32)
    big[40000] = big[12]; big[69999] = big[12]         # ties across target splits keep the lower index first
    for k in (2, 7, 35):                               # few queries x many targets: the split + merge path
        idx, d2 = ctx.knn(big[:300], big, k)
        oi, od = O.knn_brute(big[:300], big, k)
        assert np.array_equal(idx, oi) and np.array_equal(d2.view(np.uint32), od.view(np.uint32))
    idx, d2 = ctx.knn(q[:5], t[:4], 6)                 # fewer targets than k: -1 / +inf tail
    oi, od = O.knn_brute(q[:5], t[:4], 6)
    assert np.array_equal(idx, oi) and np.array_equal(d2, od) and (idx[:, 4:] == -1).all()
    for P in (pkg.synth.bumpy(4, 6000), ref_pairs[("registration", "Horse")][0]):
        n = ctx.normals(P, 20)
        r = O.normals_pcl(P, 20)
        assert np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-12
        dev = np.abs(n - r).max(axis=1)
        assert np.mean(dev < 1e-4) > 0.995               # ill-conditioned (near-isotropic) neighbourhoods may differ more
        assert np.median(dev) < 1e-6
    with pytest.raises(pkg.KssError):
        ctx.knn(q, t, 65)
