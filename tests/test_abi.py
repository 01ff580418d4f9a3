"""CPU tests: the C-ABI library loads, exports every symbol include/kssicp.h declares, its host-only
entry points agree with the oracle, and compute entry points fail loudly without a GPU."""
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def test_header_symbols_all_exported(pkg):
    hdr = open(os.path.join(ROOT, "include", "kssicp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(kss_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    exported = set(pkg.exported_symbols())
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    assert sorted(pkg.binding.SYMBOLS) == declared      # the binding's list is the header's list


def test_version_and_status_strings(pkg):
    L = pkg.load_library()
    assert L.kss_version() == 100
    assert L.kss_status_string(0) == b"ok"
    assert b"GPU" in L.kss_status_string(-4)


def test_struct_layouts(pkg):
    import ctypes as C
    assert C.sizeof(pkg.IcpResult) == 96       # the RCCL record
    assert C.sizeof(pkg.Pose) == 80


def test_grid_angles_match_oracle(pkg, O):
    for step in (6, 8, 12, 16):
        assert np.array_equal(pkg.grid_angles(step), O.grid_angles(step))


def test_rotation_candidates_match_oracle(pkg):
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    for step in (6, 8):
        best, alist = pkg.rotation_candidates(g["g2_value_%d" % step], step)
        assert np.array_equal(best, g["g2_angle_%d" % step])
        assert np.array_equal(alist, g["g2_list_%d" % step])


def test_rigid_from_sums_matches_oracle(pkg, O):
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    for s, Tk in zip(g["g4_trace_sums"], g["g4_trace_Tk"]):
        T = pkg.rigid_from_sums(s)
        assert np.abs(T - Tk).max() < 2e-7      # independent SVD algorithm, same float result to 1-2 ulp
    rng = np.random.default_rng(0)
    P = rng.normal(size=(100, 3)); P[:, 2] = 0   # planar: rank-deficient covariance
    R = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    Q = P @ R.T
    sums = np.zeros(20); sums[0] = 100; sums[1:4] = P.sum(0); sums[4:7] = Q.sum(0)
    sums[7:16] = (P[:, :, None] * Q[:, None, :]).sum(0).reshape(9)
    assert np.abs(pkg.rigid_from_sums(sums)[:3, :3] - R).max() < 1e-6
    assert np.abs(O.rigid_from_sums(sums)[:3, :3] - R).max() < 1e-6


def test_no_gpu_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.KssError) as e:
        pkg.Context(0)
    assert e.value.status == -4                # KSS_ERR_NODEVICE: there is no CPU fallback


def test_product_does_not_reference_oracle():
    """The shipped path must never import/link the oracle (it is test infrastructure)."""
    for base, _, files in os.walk(os.path.join(ROOT, "kss-icp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "kss_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
    for f in os.listdir(os.path.join(ROOT, "include")):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read().lower()


def test_synth_is_deterministic(pkg, O):
    S = pkg.synth
    assert int(S.splitmix64(42, np.array([7], dtype=np.uint64))[0]) == O.splitmix64(42, 7)
    a = S.sphere(5, 1000)
    assert np.abs(np.linalg.norm(a, axis=1) - 1).max() < 1e-12
    s1, t1 = S.config_c3_pair(3, 500)
    s2, t2 = S.config_c3_pair(3, 500)
    assert np.array_equal(s1, s2) and np.array_equal(t1, t2)
