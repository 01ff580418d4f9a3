"""GPU twins of the oracle's pins on the reference's own data (tests/test_oracle.py): every pair of
data/registration (10) and data/registration_scale (13), through the C-ABI, against the oracle on the same inputs and
against what the reference records about them (transfer.txt rotations, ICP.txt outcomes)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_oracle import ALL_TEN, SCALE_13, ICP_TXT, _axis_R, _transfer_table, _ref_npz

pytestmark = pytest.mark.gpu


def _pipeline(ctx, S, T):
    """KSSICP_init + KSSICP_Registration (KSS_ICP.hpp:53-131) on the device: AIVS both clouds (target first), kss_register."""
    m = min(min(len(S), len(T)) // 2, 2000)
    Ts, _ = ctx.downsample_aivs(T, m)
    Ss, _ = ctx.downsample_aivs(S, m)
    return Ss, Ts, ctx.register(Ss, Ts, S, 8.0, 1000)


@pytest.mark.parametrize("sub,name", [("registration", n) for n in ALL_TEN] + [("registration_scale", n) for n in SCALE_13])
def test_full_pipeline_matches_oracle_on_every_reference_pair(ctx, O, sub, name):
    d = _ref_npz(sub)
    S, T = d[name + "_gird"], d[name + "_wlop"]
    Ss, Ts, got = _pipeline(ctx, S, T)
    m = min(min(len(S), len(T)) // 2, 2000)
    oT, oS = T[O.aivs(T, m)], S[O.aivs(S, m)]
    assert np.array_equal(Ss, oS) and np.array_equal(Ts, oT)               # the down-sampler selects the same points, same order
    ref = O.kssicp_register(oS, oT, S, 8.0, 1000)
    assert got["grid"] == 9 and got["n_angle_list"] == ref["n_angle_list"]
    assert got["used_angle_list"] == ref["used_angle_list"] and got["angle_index"] == ref["angle_index"]
    assert np.array_equal(got["angle"], ref["angle"])
    assert abs(got["scale"] - ref["scale"]) <= 1e-12 * ref["scale"]
    assert np.abs(got["R"] - ref["R"]).max() < 1e-4 and np.abs(got["t"] - ref["t"]).max() < 1e-3      # north star
    assert np.abs(got["R"] - ref["R"]).max() < 2e-5
    assert got["icp_iterations"] == ref["icp_iterations"] and got["icp_converged"] == ref["icp_converged"]
    assert abs(got["E_d_init"] - ref["E_d_init"]) < 1e-8 and abs(got["final_fitness"] - ref["final_fitness"]) < 1e-8
    assert np.abs(got["pointAlign"] - ref["pointAlign"]).max() < 2e-4 * max(1.0, np.abs(T).max())
    if sub == "registration":      # what the reference records: the rotation transfer.txt names is undone
        axis, ang = _transfer_table()[name]
        assert np.abs(got["R"] - _axis_R(axis, -ang)).max() < 1.2e-2


def test_plain_icp_on_the_device_reproduces_reference_icp_txt(ctx, O):
    """ICP.txt (7 successes, 3 failures) with source = .wlop, target = .gird (transferPC.hpp:125-133 order), the
    reference's PCL settings: same outcome list from the device as from the oracle, 10 / 10; same iteration counts and
    convergence states in both directions."""
    d = _ref_npz("registration")
    tr = _transfer_table()
    outcome = {}
    for name in ALL_TEN:
        S, T = d[name + "_gird"].astype(np.float32), d[name + "_wlop"].astype(np.float32)
        axis, ang = tr[name]
        for src, tgt, Rexp, fwd in ((T, S, _axis_R(axis, ang), False), (S, T, _axis_R(axis, -ang), True)):
            got = ctx.icp(src, tgt)
            ref = O.icp(src, tgt)
            assert got["iterations"] == ref["iterations"] and got["state"] == ref["state"] and got["converged"] == ref["converged"], (name, fwd)
            assert np.abs(got["T"] - ref["T"]).max() < 2e-5 and abs(got["fitness"] - ref["fitness"]) < 1e-9, (name, fwd)
            if not fwd:
                outcome[name] = bool(np.abs(got["T"][:3, :3] - Rexp).max() < 0.05)
    assert outcome == ICP_TXT
