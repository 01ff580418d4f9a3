"""Regenerates tests/golden/oracle_vectors.npz from the CPU oracle on seeded inputs.

The reference itself cannot be built here (PCL/FLANN/Eigen absent; no stand-in headers allowed),
so these vectors pin the ORACLE against drift; the oracle in turn is pinned by the reference's
own data fixtures in tests/golden/ref_data (known rotations in transfer.txt).
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

O = graft.load_oracle()
pkg = graft.load_package()
S = pkg.synth

out = {}
# G1: pre-shape on a scaled/translated asymmetric pair
src, tgt = S.make_pair(11, 700, R=S.rot_axis_angle([1, 2, 3], 0.7), scale=2.0, t=(0.5, -0.25, 1.0), shape="bumpy", n_src=600)
src64, tgt64 = src.astype(np.float64), tgt.astype(np.float64)
ps = O.preshape_stats(src64, tgt64)
out["g1_src"], out["g1_tgt"] = src, tgt
out["g1_stats"] = np.array(list(ps.c_src) + list(ps.c_tgt) + list(ps.shift) + [ps.r_src, ps.r_tgt, ps.scale])
sp = O.similarity_apply(src64, ps)
out["g1_preshaped"] = sp
# G2: error volume / angle / angleList for accurate 6 and 8
for step in (6, 8):
    r = O.rotation_search(sp, tgt64, step)
    out["g2_value_%d" % step] = r["value"]
    out["g2_angle_%d" % step] = r["angle"]
    out["g2_list_%d" % step] = r["angle_list"]
# G3: pose application
out["g3_pose"] = O.pose_apply(src64, ps, [0.7875, 5.5125, 3.15])
# G4: ICP trace on a small pair
s4, t4 = S.make_pair(12, 1500, R=S.rot_axis_angle([0.3, -0.2, 1.0], np.deg2rad(12.0)), t=(0.03, 0.01, -0.02), shape="bumpy")
r = O.icp(s4, t4, trace_cap=64)
out["g4_src"], out["g4_tgt"] = s4, t4
out["g4_T"] = r["T"]; out["g4_iters"] = np.array([r["iterations"], int(r["converged"]), r["state"]])
out["g4_fitness"] = np.array([r["fitness"], r["last_mse"]])
out["g4_trace_sums"] = r["trace_sums"]; out["g4_trace_Tk"] = r["trace_Tk"]
idx, d2 = O.nn_brute(s4, t4)
out["g4_nn_idx"], out["g4_nn_d2"] = idx, d2
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz"), **out)
print("wrote oracle_vectors.npz", {k: v.shape for k, v in out.items()})

# second file (SURVEY 8c G5 + the section 8f tools): the tools around the path, on the G4 target cloud
f = {}
t64 = t4.astype(np.float64)
f["g5_aivs_700"] = O.aivs(t64, 700)                                   # AIVS selection, output order
oi, ores = O.octree_downsample(t64)
f["g6_octree_idx"], f["g6_octree_res"] = oi, np.array([ores])
ki, kd = O.knn_brute(t4[:64], t4, 13)
f["g7_knn13_idx"], f["g7_knn13_d2"] = ki, kd
n0 = O.normals_pcl(t64, 20)
flips = np.where((np.arange(len(t64)) * 2654435761 % 7) < 3, -1.0, 1.0)[:, None]      # deterministic sign pattern
f["g8_normals"], f["g8_oriented"] = n0, O.normals_regular(t64, n0 * flips)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_vectors_tools.npz"), **f)
print("wrote oracle_vectors_tools.npz", {k: v.shape for k, v in f.items()})
