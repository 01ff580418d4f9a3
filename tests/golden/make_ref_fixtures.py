#!/usr/bin/env python3
"""Packs the reference's own registration DATA files into two .npz fixtures (data only: point coordinates).

Source: /root/reference/PS_AIS_Simplification/data/registration/*.{gird,wlop} (10 pairs; ground-truth rotations in
transfer.txt, plain-ICP outcome list in ICP.txt -- both already copied as text next to this script's output) and
data/registration_scale/*.{gird,wlop} (13 pairs, source also scaled and translated: transferPC.hpp:100-138, parameters
not recorded by the reference).  Format of the inputs: first line N, then N lines "x y z" (Main_KSS_List.cpp:65-94).
<name>.gird = source (rotated grid resampling), <name>.wlop = target (WLOP resampling), transferPC.hpp:140-160.
Run in the build container (the reference tree does not travel):  python tests/golden/make_ref_fixtures.py
"""
import os

import numpy as np

REF = "/root/reference/PS_AIS_Simplification/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_data")


def pack(sub):
    d = os.path.join(REF, sub)
    names = sorted({f[:-5] for f in os.listdir(d) if f.endswith(".gird")})
    arrays = {}
    for n in names:
        for ext in ("gird", "wlop"):
            a = np.loadtxt(os.path.join(d, n + "." + ext), skiprows=1, dtype=np.float64)
            cnt = int(open(os.path.join(d, n + "." + ext)).readline().split()[0])
            assert a.shape == (cnt, 3), (n, ext, a.shape, cnt)
            arrays["%s_%s" % (n, ext)] = a
    np.savez_compressed(os.path.join(OUT, sub + "_all.npz"), names=np.array(names), **arrays)
    return names


if __name__ == "__main__":
    for sub in ("registration", "registration_scale"):
        print(sub, pack(sub))
