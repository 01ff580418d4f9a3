"""CPU property test of the skip rule of the fused cell-list pass (kss_grid.hip, phase A): whenever the
rule says "the last winner stands", the brute-force nearest neighbour in the kernel's own f32 arithmetic must be that
winner, with a strictly smaller computed distance than every other target.  The rule is restated here in numpy float32
(same formulas, same margins); this pins the MATHEMATICS -- the kernels themselves are checked bit for bit on the GPU
(tests/test_gpu_*.py, tools/soak_*.py)."""
import numpy as np

f32 = np.float32


def dist2(q, P):
    """(dx*dx + dy*dy) + dz*dz in f32, no fma: the reference arithmetic (FLANN L2_Simple<float>)."""
    d = (P - q).astype(f32)
    return ((d[:, 0] * d[:, 0]).astype(f32) + (d[:, 1] * d[:, 1]).astype(f32)).astype(f32) + (d[:, 2] * d[:, 2]).astype(f32)


def search(q, T):
    """brute force with the kernel's key order (distance bits, then index) -> winner, its d2, second smallest d2"""
    d = dist2(q, T)
    order = np.lexsort((np.arange(len(T)), d))
    return int(order[0]), f32(d[order[0]]), f32(d[order[1]])


def test_skip_rule_never_keeps_a_wrong_winner():
    rng = np.random.default_rng(5)
    kept = searched = 0
    for case in range(60):
        n = int(rng.integers(50, 400))
        scale = float(rng.choice([1e-3, 1.0, 50.0]))
        T = (rng.normal(size=(n, 3)) * scale).astype(f32)
        if case % 3 == 0:      # near-ties: targets in symmetric pairs around the queries' region
            T[1::2] = (-T[::2][:len(T[1::2])]).astype(f32)
        for _ in range(20):
            q = (rng.normal(size=3) * scale * 0.5).astype(f32)
            w, d_w, d_2 = search(q, T)
            # what a search establishes: every other target at computed distance >= d_2 (here the whole cloud is "walked": no
            # pruning radius, no block faces), B = sqrt(d_2 * 0.99999) * 0.999999
            B = f32(f32(np.sqrt(f32(d_2 * f32(0.99999)))) * f32(0.999999))
            acc = f32(0.0)
            step = scale * float(rng.choice([1e-6, 1e-4, 1e-2, 0.1]))
            for it in range(25):
                q_new = (q + rng.normal(size=3).astype(f32) * f32(step)).astype(f32)
                m = (q_new - q).astype(f32)
                moved = f32(np.sqrt(f32(f32(m[0] * m[0] + m[1] * m[1]) + m[2] * m[2])))
                acc = f32(f32(acc + moved) * f32(1.00001))
                q = q_new
                d0 = f32(dist2(q, T[w:w + 1])[0])
                room = f32(B - acc)
                if room > f32(1e-7) and f32(f32(room * room) * f32(0.99999)) > d0:
                    kept += 1
                    d = dist2(q, T)
                    others = np.delete(d, w)
                    assert d[w] == d0 and (others > d0).all(), (case, it, float(d0), float(others.min()))
                else:                      # search again: new winner, new bound, displacement starts over
                    searched += 1
                    w, d_w, d_2 = search(q, T)
                    B = f32(f32(np.sqrt(f32(d_2 * f32(0.99999)))) * f32(0.999999))
                    acc = f32(0.0)
    assert kept > 1000 and searched > 100, (kept, searched)   # both branches exercised
