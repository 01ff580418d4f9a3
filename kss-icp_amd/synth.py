"""Synthetic clouds for the benchmark configs.

Portable counter-based RNG (splitmix64 of seed + counter) so the same clouds can be produced from
C/C++ and numpy.  Uniform unit sphere: z = 2u-1, phi = 2*pi*v.  Target seed = 1000 + pair_id,
source = the same points in a seed-shuffled order, transformed by a similarity, plus N(0, sigma)
jitter from seed 2000 + pair_id.  Uniform spheres are rotation-degenerate, so they are used for
throughput and per-iteration kernel parity; end-to-end (R, t, s) parity uses `bumpy()`.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, counters):
    """Vectorised splitmix64: value for each counter (uint64 array)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (counters.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def u01(seed, n, offset=0):
    c = np.arange(offset, offset + n, dtype=np.uint64)
    return (splitmix64(seed, c) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def sphere(seed, n):
    """n points uniform on the unit sphere, float64."""
    u = u01(seed, n, 0)
    v = u01(seed, n, n)
    z = 2.0 * u - 1.0
    phi = 2.0 * np.pi * v
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)


def bumpy(seed, n):
    """Asymmetric closed surface: radius 1 + 0.3 sin(3 theta) cos(2 phi) + 0.2 z."""
    p = sphere(seed, n)
    theta = np.arccos(np.clip(p[:, 2], -1, 1))
    phi = np.arctan2(p[:, 1], p[:, 0])
    rad = 1.0 + 0.3 * np.sin(3 * theta) * np.cos(2 * phi) + 0.2 * p[:, 2]
    return p * rad[:, None]


def normal(seed, n):
    """n standard normals by Box-Muller."""
    u1 = 1.0 - u01(seed, n, 0)   # (0, 1]
    u2 = u01(seed, n, n)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def permutation(seed, n):
    return np.argsort(splitmix64(seed, np.arange(n, dtype=np.uint64)), kind="stable")


def rot_axis_angle(axis, angle):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def make_pair(pair_id, n, R=None, scale=1.0, t=(0, 0, 0), jitter=1e-3, shape="sphere", n_src=None):
    """Returns (source, target) float32 [n,3].  source = scale*R*target_shuffled + t + jitter."""
    gen = sphere if shape == "sphere" else bumpy
    tgt = gen(1000 + pair_id, n)
    perm = permutation(3000 + pair_id, n)
    src = tgt[perm]
    if n_src is not None:
        src = src[:n_src]
    if R is None:
        R = np.eye(3)
    src = scale * (src @ np.asarray(R, dtype=np.float64).T) + np.asarray(t, dtype=np.float64)
    if jitter:
        m = len(src)
        noise = normal(2000 + pair_id, 3 * m).reshape(m, 3) * jitter
        src = src + noise
    return src.astype(np.float32), tgt.astype(np.float32)


def config_c2(n=100000):
    """C2: single 100k x 100k pair, R_z(10 deg) (inside ICP's basin)."""
    return make_pair(0, n, R=rot_axis_angle([0, 0, 1], np.deg2rad(10.0)))


def config_c3_pair(pair_id, n=10000):
    """C3/C5 pair: random axis, angle U(0, 15 deg)."""
    r = u01(4000 + pair_id, 4)
    axis = sphere(5000 + pair_id, 1)[0]
    return make_pair(pair_id, n, R=rot_axis_angle(axis, np.deg2rad(15.0 * r[0])))


def config_c4(n=1000000):
    """C4: 1M x 1M, 2x scale + 60 deg about (1,1,1)/sqrt(3), t = (0.5, -0.25, 1.0)."""
    return make_pair(0, n, R=rot_axis_angle([1, 1, 1], np.deg2rad(60.0)), scale=2.0, t=(0.5, -0.25, 1.0))


def write_ply(path, pts):
    """ASCII PLY the reference loader accepts: needs 'element vertex', 'element face', 'end_header'."""
    pts = np.asarray(pts)
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                "element face 0\nproperty list uchar int vertex_indices\nend_header\n" % len(pts))
        for p in pts:
            f.write("%.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
