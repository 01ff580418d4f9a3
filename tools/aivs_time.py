import sys, time, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle(); S = pkg.synth
ctx = pkg.Context(0)
for n, m in ((100000, 2000), (1000000, 2000), (20000, 2000)):
    P = S.bumpy(3, n)
    ctx.downsample_aivs(P, m)
    t0 = time.perf_counter(); out, idx = ctx.downsample_aivs(P, m); t1 = time.perf_counter()
    r = O.aivs(P, m); t2 = time.perf_counter()
    print("n=%d m=%d: gpu %.2f ms (incl. %d MB upload), cpu oracle %.1f ms, selected %d, identical %s" % (n, m, (t1 - t0) * 1e3, n * 24 // 1000000, (t2 - t1) * 1e3, len(idx), np.array_equal(idx, r)))
# octree down-sampler, PCL-style normals and their orientation (section 8f #3), same protocol
for n in (100000, 1000000):
    P = S.bumpy(3, n).astype(np.float64)
    ctx.downsample_octree(P)
    t0 = time.perf_counter(); idx, res = ctx.downsample_octree(P); t1 = time.perf_counter()
    r, _ = O.octree_downsample(P); t2 = time.perf_counter()
    print("octree n=%d: gpu %.2f ms, cpu oracle %.1f ms, voxels %d, identical %s" % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(idx), np.array_equal(idx, r)))
for n in (100000,):
    P = S.bumpy(3, n).astype(np.float64)
    ctx.normals(P, 20)
    t0 = time.perf_counter(); nr = ctx.normals(P, 20); t1 = time.perf_counter()
    no = ctx.normals_orient(P, nr); t2 = time.perf_counter()
    ro = O.normals_pcl(P, 20); t3 = time.perf_counter()
    rr = O.normals_regular(P, ro); t4 = time.perf_counter()
    print("normals n=%d: gpu %.2f ms + orientation %.2f ms; cpu oracle (brute k-NN, all cores) %.0f ms + %.0f ms; orientation identical given the same input %s" % (
        n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, np.array_equal(ctx.normals_orient(P, ro), rr)))
