import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g, torch
pkg = g.load_package(); ctx = pkg.Context(0); S = pkg.synth
n = 1000000
src, tgt = S.config_c4(n)
ds = torch.from_numpy(src).cuda(); dt = torch.from_numpy(tgt).cuda()
for _ in range(5): r = ctx.preshape_stats_pair_dev(ds.data_ptr(), n, dt.data_ptr(), n, pkg.binding.F32)
ctx.profile_enable(True); ctx.profile_reset()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): r = ctx.preshape_stats_pair_dev(ds.data_ptr(), n, dt.data_ptr(), n, pkg.binding.F32)
torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / 50
print("C4 pre-shape of both clouds: %.1f us per call; events (both launches) %s; scale %.9f" % (dt_ * 1e6, ctx.profile_get(pkg.K_PRESHAPE), r[1][1] / r[0][1]))
x = torch.rand((64 * 1024 * 1024, 3), dtype=torch.float32, device="cuda")
ctx.preshape_stats_dev(x.data_ptr(), pkg.binding.F32, len(x))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ctx.preshape_stats_dev(x.data_ptr(), pkg.binding.F32, len(x))
torch.cuda.synchronize(); d64 = (time.perf_counter() - t0) / 5
print("64M points: %.3f ms = %.2f TB/s" % (d64 * 1e3, 24.0 * len(x) / d64 / 1e12))
