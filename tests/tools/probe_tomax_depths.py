"""Up to which depths does the config-3 workload call render_up_to_maxdepth...?  Per call: min / median / max of depth_max and the share
of rays below 8 / 16 / 32 / 64 / 256 -- the measurement behind routing render_to_max passes BY VALUE under inverse depth (DESIGN 3.8:
97-100 % of the rays stay below t = 8 after the first steps, so whole passes in fp32 paid for a region almost nobody enters).

    python tests/tools/probe_tomax_depths.py        (on the GPU box)
"""
import sys, torch
sys.path[:0] = ["/root/repo", "/root/repo/compat"]
import bench_workloads as BW
w = BW.Workload(3, "bf16x3", torch.device("cuda:0"), rays=4096)
g = w.graph
orig = g.render_to_max
stats = []
def spy(opt, pose, **kw):
    dm = kw["depth_max"].float()
    stats.append([float(dm.min()), float(dm.median()), float(dm.max())] + [float((dm <= t).float().mean()) for t in (8, 16, 32, 64, 256)])
    return orig(opt, pose, **kw)
g.render_to_max = spy
for i in range(30):
    w.step()
torch.cuda.synchronize()
for s in stats[::5]:
    print(["%.3g" % x for x in s])
