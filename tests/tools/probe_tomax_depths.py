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
