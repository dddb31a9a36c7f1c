"""Which loss term's gradient differs?  Per-term diagnosis of tests/test_reference_callers_gpu.py: one iteration of the reference's own
training code on the reference `Graph` (recorded) and on the HIP `Graph` (replayed), the gradient of EVERY loss term on its own
(render / corres / depth_cons) compared, then per-tensor errors of the total.  How round 4 found that the depth-consistency loss
differentiates through the pixel coordinates it renders at (profiles/r04e_callers_per_term_before_pixel_grad.log).

    python tests/tools/debug_callers.py <dtu_nerf|dtu_barf|llff_sparf|replica_sparf> <fp32|bf16x3>
"""
import sys, json, torch
sys.path[:0] = ["/root/repo", "/root/repo/compat"]
from tests import ref_harness as RH
name, prec = sys.argv[1], sys.argv[2]
dev = "cuda:0"
opt = RH.load_settings(name, rays=4096, samples=(64, 128))
scene = RH.make_scene(name, opt, dev)
torch.manual_seed(0)
g_ref, o_ref = RH.build_graph("reference", opt, scene, dev)
state = {k: v.detach().clone() for k, v in g_ref.state_dict().items()}
tape = RH.DrawTape()
r_ref = RH.training_iteration(g_ref, o_ref, scene, 110000, tape, "record", per_term_grads=True)
g_hip, o_hip = RH.build_graph("hip", opt, scene, dev, state=state, precision=prec)
r_hip = RH.training_iteration(g_hip, o_hip, scene, 110000, tape, "replay", per_term_grads=True)
c = RH.compare(r_ref, r_hip)
print(json.dumps(c["per_term"], indent=1))
print({k: float('%.2g' % v) for k, v in sorted(c["grad_per_tensor"].items(), key=lambda kv: -kv[1])[:12]})
print("all", c["grad_all"], "worst", c["grad_worst_name"], c["grad_worst_tensor"])
