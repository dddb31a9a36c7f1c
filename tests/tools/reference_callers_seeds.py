"""FREE-RUNNING comparison of the HIP renderer with the reference under the reference's own loss code, over numpy seeds
(VERDICT r04 next-1b): for each seed and settings file one iteration is run on the reference `Graph` (fp32 PyTorch-ROCm ops on
this GPU) with every draw -- torch's and np.random's -- from np.random.RandomState(seed), then replayed on the HIP `Graph` in
fp32 and bf16x3 mode as the trainer would run it (later calls derive their pixel lists / depth caps from the graph's OWN
earlier outputs; data-dependent ray counts may differ by a threshold flip).  Statistical quantities only: loss terms,
gradient distances, ray counts.  -> gpurun_out/r05_reference_callers_seeds.json (committed under profiles/); the bounds of
tests/test_reference_callers_gpu.py::test_free_running are ~2x the worst value here.

    SPARF_REFERENCE_ROOT=oracle/_ref/reference_tree.zip python tests/tools/reference_callers_seeds.py [--seeds 8]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]

import torch  # noqa: E402

from tests import callers_tape as CT  # noqa: E402
from tests import ref_harness as RH  # noqa: E402

ITER = 110000


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--names", nargs="*", default=["llff_sparf", "replica_sparf", "dtu_barf", "dtu_nerf"])
    ap.add_argument("--few", type=int, default=2, help="seeds for the settings without data-dependent calls (dtu_*)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_reference_callers_seeds.json"))
    a = ap.parse_args()
    dev = "cuda:0"
    doc = dict(runs={}, summary={})
    t0 = time.time()
    for name in a.names:
        nseeds = a.seeds if name.endswith("sparf") else min(a.few, a.seeds)
        for seed in range(nseeds):
            opt = RH.load_settings(name, rays=4096, samples=(64, 128))
            scene = RH.make_scene(name, opt, dev)
            torch.manual_seed(0)
            g_ref, o_ref = RH.build_graph("reference", opt, scene, dev)
            CT.load_seeded(g_ref, 1000)
            state = {k: v.detach().clone() for k, v in g_ref.state_dict().items()}
            tape = RH.DrawTape(seed=seed)
            r_ref = RH.training_iteration(g_ref, o_ref, scene, ITER, tape, "record")
            del g_ref
            torch.cuda.empty_cache()
            fifo0 = {k: list(v) for k, v in tape.fifo.items()}
            for precision in ("fp32", "bf16x3"):
                tape.fifo = {k: list(v) for k, v in fifo0.items()}
                tape.resized = []
                g_hip, o_hip = RH.build_graph("hip", opt, scene, dev, state=state, precision=precision)
                r_hip = RH.training_iteration(g_hip, o_hip, scene, ITER, tape, "replay")
                c = RH.compare(r_ref, r_hip)
                doc["runs"][f"{name}/{precision}/seed{seed}"] = dict(
                    calls_ref=c["calls"]["ref"], calls_hip=c["calls"]["test"], loss={k: v["rel"] for k, v in c["loss"].items()},
                    grad_worst_tensor=c["grad_worst_tensor"], grad_worst_name=c["grad_worst_name"], grad_all=c["grad_all"], grad_pose=c["grad_pose"],
                    resized_draws=len(tape.resized), leftover_draws={str(k): v for k, v in tape.leftover().items()})
                del g_hip
                torch.cuda.empty_cache()
            print(name, seed, f"{time.time() - t0:.0f} s", {p: (doc["runs"][f"{name}/{p}/seed{seed}"]["grad_worst_tensor"], doc["runs"][f"{name}/{p}/seed{seed}"]["calls_hip"][-1])
                                                               for p in ("fp32", "bf16x3")}, flush=True)
    for precision in ("fp32", "bf16x3"):
        for name in a.names:
            rs = [v for k, v in doc["runs"].items() if k.startswith(f"{name}/{precision}/")]
            if not rs:
                continue
            mx = lambda f: max(f(r) for r in rs)
            doc["summary"][f"{name}/{precision}"] = dict(
                seeds=len(rs), loss_rel_max=mx(lambda r: max(r["loss"].values())), grad_worst_tensor_max=mx(lambda r: r["grad_worst_tensor"]),
                grad_all_max=mx(lambda r: r["grad_all"]), grad_pose_max=(mx(lambda r: r["grad_pose"]) if rs[0]["grad_pose"] is not None else None),
                ray_counts_of_the_last_call_ref=sorted({r["calls_ref"][-1][1] for r in rs}),
                max_ray_count_difference=mx(lambda r: max(abs(x[1] - y[1]) for x, y in zip(r["calls_ref"], r["calls_hip"]))))
    doc["what"] = __doc__.split("\n\n")[0]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(doc, open(a.out, "w"), indent=1, default=str)
    print(json.dumps(doc["summary"], indent=1))


if __name__ == "__main__":
    main()
