"""Headline benchmark: training rays/sec of the hierarchical NeRF renderer hot path.

    python bench.py --gpus N --steps K --warmup W [--config 1|2|3|4] [--precision bf16x3|bf16|fp32]
    (N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
     or started plainly, in which case it re-launches itself under torch.distributed.run)

Precision: the measured mode is the product's default, bf16x3 (bf16 MFMA with head + tail operands; outputs within 1e-4 of the
reference -- measured 2.8e-5 at the metric-depth configs 1 / 2 / 4, profiles/r*_parity_scale.json and `parity_live`; inverse-depth
passes (config 3) send the last samples of every ray through the fp32 kernels to keep that bound: 1.3e-5; gradients 7e-3 relative
L2 under a random linear loss, 2e-3 under the photometric loss); the plain bf16 throughput mode and the fp32-MFMA mode are
measured briefly and reported in `other_modes` on the same line.

One step = one optimiser-ready training iteration on synthetic data of the chosen BASELINE.json
config (bench_workloads.py): config 1 (default, the config the metric is quoted on) = 4096 rays x
(64 coarse + 128 fine samples), two 8x256 MLPs, forward + backward (dgrad + wgrad) through both
passes, photometric MSE loss on rgb and rgb_fine, gradient all-reduce when N > 1, gradient-norm
clipping + Adam on both networks.  Inputs are resident in HBM before the timed region.  Each rank
renders its own ray batch (weak scaling).

The JSON line also carries
  roofline     : dominant kernel (by time per step) vs its MI355X bound -- algorithmic flops (or
                 bytes) per launch / average launch duration measured here with stream events
                 around single-kernel launches (C ABI sparf_launch_kernel) / the guide's peak
                 (2.5 PF dense bf16 MFMA for every bf16-operand mode, 157.3 TF fp32, 8 TB/s HBM).
                 `mfma_issue_util` next to it counts the emulation MFMAs bf16x3 issues per product;
  cpu_baseline : the CPU oracle (oracle/nerf_oracle.py, the pinned restatement of the reference's
                 PyTorch path) timed on this box's host cores on a bounded sample (config 0: 255
                 rays, 64 coarse + 128 fine and coarse-only, forward + backward);
  psnr_vs_ref  : the HIP path and the oracle (as fp32 PyTorch-ROCm ops on the same GPU) trained side by
                 side at the benchmark's batch (4096 rays x (64+128)) from identical initialisation with
                 identical rays and random draws, for a few hundred steps under a wall-time cap: held-out
                 PSNR of both and the gap; the committed 2000-step curves of every precision mode
                 (tests/tools/psnr_curve.py -> profiles/r03_psnr_curve_c{1,2}.json) are quoted next to it;
  parity_live  : one 4096-ray config-1 render + backward of THIS run against the float64 referee (on the
                 GPU, ~2 s); `parity` = the committed bounds over all four configs (profiles/r*_parity_scale.json);
  sustained    : after the K contract steps the same step loop runs on for >= --min-seconds with one
                 event per step: rays/s over the whole loop and over its last second, ms/step p50 / p95.
"""
import argparse
import contextlib
import ctypes
import glob
import json
import math
import os
import re
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]

MACS_PER_ROW = 527872
FLOP_FWD_ROW = 2 * MACS_PER_ROW
FLOP_TRAIN_RAY = 3 * FLOP_FWD_ROW * 256    # fwd + dgrad + wgrad, 64 + 192 rows per ray = 810.8 MFLOP
# dense MFMA peaks, MI355X_MICROARCH.md: every bf16-operand mode is priced against the bf16 peak with
# ALGORITHMIC flops (emulation / padding / recompute flops are not counted)
PEAK = {"bf16": 2500.0, "bf16x3": 2500.0, "fp32": 157.3}
PEAK.update({"bf16+q8": PEAK["bf16"], "bf16x3+q8": PEAK["bf16x3"]})            # '+q8': 8-bit save / gradient areas (C ABI 5), same arithmetic
# MFMAs issued per algorithmic product (bf16x3: head*head + head*tail + tail*head in the forward,
# weights head + tail against a bf16 gradient in dgrad, head planes only in wgrad)
MFMA_PER_PRODUCT = {"bf16": {"mlp_fwd": 1, "mlp_dgrad": 1, "wgrad": 1}, "fp32": {"mlp_fwd": 1, "mlp_dgrad": 1, "wgrad": 1},
                    "bf16x3": {"mlp_fwd": 3, "mlp_dgrad": 2, "wgrad": 1}}
MFMA_PER_PRODUCT.update({"bf16+q8": MFMA_PER_PRODUCT["bf16"], "bf16x3+q8": MFMA_PER_PRODUCT["bf16x3"]})
HBM_PEAK_GBS = 8000.0
PEAK_CLOCK_GHZ = 2.4                        # engine clock behind the 2.5 PF dense bf16 figure (MI355X_MICROARCH.md)


def default_precision():
    """the product's default mode (sparf_amd.frequency_nerf.DEFAULT_PRECISION, or $SPARF_PRECISION): bench.py measures what an
    unmodified trainer gets"""
    from sparf_amd.frequency_nerf import DEFAULT_PRECISION
    return os.environ.get("SPARF_PRECISION") or DEFAULT_PRECISION


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def reference_graph_cpu():
    """The REFERENCE's own `source.models.renderer.Graph` class (CPU PyTorch) -- only when the user points at a reference checkout
    or archive with $SPARF_REFERENCE_ROOT (oracle/stage_reference.py); None otherwise: the default `cpu_baseline` is the pinned oracle
    port (a default bench run executes no reference code).  Checker-side only, never the timed GPU region."""
    if not os.environ.get("SPARF_REFERENCE_ROOT"):
        return None
    try:
        from tests import ref_harness as RH
        if RH.install_reference() is None:
            return None
        from source.models.renderer import Graph as RefGraph
        return RefGraph
    except Exception:
        return None


def cpu_baseline(max_seconds=30.0):
    """The reference renderer forward+backward on the host: config 0 (3 views x 85 rays) with 64 coarse + 128 fine
    samples and coarse-only ("64 coarse" as BASELINE.json words it).  `kind` = "port" (default): the oracle (oracle/nerf_oracle.py, the
    restatement pinned to the reference's golden vectors; 0.93-1.0 of the reference module's rate on the same host and rays,
    profiles/r03_cpu_ref_vs_port.json); `kind` = "reference" with $SPARF_REFERENCE_ROOT set: the reference's own `Graph.render`
    (source/models/renderer.py:250-345) under torch autograd.  torch's intra-op pool does not
    scale to every core of a 2-socket host for GEMMs this small, so a few thread counts are tried
    (one timed iteration each) and the fastest is used for the reported median; `cores` is that
    thread count."""
    from oracle import nerf_oracle as O
    from bench_workloads import cameras
    from sparf_amd.config import baseline_opt
    ncpu = os.cpu_count() or 1
    B, R = 3, 85
    pose, intr = cameras(2, "cpu")
    g = torch.Generator().manual_seed(1)
    idx = torch.randperm(300 * 400, generator=g)[:R]
    center, ray = O.rays_at_index(pose, intr, 300, 400, idx)
    RefGraph = reference_graph_cpu()

    def make_reference(fine):
        opt = baseline_opt(0)
        opt.nerf.fine_sampling = fine
        torch.manual_seed(0)
        ref = RefGraph(opt, torch.device("cpu"))
        rng = torch.tensor([1.2, 5.2])

        def one_iter():
            t0 = time.perf_counter()
            out = ref.render(opt, pose, H=300, W=400, intr=intr, ray_idx=idx, depth_range=rng, iter=1000, mode="train")
            (out.rgb.mean() + (out.rgb_fine.mean() if fine else 0.0)).backward()
            dt = time.perf_counter() - t0
            ref.zero_grad(set_to_none=True)
            return dt
        return one_iter

    def make_port(fine):
        opt = baseline_opt(0)
        opt.nerf.fine_sampling = fine
        Nc, Nf = opt.nerf.sample_intvs, opt.nerf.sample_intvs_fine
        pc, pf = O.init_params(opt, 0), O.init_params(opt, 1, fine=True)
        for p in (pc, pf):
            for k, v in p.items():
                if k != "progress":
                    v.requires_grad_(True)

        def one_iter():
            jitter, grid = torch.rand(B, R, Nc, 1, generator=g), torch.rand(Nf + 1, generator=g)
            nc, nf = torch.randn(B, R, Nc, generator=g), torch.randn(B, R, Nc + Nf, generator=g)
            t0 = time.perf_counter()
            out = O.render(opt, pc, pf, center, ray, [1.2, 5.2], mode="train", it=1000, jitter=jitter, grid=grid, noise_c=nc, noise_f=nf)
            (out["rgb"].mean() + (out["rgb_fine"].mean() if fine else 0.0)).backward()
            dt = time.perf_counter() - t0
            for p in (pc, pf):
                for v in p.values():
                    v.grad = None
            return dt
        return one_iter

    make = make_reference if RefGraph is not None else make_port
    t_start = time.perf_counter()
    one_iter = make(True)
    best_n, best_t = None, float("inf")
    for n in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(n)
        one_iter()                                   # warm-up at this thread count
        dt = one_iter()
        if dt < best_t:
            best_n, best_t = n, dt
        if time.perf_counter() - t_start > max_seconds * 0.4:
            break
    torch.set_num_threads(best_n)

    def median(fn, budget):
        t1, times = time.perf_counter(), []
        while len(times) < 10 and (time.perf_counter() - t1 < budget or len(times) < 3):
            times.append(fn())
        times.sort()
        return times[len(times) // 2], len(times)

    med, n_it = median(one_iter, max_seconds * 0.45)
    coarse_iter = make(False)
    coarse_iter()
    med_c, n_c = median(coarse_iter, max_seconds * 0.15)
    kind = "reference" if RefGraph is not None else "port"
    what = ("reference Graph.render + backward (source/models/renderer.py:250-345 under torch autograd, staged reference tree)"
            if kind == "reference" else "oracle fwd+bwd (oracle/nerf_oracle.py)")
    out = dict(value=B * R / med, unit="rays/s", cores=best_n, kind=kind, cpu=cpu_model(), host_threads=ncpu, torch=torch.__version__,
               coarse_only=dict(value=B * R / med_c, unit="rays/s", sample=f"{B}x{R}=255 rays x 64 coarse samples, one network, median of {n_c} ({med_c * 1e3:.0f} ms each)"),
               sample=f"{what}, {B}x{R}=255 rays x (64+128) samples, median of {n_it} iterations "
                      f"({med * 1e3:.0f} ms each) with {best_n} of {ncpu} host threads ({cpu_model()}), torch {torch.__version__} CPU")
    if kind == "reference":           # the port next to it, same host and thread count (a few iterations)
        port_iter = make_port(True)
        port_iter()
        med_p, n_p = median(port_iter, max_seconds * 0.15)
        out["port"] = dict(value=B * R / med_p, unit="rays/s", port_over_reference=(B * R / med_p) / (B * R / med),
                           sample=f"oracle/nerf_oracle.py on the same rays, median of {n_p} ({med_p * 1e3:.0f} ms each), {best_n} threads")
    return out


def round_of(path):
    """r03_..., r03b_... -> (3, "b"): newest round first, not lexicographic (r100 > r99)"""
    m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(path))
    return (int(m.group(1)), m.group(2)) if m else (-1, "")


def pmc_traffic(kernel, prec_name, rows):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_<prec>.json, written by tools/pmc_profile.sh + tools/pmc_summary.py:
    FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections applied there).  The
    counters cannot be read from inside this process, so the newest committed profile of the
    same kernel, precision and row count is reported; None if there is none."""
    q8 = prec_name.endswith("+q8")
    tags = _kernel_tags(kernel, prec_name)
    q8_ok = lambda name: kernel != "mlp_dgrad" or "MlpBwdArgs" not in name or name.split("(")[0].rstrip().endswith(", true>") == q8 or not name.split("(")[0].rstrip().endswith(("true>", "false>"))
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{prec_name}.json")), key=round_of, reverse=True):
        try:
            prof = json.load(open(f))
        except (OSError, ValueError):
            continue
        if prof.get("_meta", {}).get("rows", 786432) != rows:
            continue
        for name, e in prof.items():
            if any(tag in name for tag in tags) and q8_ok(name) and "hbm_read_bytes" in e and "hbm_write_bytes" in e:
                util = e.get("mfma_util")
                if util is not None and e.get("mfma_busy_cycles") and e.get("duration_ns_under_pmc"):
                    # matrix-pipe busy cycles over the cycles 1024 SIMDs would tick at the clock the peak figures assume: the PMC
                    # ratio above is relative to the clock the chip actually held under this load (power cap)
                    util = dict(at_clock_held=util, at_peak_clock=e["mfma_busy_cycles"] / (1024 * PEAK_CLOCK_GHZ * e["duration_ns_under_pmc"]),
                                clock_held_ghz=util and e["mfma_busy_cycles"] / util / 1024 / e["duration_ns_under_pmc"])
                return e["hbm_read_bytes"] + e["hbm_write_bytes"], os.path.relpath(f, ROOT), util
    return None, None, None


def _kernel_tags(kernel, prec_name):
    """kernel names as rocprofv3 prints them: this round's template arguments (save mode 0 / 1 / 2, the 8-bit flag last) and the earlier ones"""
    base, q8 = prec_name.split("+")[0], prec_name.endswith("+q8")
    pid = {"bf16": 0, "fp32": 1, "bf16x3": 2}[base]
    return {"mlp_fwd": [f"mlp_fwd_kernel<{pid}, {2 if q8 else 1}>", f"mlp_fwd_kernel<{pid}, true>"],
            "mlp_dgrad": [f"mlp_bwd_kernel<{pid}, false"],
            "wgrad": [f"wgrad_kernel<{0 if q8 else pid}, {'true' if q8 else 'false'}>", f"wgrad_kernel<{pid}>"]}[kernel]


def live_pmc_traffic(kernel, prec_name, rows, max_seconds=150.0):
    """HBM bytes per launch of `kernel` measured NOW, on this box: two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE: one counter per
    pass, kernel trace only -- MI355X_MICROARCH.md "HBM / rocprofv3") over tools/kernel_bench.py restricted to that kernel, each in its
    own subprocess (a process cannot read its own PMC counters); the last five launches of the kernel are averaged; FETCH_SIZE x 2 on
    gfx950 and KiB units as tools/pmc_summary.py applies them.  -> (bytes per launch, description) or (None, why not)."""
    import csv
    import shutil
    import tempfile
    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself running under rocprofv3: no nested profiler (the committed profile is quoted)"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or rows != 786432:
        return None, "rocprofv3 not found" if exe is None else "tools/kernel_bench.py measures the 786 432-row fine pass only"
    only = {"mlp_fwd": "fwd save$", "mlp_dgrad": "dgrad$", "wgrad": "wgrad$"}[kernel]
    tags = _kernel_tags(kernel, prec_name)
    t0, got = time.perf_counter(), {}
    tmp = tempfile.mkdtemp(prefix="sparf_pmc_", dir="/tmp")
    try:
        # third pass (optional: a failure there keeps the traffic figures): matrix-pipe busy cycles over the cycles the chip was active
        for group in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]):
            counter, must = group[0], len(group) == 1
            left = max_seconds - (time.perf_counter() - t0)
            if left < 20:
                if must:
                    return None, "time budget of the live PMC passes exhausted"
                break
            env = dict(os.environ, KB_ONLY=only, TMPDIR="/tmp")
            cmd = [exe, "--pmc"] + group + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", counter, "--",
                                            sys.executable, os.path.join(ROOT, "tools", "kernel_bench.py"), prec_name]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left)
            files = glob.glob(os.path.join(tmp, "**", f"{counter}_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                if must:
                    return None, f"rocprofv3 --pmc {counter} failed (exit {r.returncode})"
                break
            rows_ = [row for row in csv.DictReader(open(files[0])) if any(t in row["Kernel_Name"] for t in tags)]
            for name in group:
                vals = [float(row["Counter_Value"]) for row in rows_ if row["Counter_Name"] == name]
                if vals:
                    got[name] = sum(vals[-5:]) / len(vals[-5:])
            if must and counter not in got:
                return None, f"no {counter} rows for {kernel} in the PMC pass"
            if not must and rows_:
                dur = [int(row["End_Timestamp"]) - int(row["Start_Timestamp"]) for row in rows_ if row["Counter_Name"] == "GRBM_GUI_ACTIVE"]
                if dur:
                    got["duration_ns"] = sum(dur[-5:]) / len(dur[-5:])
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as exc:
        return None, f"{type(exc).__name__}: {str(exc)[:160]}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = got["FETCH_SIZE"] * 1024 * 2 + got["WRITE_SIZE"] * 1024
    live_pmc_traffic.extra = None
    if got.get("SQ_VALU_MFMA_BUSY_CYCLES") and got.get("GRBM_GUI_ACTIVE") and got.get("duration_ns"):
        # matrix-pipe busy share as tools/pmc_summary.py computes it (1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs): at the clock
        # the chip HELD under the profiler, and as a share of the issue slots at the 2.4 GHz the peak figures assume
        busy, act, dur = got["SQ_VALU_MFMA_BUSY_CYCLES"], got["GRBM_GUI_ACTIVE"], got["duration_ns"]
        live_pmc_traffic.extra = dict(pmc_mfma_busy=busy / (1024 * act / 8), mfma_busy_at_peak_clock=busy / (1024 * PEAK_CLOCK_GHZ * dur),
                                      clock_held_ghz_under_pmc=act / 8 / dur, launch_ms_under_pmc=dur * 1e-6)
    return total, (f"measured by this run: rocprofv3 --pmc passes (FETCH_SIZE x 2 as gfx950 counts it; WRITE_SIZE; KiB units) over tools/kernel_bench.py in "
                   f"subprocesses on this box, mean of the last launches: read {got['FETCH_SIZE'] * 2048 / 1e9:.3f} GB + written "
                   f"{got['WRITE_SIZE'] * 1024 / 1e9:.3f} GB, {time.perf_counter() - t0:.0f} s")


def measured_parity(prec_name):
    """Error bounds of a precision mode at the benchmark shapes, from the newest committed
    profiles/r*_parity_scale.json (tests/tools/scale_parity.py: HIP path vs the oracle's float64 referee at
    BASELINE configs 1-4; tests/test_00_scale_gpu.py asserts them).  `metric_depth` = configs 1, 2, 4;
    `inverse_depth` = config 3, whose far samples (t up to 1e8) make the per-sample values and the
    gradients heavy-tailed for the fp32 reference itself (`reference_fp32` = its own distance to the
    referee on the same inputs)."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_parity_scale.json")), key=round_of, reverse=True):
        try:
            doc = json.load(open(f))
            summ = doc["summary"]
        except (OSError, ValueError, KeyError):
            continue
        if prec_name in summ and "metric_depth" in summ[prec_name]:
            out = dict(summ[prec_name])
            from sparf_amd.build import source_hash
            stamp = doc.get("_meta", {}).get("kernel_source_hash")
            out["stale"] = None if stamp is None else (stamp != source_hash())      # measured on other kernel sources than the ones built here (see parity_live)
            out["reference_fp32"] = summ.get("reference_fp32")
            out["referee"] = "oracle float64 on identical rays / depths / noise, 4096-ray batches; outputs max|a-b|/max|b|, gradients relative L2"
            out["source"] = os.path.relpath(f, ROOT)
            return out
    return None


def box_summary(tel, box, value_per_gpu, contract_step_ms, sustained):
    """The few numbers that tell a slow box from a slow build, for `config.box` / `roofline.box` (the keys the driver keeps): clocks and
    power over the contract region, the calibration kernels before / after, and the headline divided by the calibration figures.  The
    ratios are indicators, not constants: round 6's survey (profiles/r06_box_survey.jsonl) found the step following the pure-MFMA figure
    with an elasticity of ~0.37 (a throttled lease reads 9 % less on it and 3 % less on the step)."""
    reg = (tel.get("regions") or {}).get("contract") or {}
    sus = (tel.get("regions") or {}).get("sustained") or {}
    pick = lambda r, k, f: (r.get(k) or {}).get(f) if r else None
    cal = [c for c in (box.get("calib_before"), box.get("calib_after")) if isinstance(c, dict)]
    mf = [c["mfma"]["tflops_second_half"] for c in cal]
    out = dict(sensor_backend=tel.get("backend"), power_cap_w=tel.get("power_cap_w"),
               clock_ghz_mean=pick(reg, "clock_ghz", "mean"), clock_ghz_min=pick(reg, "clock_ghz", "min"), power_w_mean=pick(reg, "power_w", "mean"),
               temp_c_max=pick(reg, "temp_c", "max"), contract_samples=reg.get("n"),
               sustained_clock_ghz_mean=pick(sus, "clock_ghz", "mean"), sustained_power_w_mean=pick(sus, "power_w", "mean"),
               calib_mfma_tflops=(sum(mf) / len(mf)) if mf else None, calib_mfma_tflops_before_after=mf or None,
               calib_hbm_read_tbs=[round(c["hbm"]["read_lds_dma_tbs"], 3) for c in cal] or None,
               calib_hbm_copy_tbs=[round(c["hbm"]["copy_tbs"], 3) for c in cal] or None,
               contract_step_ms=contract_step_ms, sustained_value=sustained["value"] if sustained else None)
    if mf:
        out["value_per_calib_mfma_tflop"] = value_per_gpu / (sum(mf) / len(mf))          # rays/s per GPU per sustained issued bf16 TFLOP/s of this box
        if sustained:
            out["sustained_value_per_calib_mfma_tflop"] = sustained["value"] / (sum(mf) / len(mf))
    mx = [c["mix"]["cycles_per_s_second_half"] for c in cal if "mix" in c]
    if mx:          # the two calibration kernels alternating at the step's cadence (bench_telemetry.Calibration.mix): closer to what the step asks of the chip
        out["calib_mix_cycles_per_s"] = sum(mx) / len(mx)
        out["value_per_calib_mix_cycle"] = value_per_gpu / (sum(mx) / len(mx))
        if sustained:
            out["sustained_value_per_calib_mix_cycle"] = sustained["value"] / (sum(mx) / len(mx))
    return out


def kernel_roofline(graph, opt, prec_name, device, rays=4096, reps=5, replayed=None, live_pmc=False):
    """Time the three heavy kernels of the FINE pass (786 432 rows: 3/4 of the step's MLP work) and of the COARSE pass (262 144 rows) one
    launch at a time and return the roofline entry of the dominant one.  `achieved` / `frac` are the fine launch's (as in every earlier
    round); `avg_launch_ms` = mean of the coarse and the fine launch is what `rocprofv3 --kernel-trace --stats` of this command averages
    over (one coarse and one fine launch per step), `frac_coarse_plus_fine` the same fraction over both launches."""
    from sparf_amd import lib as L, ops
    lib = L.load()
    prec = L.PREC_IDS[prec_name]
    s = L.stream_ptr(device)

    def time_pass(net, N, seed):
        g = torch.Generator().manual_seed(seed)
        c = (torch.rand(rays, 3, generator=g) - 0.5 + torch.tensor([0.0, 0.0, -3.0])).to(device)
        d = (torch.rand(rays, 3, generator=g) * 0.6 - 0.3 + torch.tensor([0.0, 0.0, 1.0])).to(device)
        t = (torch.sort(torch.rand(rays, N, generator=g), dim=1).values * 4.0 + 1.2).to(device)
        packed, c2f = net.packed(prec), net.band_weights()
        fa, out, save, keep1 = ops.build_pass_fwd(prec, c, d, t, None, 0.0, False, packed, c2f, True)
        L.check(lib.sparf_pass_forward(ctypes.byref(fa), s), "fwd")
        grads = (torch.rand(rays, 3, device=device), None, None, None)
        ba, gp, _, _, keep2 = ops.build_pass_bwd(prec, c, d, t, None, 0.0, False, packed, c2f, save, out, grads, False)
        L.check(lib.sparf_pass_backward(ctypes.byref(ba), s), "bwd")
        res = {}
        for which, name in ((0, "mlp_fwd"), (1, "mlp_dgrad"), (2, "wgrad")):
            L.check(lib.sparf_launch_kernel(which, ctypes.byref(fa), ctypes.byref(ba), s), name)     # warm
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                L.check(lib.sparf_launch_kernel(which, ctypes.byref(fa), ctypes.byref(ba), s), name)
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / reps * 1e-3
        return res

    N = opt.nerf.sample_intvs + opt.nerf.sample_intvs_fine
    rows = rays * N
    res = time_pass(graph.nerf_fine, N, 3)
    res_c = time_pass(graph.nerf, opt.nerf.sample_intvs, 4)
    rows_c = rays * opt.nerf.sample_intvs
    ab = 4 if prec_name == "fp32" else 1 if prec_name.endswith("+q8") else 2          # bytes per saved element (bf16x3 saves the bf16 head plane)
    flops = rows * FLOP_FWD_ROW                       # each of fwd / dgrad / wgrad: 2*MACs per row
    wgrad_bytes = rows * (2272 + 2240 + 64) * ab         # X + dY read once (+ the 64 x0 columns, used by layers 0 and 4)
    mfma_peak = PEAK[prec_name]
    entries = {
        "mlp_fwd": dict(bound="mfma", achieved=flops / res["mlp_fwd"] / 1e12, peak=mfma_peak, unit="TFLOP/s"),
        "mlp_dgrad": dict(bound="mfma", achieved=flops / res["mlp_dgrad"] / 1e12, peak=mfma_peak, unit="TFLOP/s"),
        "wgrad": dict(bound="hbm", achieved=wgrad_bytes / res["wgrad"] / 1e9, peak=HBM_PEAK_GBS, unit="GB/s"),
    }
    for k, e in entries.items():
        e.update(frac=e["achieved"] / e["peak"], traffic=None, kernel=k, launch_ms=res[k] * 1e3, rows=rows,
                 coarse_launch_ms=res_c[k] * 1e3, rows_coarse=rows_c, avg_launch_ms=(res[k] + res_c[k]) * 0.5e3)
        work = lambda r: r * (2272 + 2240 + 64) * ab if e["bound"] == "hbm" else r * FLOP_FWD_ROW
        scale = 1e9 if e["bound"] == "hbm" else 1e12
        e["frac_coarse_plus_fine"] = (work(rows) + work(rows_c)) / (res[k] + res_c[k]) / scale / e["peak"]
        if e["bound"] == "mfma":      # share of MFMA issue slots the kernel fills, emulation products included (compare with PMC mfma_util)
            e["mfma_issue_util"] = e["frac"] * MFMA_PER_PRODUCT[prec_name][k]
    dom = max(res, key=res.get)
    roof = dict(entries[dom])
    roof["traffic"], src, pmc_util = pmc_traffic(dom, prec_name, rows)
    if live_pmc:          # the same counters, collected now on this box (subprocesses): replaces the committed figure when it works
        torch.cuda.synchronize()
        live, how = live_pmc_traffic(dom, prec_name, rows)
        if live is not None:
            if replayed is not None and roof["traffic"] is not None:
                replayed["traffic_committed_profile"] = dict(bytes_per_launch=roof["traffic"], source=src)
            roof["traffic"], roof["traffic_source"] = live, how
            if getattr(live_pmc_traffic, "extra", None):       # matrix-pipe busy share of the same kernel, from a third live pass (emulation MFMAs included:
                roof.update(live_pmc_traffic.extra)            # NOT comparable with `frac`, which counts algorithmic flops)
        else:
            roof["traffic_live_failed"] = how
    if src and "traffic_source" in roof and roof["traffic_source"].startswith("measured by this run"):
        if replayed is not None:
            pm = dict(source=src, kernel=dom)
            if isinstance(pmc_util, dict):
                pm.update(pmc_mfma_busy=pmc_util["at_clock_held"], mfma_busy_at_peak_clock=pmc_util["at_peak_clock"], clock_held_ghz_under_pmc=pmc_util["clock_held_ghz"])
            replayed["pmc"] = pm
    elif src:
        # PMC counters cannot be read from inside this process: `traffic` (a key the contract prescribes) is the committed rocprofv3 --pmc
        # figure of the same kernel, precision and row count, and says so; every other replayed figure lives under `replayed_from_profiles`
        roof["traffic_source"] = "REPLAYED, not measured by this run: " + src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/kernel_bench.py, bytes per launch)"
        if replayed is not None:
            pm = dict(source=src, kernel=dom)
            if isinstance(pmc_util, dict):
                # PMC matrix-pipe busy share: emulation MFMAs included (bf16x3 issues 3 per product in the forward), so NOT comparable with
                # `frac` (algorithmic flops / peak); at the clock the chip held, and as a share of the issue slots at the peak clock
                pm.update(pmc_mfma_busy=pmc_util["at_clock_held"], mfma_busy_at_peak_clock=pmc_util["at_peak_clock"], clock_held_ghz_under_pmc=pmc_util["clock_held_ghz"])
            elif pmc_util is not None:
                pm["pmc_mfma_busy"] = pmc_util
            replayed["pmc"] = pm
    roof["algorithmic_per_launch"] = wgrad_bytes if dom == "wgrad" else flops
    roof["mfmas_per_product"] = MFMA_PER_PRODUCT[prec_name][dom]
    roof["all_kernels"] = {k: dict(launch_ms=round(v["launch_ms"], 4), coarse_launch_ms=round(v["coarse_launch_ms"], 4), avg_launch_ms=round(v["avg_launch_ms"], 4),
                                   achieved=round(v["achieved"], 2), unit=v["unit"], frac=round(v["frac"], 4), frac_coarse_plus_fine=round(v["frac_coarse_plus_fine"], 4),
                                   **({"mfma_issue_util": round(v["mfma_issue_util"], 4)} if "mfma_issue_util" in v else {}))
                           for k, v in entries.items()}
    roof["avg_launch_note"] = ("avg_launch_ms = (coarse 262 144-row + fine 786 432-row launch) / 2: the figure `rocprofv3 --kernel-trace --stats` of this command "
                               "reports as the kernel's average duration (one coarse and one fine launch per step)")
    return roof


def psnr_vs_reference(precision, device, steps=300, max_seconds=75.0):
    """BASELINE's second metric, measured in this run: tests/tools/psnr_curve.py (HIP path vs the oracle
    as fp32 PyTorch-ROCm ops on this GPU, identical init / rays / draws, 4096 rays x (64+128) per step)
    for `steps` steps or `max_seconds`; plus the committed long curves."""
    from tests.tools import psnr_curve as PC
    args = PC.parse(["--config", "1", "--steps", str(steps), "--modes", precision, "--eval-every", str(max(1, steps // 3)),
                     "--eval-rays", "2048", "--grad-check-at", "-1", "--max-seconds", str(max_seconds), "--quiet"])
    doc = PC.run(args, device)
    fin = doc["final"]
    out = dict(steps=doc["steps_done"], rays_per_step=doc["rays_per_step"], samples=doc["samples"], scene=doc["scene"],
               psnr_hip=fin[precision]["psnr"], psnr_ref=fin["oracle_fp32"]["psnr"], psnr_delta=doc["psnr_delta_vs_reference"][precision],
               curve=[dict(step=r["step"], hip=r[precision]["psnr"], ref=r["oracle_fp32"]["psnr"]) for r in doc["curve"]],
               reference="oracle/nerf_oracle.py as fp32 PyTorch-ROCm ops on the same GPU, torch.optim.Adam + clip_grad_norm_(0.1)",
               seconds=doc["seconds"])
    long_runs = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_psnr_curve_c*.json"))):
        try:
            d = json.load(open(f))
            long_runs[os.path.relpath(f, ROOT)] = dict(config=d["config"], steps=d["steps_done"], final={k: v for k, v in d["final"].items() if k != "step"},
                                                         psnr_delta_vs_reference=d["psnr_delta_vs_reference"])
        except (OSError, ValueError, KeyError):
            continue
    if long_runs:
        out["committed_long_runs"] = long_runs
    return out


def live_parity(precision, device):
    """One BASELINE config-1 case (4 x 1024 rays x (64+128), sigma noise) of tests/scale_cases.py through the
    public API in this process, against the oracle's float64 referee running on the same GPU."""
    from tests import scale_cases as S
    r = S.run_case(1, precision, device=device, referee_device=str(device), chunk=1024, log=lambda *a: None)
    e = r["hip"]
    return dict(config=1, rays=r["rays"], rendered_outputs_max_rel=e["rendered_worst"], per_sample_outputs_max_rel=e["per_sample_worst"],
                param_grad_rel_l2_worst_tensor=e["param_grad_rel_l2_worst"], param_grad_rel_l2_all=e["param_grad_rel_l2_all"],
                t_coarse_bit_exact=r["t_coarse_bit_exact"], t_fine_vs_sampler_oracle_maxabs=r["t_fine_vs_sampler_oracle_maxabs"],
                referee="oracle float64 (tests/scale_cases.referee) on this GPU, measured in this run", seconds=r["referee_seconds"])


def reduce_leg(dt, nrays, steps, world, device):
    """One timed leg of the contract: K steps took `dt` seconds on this rank and rendered `nrays` rays.  -> whole-job figures:
    the time is the MAX over ranks, the rays the SUM (value = units all ranks processed / that time), plus the per-rank
    ms/step spread.  Collective calls: every rank must call it."""
    import torch.distributed as dist
    if world > 1:
        tt = torch.tensor([dt, float(nrays)], device=device, dtype=torch.float64)
        tmax, tmin, tsum = tt[:1].clone(), tt[:1].clone(), tt[1:].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        rank_ms = dict(min=float(tmin.item()) / steps * 1e3, max=float(tmax.item()) / steps * 1e3)
        dt_all, nrays_all = float(tmax.item()), float(tsum.item())
    else:
        rank_ms, dt_all, nrays_all = None, dt, float(nrays)
    return dict(value=nrays_all / dt_all, unit="rays/s", ms_per_step=dt_all / steps * 1e3, steps=steps, rays_per_step_all_ranks=nrays_all / steps,
                per_rank_ms_per_step=rank_ms, seconds=dt_all)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=None,
                    help="untimed steps before the K timed ones (default 5; 20 for the eager configs 3 / 4, whose ray counts are data-dependent: "
                         "the caching allocator and the clocks settle over the first ~0.5 s, profiles/r04_config3_warmup.log)")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[i]: 1 = 4096 rays x (64+128), fixed poses (the config the metric is quoted on); 2 = joint "
                         "pose-NeRF step (c2f + SE(3)); 3 = LLFF-shaped SPARF call mix; 4 = Replica-shaped, 9 views, SPARF call mix")
    ap.add_argument("--precision", default=default_precision(), choices=["bf16", "fp32", "bf16x3", "bf16+q8", "bf16x3+q8"],
                    help="headline mode; default bf16x3 = the fastest mode whose outputs meet the 1e-4 parity bar "
                         "(bf16 MFMA, operands split in head + tail); the other modes are measured briefly and reported in `other_modes`")
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU (weak scaling, the default) or in total (--strong)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --rays is the global batch, each rank renders rays/N")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1: skip the short strong-scaling leg that follows the (weak) contract region")
    ap.add_argument("--batched", action="store_true", help="configs 3 / 4: issue the independent render calls of an iteration through Graph.render_batch")
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"],
                    help="fused: sparf_amd.optim.FusedAdam (clip + Adam, 2 launches per network); torch: torch.optim.Adam + clip_grad_norm_")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed PMC profile instead of two live rocprofv3 --pmc passes (~40 s)")
    ap.add_argument("--no-telemetry", action="store_true", help="skip the clock / power sampler and the calibration kernels (bench_telemetry.py)")
    ap.add_argument("--no-psnr", action="store_true", help="skip the side-by-side training run against the oracle (psnr_vs_ref)")
    ap.add_argument("--no-live-parity", action="store_true", help="skip the in-run parity spot check (parity_live)")
    ap.add_argument("--min-seconds", type=float, default=3.0, help="length of the sustained loop that follows the K contract steps (0 = skip)")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the brief measurements of the other precision modes")
    ap.add_argument("--no-other-sizes", action="store_true", help="skip the small-batch measurements (512 / 1024 / 2048 rays per step, eager and as one hipGraph)")
    ap.add_argument("--graph", action="store_true", help="configs 1 / 2, one GPU: replay the whole step as ONE captured hipGraph (Workload.capture) in the timed region")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly: become the launcher (one rank per GPU over RCCL), pass the arguments through
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch.distributed as dist
    from bench_workloads import SHAPES, Workload
    from sparf_amd.parallel import GradBucket, broadcast_parameters

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = local % max(1, torch.cuda.device_count())      # (one-GPU boxes: lets a gloo functional test run N ranks on cuda:0)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("SPARF_DIST_BACKEND", "nccl")         # nccl = RCCL over xGMI; gloo only for functional tests
        dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    if args.strong:
        args.rays = max(SHAPES[args.config]["B"], args.rays // world)

    def buckets_for(w):
        """gradient exchange of a workload: ONE all-reduce per step -- both networks' flat gradient buffers, the pose
        parameters' gradients and the loss scalar + a NaN flag (iter_based_trainer.py:248-252) in a single message"""
        if world == 1:
            return None
        bucket = GradBucket(w.net_params + ([w.graph.se3_refine] if w.optim_pose is not None else []))

        def exchange(loss):
            w.last_scalars = bucket.allreduce_(extra=torch.stack([loss.detach(), torch.isnan(loss.detach()).float()]))
        w.bucket = bucket
        return exchange

    use_graph = args.graph and args.config in (1, 2) and args.optimizer == "fused"
    graph_strong = world > 1 and args.config in (1, 2) and args.optimizer == "fused"       # the strong leg's small steps: two hipGraphs around the all-reduce

    def make(precision):
        w = Workload(args.config, precision, device, rays=args.rays, optimizer=args.optimizer, batched=args.batched, bucket_factory=buckets_for,
                     graph_capture=use_graph)
        if world > 1:
            broadcast_parameters(w.graph)
        return w

    w = make(args.precision)
    torch.cuda.manual_seed(1234 + rank)                                # each rank: its own ray shard / draws
    if use_graph:
        w.step = w.capture()                                           # the same iteration, replayed as one hipGraph

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(wl, steps, warmup):
        """the contract's timed region: W untimed steps, then exactly K steps between barrier + synchronize on both sides"""
        for _ in range(warmup):
            wl.step()
        sync()
        n, last = 0, None
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]       # one event record per step: ~2 us of host time each
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(steps):
            last = wl.step()
            n += wl.rays_last
            evs[i + 1].record()
        sync()
        dt_region = time.perf_counter() - t0
        timed.step_ms = [round(evs[i].elapsed_time(evs[i + 1]), 4) for i in range(steps)]
        return dt_region, n, last

    if args.warmup is None:
        args.warmup = 20 if args.config in (3, 4) else 5
    # what the box does while it is measured (bench_telemetry.py): clock / power / temperature at 50 Hz from a side thread, and the
    # library's two fixed calibration kernels before and after the measurement (rank 0; every rank idles behind the barrier meanwhile)
    sampler = calib = None
    box = {}
    if rank == 0 and not args.no_telemetry:
        import bench_telemetry as BT
        sampler = BT.Sampler(local).start()
        try:
            calib = BT.Calibration(device)
            with sampler.window("calib_before"):
                box["calib_before"] = calib.run()
        except Exception as exc:
            calib, box["calib_before"] = None, f"failed: {type(exc).__name__}: {str(exc)[:200]}"
    win = (lambda name: sampler.window(name)) if sampler is not None else (lambda name: contextlib.nullcontext())
    for _ in range(args.warmup):           # (outside the sampled window; timed() then runs no further warm-up)
        w.step()
    with win("contract"):
        dt, nrays, loss = timed(w, args.steps, 0)
    contract_step_ms = timed.step_ms
    leg = reduce_leg(dt, nrays, args.steps, world, device)
    rank_ms, dt, nrays_all, value = leg["per_rank_ms_per_step"], leg["seconds"], leg["rays_per_step_all_ranks"] * args.steps, leg["value"]
    # sustained loop: the same steps for >= --min-seconds, one event per step (the contract region above is 0.15 s at
    # K = 20: a burst on a chip that clocks down under MFMA load)
    sustained = None
    if args.min_seconds > 0:
        n_sus = max(args.steps, int(math.ceil(args.min_seconds / (dt / args.steps))))      # from the rank-reduced dt: the same count on every rank
        evs, rays_seq = [torch.cuda.Event(enable_timing=True)], []
        with win("sustained"):
            evs[0].record()
            for _ in range(n_sus):
                w.step()
                rays_seq.append(w.rays_last)
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
            sync()
        ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(rays_seq))]
        tot_ms = sum(ms)
        srt = sorted(ms)
        acc, r_last, k = 0.0, 0, len(ms)
        while k > 0 and acc < 1000.0:
            k -= 1
            acc += ms[k]
            r_last += rays_seq[k]
        sustained = dict(steps=len(ms), seconds=tot_ms * 1e-3, value=sum(rays_seq) / (tot_ms * 1e-3) * world, value_last_second=r_last / (acc * 1e-3) * world,
                         ms_per_step_mean=tot_ms / len(ms), ms_per_step_p50=srt[len(srt) // 2], ms_per_step_p95=srt[min(len(srt) - 1, int(len(srt) * 0.95))],
                         ms_per_step_min=srt[0], ms_per_step_max=srt[-1], note="rank 0 event timing; value = rank-0 rate x ranks")
    if calib is not None:
        try:
            with sampler.window("calib_after"):
                box["calib_after"] = calib.run()
        except Exception as exc:
            box["calib_after"] = f"failed: {type(exc).__name__}: {str(exc)[:200]}"
        calib.release()
        calib = None
    if world > 1:
        dist.barrier()
    # N > 1: ONE driver command yields both scaling numbers SURVEY 8e asks for -- the contract line above is the weak leg
    # (--rays per GPU), then a short strong leg: the same global batch split over the ranks (--rays / N per GPU)
    scaling_legs = None
    if world > 1 and not args.strong and not args.no_strong_leg:
        weak = dict(leg, rays_per_gpu_per_step=nrays / max(1, args.steps))
        r_strong = max(SHAPES[args.config]["B"], args.rays // world)
        ws = Workload(args.config, args.precision, device, rays=r_strong, optimizer=args.optimizer, batched=args.batched, bucket_factory=buckets_for,
                      graph_capture=graph_strong)
        broadcast_parameters(ws.graph)
        if graph_strong:
            # every rank must take the same path: a rank that cannot capture (an exotic driver / RCCL combination) sends all of them to
            # the eager step, and the line says so
            ok = torch.ones((), device=device)
            try:
                captured = ws.capture()
            except Exception as exc:
                captured, ok = None, torch.zeros((), device=device)
                print(f"[bench] rank {rank}: strong leg falls back to eager steps ({type(exc).__name__}: {str(exc)[:200]})", file=sys.stderr, flush=True)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            graph_strong = bool(ok.item() > 0)
            if graph_strong:
                ws.step = captured
            else:
                del ws
                torch.cuda.empty_cache()
                ws = Workload(args.config, args.precision, device, rays=r_strong, optimizer=args.optimizer, batched=args.batched, bucket_factory=buckets_for)
                broadcast_parameters(ws.graph)
        k_strong = max(args.steps, 30)
        sdt, sn, _ = timed(ws, k_strong, 5)
        strong = dict(reduce_leg(sdt, sn, k_strong, world, device), rays_per_gpu_per_step=sn / k_strong,
                      launch="two hipGraphs (forward + backward | clip + Adam) around the eager all-reduce" if graph_strong else "eager")
        scaling_legs = dict(weak=weak, strong=strong,
                            note="weak: --rays per GPU (the contract line's value); strong: --rays in total, --rays / N per GPU; efficiency is the driver's to compute")
        del ws
        torch.cuda.empty_cache()
    s = SHAPES[args.config]
    rays_step = nrays / max(1, args.steps)
    workload = (f"BASELINE configs[{args.config}]: {s['what']}; {s['B']} views {s['H']}x{s['W']}, {rays_step:.0f} rays x (64 coarse + 128 fine) "
                f"per GPU and step, fwd+bwd+clip+Adam, both 8x256 MLPs")
    line = {
        "metric": "training rays/sec (64c+128f samples, 8x256 MLP)", "value": value, "unit": "rays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": workload, "baseline_config": args.config, "rays_per_gpu_per_step": rays_step, "samples": "64+128",
                   "precision_mode": args.precision, "launch": ("hipGraph (whole step captured, Workload.capture)" if world == 1 else "two hipGraphs around the eager all-reduce (Workload.capture)") if use_graph else "eager",
                   "arithmetic": {"bf16x3": "bf16 MFMA, every fp32 operand split into bf16 head + tail (3 products forward, 2 dgrad, 1 wgrad), fp32 accumulate",
                                  "bf16": "bf16 MFMA operands, fp32 accumulate", "fp32": "fp32 MFMA (exact fp32 FMA chains)"}[args.precision.split("+")[0]]
                                 + (", 8-bit save / gradient areas (linear grid, one step per row and vector)" if args.precision.endswith("+q8") else ""),
                   "render_calls": ("separate calls, as the unmodified losses issue them (the two back-to-back correspondence renders meet in one launch "
                                    "set: Graph lazy batching, opt.hip.lazy_batch)") if not args.batched else "Graph.render_batch",
                   "optimizer": "clip_grad_norm(0.1) + Adam, " + ("sparf_amd.optim.FusedAdam" if args.optimizer == "fused" else "torch"),
                   "parallelism": f"dp{world} (ray-batch sharded; ONE all-reduce per step: both networks' flat gradients" + (" + pose gradients" if args.config != 1 else "") + " + loss / NaN scalars)"},
        "final_loss": float(loss.item()),
        "per_rank_ms_per_step": rank_ms,
        "scaling_legs": scaling_legs,
        "rccl_ranks": world if (world > 1 and os.environ.get("SPARF_DIST_BACKEND", "nccl") == "nccl") else 0,
        "collectives_per_step": (getattr(w, "bucket", None).collectives if getattr(w, "bucket", None) is not None else 0),
        "sustained": sustained,
        # whole-step algorithmic MFMA fraction: rays/s x 810.8 MFLOP / dense peak of the operand type
        "mfma_fraction_of_step": value / world * FLOP_TRAIN_RAY / (PEAK[args.precision] * 1e12) if args.config in (1, 2) else None,
    }
    if rank == 0:
        replayed = {}          # everything on this line that was NOT measured by this run (committed profiles of other sessions), in one place
        if sampler is not None:
            sampler.stop()
            tel = sampler.summary()
            line["telemetry"] = dict(tel, calib_before=box.get("calib_before"), calib_after=box.get("calib_after"))
            line["config"]["box"] = box_summary(tel, box, value / world, contract_step_ms, sustained)
        if not args.no_roofline:
            line["roofline"] = kernel_roofline(w.graph, w.opt, args.precision, device, rays=4096, replayed=replayed,
                                               live_pmc=(world == 1 and not args.no_live_pmc))
            if "box" in line["config"]:
                line["roofline"]["box"] = line["config"]["box"]
            if args.config in (1, 2):       # whole step: algorithmic FLOP per step / measured step time / dense peak
                ms_step = sustained["ms_per_step_p50"] if sustained else dt / args.steps * 1e3
                line["roofline"]["step"] = dict(algorithmic_flop=rays_step * FLOP_TRAIN_RAY, ms_per_step=ms_step, peak=PEAK[args.precision], unit="TFLOP/s",
                                                achieved=rays_step * FLOP_TRAIN_RAY / (ms_step * 1e-3) / 1e12,
                                                frac=rays_step * FLOP_TRAIN_RAY / (ms_step * 1e-3) / 1e12 / PEAK[args.precision])
        if world == 1 and not args.no_live_parity:
            try:
                line["parity_live"] = live_parity(args.precision, device)
            except Exception as exc:                          # the checker must not take the measurement down with it
                line["parity_live"] = f"failed: {type(exc).__name__}: {exc}"
        par = measured_parity(args.precision)
        replayed["parity"] = par if par is not None else "no committed profiles/r*_parity_scale.json for this mode"
        if world == 1 and not args.no_other_modes:
            line["other_modes"] = {}
            for pm in ("bf16", "bf16x3", "fp32"):
                if pm == args.precision:
                    continue
                wp = make(pm)
                for _ in range(2):
                    wp.step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                nst, nr = (5 if pm == "fp32" else 10), 0
                for _ in range(nst):
                    wp.step()
                    nr += wp.rays_last
                torch.cuda.synchronize()
                pdt = time.perf_counter() - t1
                line["other_modes"][pm] = {"value": nr / pdt, "unit": "rays/s", "ms_per_step": pdt / nst * 1e3, "steps": nst, "parity": measured_parity(pm),
                                           "mfma_fraction_of_step": nr / pdt * FLOP_TRAIN_RAY / (PEAK[pm] * 1e12) if args.config in (1, 2) else None}
                del wp
                torch.cuda.empty_cache()
        if world == 1 and not args.no_other_sizes and args.config in (1, 2) and args.optimizer == "fused":
            # small batches (the reference's default rand_rays 1024 / 2048, default_config.py:118,256; 512 = a 4096-ray batch strong-scaled
            # over 8 GPUs): launch-bound when issued eagerly (~40 launches per step), so also as ONE captured hipGraph
            line["other_sizes"] = {}
            for rr in (512, 1024, 2048, 4096):
                entry = {}
                for how in ("eager", "hipgraph"):
                    try:
                        ws = Workload(args.config, args.precision, device, rays=rr, optimizer="fused", graph_capture=(how == "hipgraph"))
                        stepf = ws.capture() if how == "hipgraph" else ws.step
                        for _ in range(3):
                            stepf()
                        torch.cuda.synchronize()
                        nst = max(20, min(200, int(0.4 / (7.5e-3 * rr / 4096 + 1e-3))))
                        t1 = time.perf_counter()
                        for _ in range(nst):
                            stepf()
                        torch.cuda.synchronize()
                        pdt = time.perf_counter() - t1
                        entry[how] = {"value": ws.rays_last * nst / pdt, "unit": "rays/s", "ms_per_step": pdt / nst * 1e3, "steps": nst}
                    except Exception as exc:
                        entry[how] = f"failed: {type(exc).__name__}: {str(exc)[:200]}"
                    finally:
                        ws = stepf = None
                        torch.cuda.empty_cache()
                line["other_sizes"][str(rr)] = entry
        if world == 1 and not args.no_psnr:
            line["psnr_vs_ref"] = psnr_vs_reference(args.precision, device)
            if "committed_long_runs" in line["psnr_vs_ref"]:
                replayed["psnr_committed_long_runs"] = line["psnr_vs_ref"].pop("committed_long_runs")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            try:        # kind "port" (no staged reference): the port vs the reference module itself, same host / threads (tests/tools/cpu_ref_vs_port.py, build container)
                if line["cpu_baseline"]["kind"] == "reference":
                    raise KeyError("measured live")
                rp = json.load(open(os.path.join(ROOT, "profiles", "r03_cpu_ref_vs_port.json")))
                replayed["cpu_port_over_reference"] = rp["summary"]["port_over_reference"]
                replayed["cpu_port_over_reference_source"] = ("profiles/r03_cpu_ref_vs_port.json: reference Graph.render + backward vs the port, "
                                                                      f"{rp['host']['cpu']}, {rp['summary']['threads']} threads, range over thread counts "
                                                                      f"{rp['summary']['port_over_reference_range']}")
            except (OSError, ValueError, KeyError):
                pass
        if replayed:
            replayed["note"] = "figures of committed profiles measured in other sessions / on other machines; nothing else on this line is replayed"
            line["replayed_from_profiles"] = replayed
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
