#!/bin/bash
# one quick bench line per call: value, sustained value, the calibration figures and the normalised figures (box-to-box survey; gpurun leases land on different GPUs)
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --no-cpu-baseline --no-psnr --no-other-modes --no-other-sizes --no-live-parity --no-live-pmc --steps 30 2>/dev/null | python -c '
import json, sys
d = json.loads(sys.stdin.read()); b = d["config"]["box"]; k = d["roofline"]["all_kernels"]
print(json.dumps(dict(device=d["telemetry"]["device"], value=round(d["value"]), sustained=round(d["sustained"]["value"]), calib_mfma=round(b["calib_mfma_tflops"], 1),
                      calib_mix=round(b["calib_mix_cycles_per_s"], 2), hbm=b["calib_hbm_read_tbs"], clock_contract=b["clock_ghz_mean"], clock_sustained=b["sustained_clock_ghz_mean"],
                      power_sustained=b["sustained_power_w_mean"], per_mfma=round(b["sustained_value_per_calib_mfma_tflop"], 1), per_mix=round(b["sustained_value_per_calib_mix_cycle"], 1),
                      fwd=k["mlp_fwd"]["launch_ms"], dgrad=k["mlp_dgrad"]["launch_ms"], wgrad=k["wgrad"]["launch_ms"])))' | tee -a gpurun_out/r06_box_survey.jsonl
