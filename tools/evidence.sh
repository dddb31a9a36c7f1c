#!/bin/bash
# Regenerates the measured evidence of round 6, one section per committed artefact under profiles/ (one entry
# point instead of forty session scripts).  Run on the GPU box from the repository root, e.g.
#     gpurun --timeout 900 -- 'bash tools/evidence.sh tests tape'
# Every section writes gpurun_out/r06_<name>.*; copy what is to be judged into profiles/.
#
#   tests        pytest -m gpu (the whole suite, no -x) + smoke()                                   -> r06_pytest_gpu.log
#   tape         the committed reference tapes replayed on the HIP renderer (also part of `tests`)   -> r06_reference_tape.json
#   live         OPT-IN reference-callers comparison (needs the staged archive, see below)          -> r06_reference_callers.json
#   seeds        free-running reference-callers comparison over 8 numpy seeds (needs the archive)   -> r06_reference_callers_seeds.json
#   host         host time of one render call + backward (tests/tools/host_overhead.py)             -> r06_host_overhead.log
#   bench        bench.py default line                                                               -> r06_bench_bf16x3.json
#   configs      bench.py --config 2 / 3 / 4                                                         -> r06_bench_c{2,3,4}.json
#   kernels      per-kernel launch times of every precision mode (tools/kernel_bench.py)            -> r06_kernel_bench.log
#   parity       float64-referee parity at the BASELINE shapes (tests/tools/scale_parity.py)        -> r06_parity_scale.json
#   rocprof      rocprofv3 --kernel-trace --stats of the bench command                               -> r06_bf16x3_kernel_stats.csv
#   pmc          rocprofv3 --pmc passes over tools/kernel_bench.py (separate passes, no trace domains) -> r06_pmc_bf16x3.json
#   gaps         GPU idle share of configs 3 / 4 (tools/gap_analysis.py)                             -> r06_gap_analysis.log
#   poseseeds    oracle / HIP fp32 / HIP bf16x3 trained side by side over seeds: PSNR and pose error as distributions -> r06_pose_seeds_c{2,3}.json
#   ab           per-kernel timings of variant libraries next to the default build (AB_TAGS, AB_PRECS)  -> r06_kernel_ab_<AB_NAME>.log
#   fwdprobes    the same for the bf16x3 training forward (f0..f4 libraries over mlp_fwd_x3_train.hip)          -> r06_fwd_lap_table.log
#   probe        which clock / power sensors the box offers + the calibration kernels (bench_telemetry.py)   -> r06_telemetry_probe.log
#   geometry     bf16x3 data-gradient kernel, 8-wave vs 4-wave geometry by row count (api.hip x3_dgrad_waves) -> r06_dgrad_geometry.log
#   smallstep    rocprofv3 kernel stats of a 512-ray step replayed as one hipGraph                    -> r06_r512_kernel_stats.csv
#   registration joint pose-NeRF registration, oracle and HIP side by side (tests/tools/registration_run.py) -> r06_registration.json
#   fwdprobes6   the bf16x3 training forward without its encoding / tile end (upper bound of a cross-tile pipeline) -> r06_fwd_pipeline_probes.log
#   q8halves     8-bit weight-gradient jobs as half jobs, two resident workgroups per CU (variant library)  -> r06_wgrad_q8_halves.log
#   dgradprobes  wave-time accounting ("lap table") of the data-gradient kernel + its timing probes (variant libraries of tools/build_flag_variant.py) -> r06_dgrad_lap_table.log
#
# live / seeds: the reference tree is NOT part of the repository snapshot.  A builder who wants these sections packs it first, in the
# build container:   python oracle/stage_reference.py --out oracle/_ref/reference_tree.zip      (git-ignored; delete it afterwards)
set -u
TAG=${TAG:-r06}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
REFZIP=oracle/_ref/reference_tree.zip
quick="--no-cpu-baseline --no-psnr --no-other-modes --no-other-sizes --no-live-parity --no-live-pmc"
for sec in "$@"; do
  echo "=================== $sec"
  case $sec in
    tests)
      timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -v "Warning\|warnings.warn\|^$" gpurun_out/${TAG}_pytest_gpu.log | tail -40
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v Warn | tail -4 | tee gpurun_out/${TAG}_smoke.log ;;
    tape)
      timeout 900 python -m pytest tests/test_01_reference_tape_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -30
      python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_reference_tape.json"))
for k, r in d.items():
    pc = [{kk: (f"{v:.1e}" if isinstance(v, float) else v) for kk, v in e.items() if not kk.startswith("_")} for e in r["per_call"]]
    print(k, "worst", r.get("grad_worst_name"), f"{r.get('grad_worst_tensor', 0):.2e}", "all", f"{r.get('grad_all', 0):.2e}", "norm", f"{r.get('grad_norm_ratio_worst', 0):.1e}")
    for i, e in enumerate(pc):
        print("   call", i, r["calls"][i], e)
PY
      ;;
    live)
      [ -f $REFZIP ] || { echo "no $REFZIP: section skipped"; continue; }
      SPARF_REFERENCE_ROOT=$REFZIP timeout 1200 python -m pytest tests/test_reference_callers_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -30 ;;
    seeds)
      [ -f $REFZIP ] || { echo "no $REFZIP: section skipped"; continue; }
      SPARF_REFERENCE_ROOT=$REFZIP timeout 1200 python tests/tools/reference_callers_seeds.py 2>&1 | grep -v "Warning\|warnings.warn\|meshgrid\|Computing\|possible flow" | tail -60 ;;
    host)
      timeout 600 python tests/tools/host_overhead.py bf16x3 --profile 2>&1 | grep -v "Warning\|warnings.warn" | tee gpurun_out/${TAG}_host_overhead.log | head -110 ;;
    bench)
      timeout 900 python bench.py > gpurun_out/${TAG}_bench_bf16x3.json 2> gpurun_out/${TAG}_bench_bf16x3.err; cut -c1-600 gpurun_out/${TAG}_bench_bf16x3.json; tail -3 gpurun_out/${TAG}_bench_bf16x3.err ;;
    benchquick)
      timeout 600 python bench.py $quick --steps 30 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("config 1:", round(d["value"]), "rays/s", round(d["ms_per_step"],3), "ms;", {k:(v["launch_ms"]) for k,v in d["roofline"]["all_kernels"].items()}, "sustained", d["sustained"] and round(d["sustained"]["value"]))' ;;
    configs)
      for c in 2 3 4; do
        timeout 500 python bench.py --config $c --no-cpu-baseline --no-psnr --no-roofline --no-other-sizes --no-other-modes --no-live-parity --steps 20 > gpurun_out/${TAG}_bench_c$c.json 2> gpurun_out/${TAG}_bench_c$c.err
        python -c 'import sys,json; d=json.loads(open(sys.argv[1]).read()); print("config", sys.argv[2], round(d["value"]), "rays/s", round(d["ms_per_step"],2), "ms; sustained", d["sustained"] and round(d["sustained"]["value"]))' gpurun_out/${TAG}_bench_c$c.json $c || tail -3 gpurun_out/${TAG}_bench_c$c.err
      done ;;
    kernels)
      for P in ${KERNEL_PRECS:-bf16x3 bf16x3+q8 bf16 bf16+q8}; do timeout 300 python tools/kernel_bench.py $P 2>&1 | grep -v "Warning\|warnings.warn\|amdgpu.ids"; done | tee gpurun_out/${TAG}_kernel_bench.log ;;
    parity)
      timeout 1200 python tests/tools/scale_parity.py --yardstick --referee-device cuda:0 --out gpurun_out/${TAG}_parity_scale.json 2>&1 | grep '^{' | cut -c1-240 ;;
    rocprof)
      mkdir -p gpurun_out/prof
      timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_bf16x3 -- python bench.py $quick > gpurun_out/${TAG}_prof_bench_bf16x3.log 2>&1
      python tools/prof_summary.py gpurun_out/prof/${TAG}_bf16x3_results.db gpurun_out/${TAG}_bf16x3_kernel_stats.csv; head -10 gpurun_out/${TAG}_bf16x3_kernel_stats.csv | cut -c1-120,160-
      rm -rf gpurun_out/prof ;;
    pmc)
      for P in ${PMC_PRECS:-bf16x3}; do
        bash tools/pmc_profile.sh ${TAG}_$P $P | grep "pass "
        python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$P gpurun_out/${TAG}_pmc_$P | grep "mlp_\|wgrad_kernel" | cut -c1-260
        rm -rf gpurun_out/pmc_${TAG}_$P
      done ;;
    gaps)
      mkdir -p gpurun_out/prof
      for c in 3 4; do
        timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -o gaps_c$c -- python bench.py --config $c --steps 20 --warmup 20 --min-seconds 0 $quick --no-roofline --no-telemetry > gpurun_out/prof/gaps_c$c.log 2>&1
        echo "config $c: $(grep -o '"value": [0-9.]*' gpurun_out/prof/gaps_c$c.log | head -1) rays/s under the profiler"
        python tools/gap_analysis.py "$(find gpurun_out/prof -name "*gaps_c${c}*kernel_trace.csv" | head -1)" | head -16
      done 2>&1 | tee gpurun_out/${TAG}_gap_analysis.log
      rm -rf gpurun_out/prof ;;
    poseseeds)
      timeout 1500 python tests/tools/pose_seeds.py --config 2 --seeds ${POSE_SEEDS:-5} --steps ${POSE_STEPS:-1000} --out gpurun_out/${TAG}_pose_seeds_c2.json 2>&1 | grep -v "Warning\|warnings.warn" | tail -50
      timeout 1500 python tests/tools/pose_seeds.py --config 3 --seeds ${POSE_SEEDS3:-3} --steps ${POSE_STEPS:-1000} --out gpurun_out/${TAG}_pose_seeds_c3.json 2>&1 | grep -v "Warning\|warnings.warn" | tail -40 ;;
    ab)     # same-box A/B of kernel variants: AB_TAGS="wgspread bwdspread" (sparf_amd/libsparf_hip_<tag>.so, tools/build_variant.py / sparf_amd.build.build(tag=...))
      AB_PRECS="${AB_PRECS:-bf16x3 bf16x3+q8}" bash tools/ab_kernels.sh ${AB_TAGS:-} 2>&1 | tee gpurun_out/${TAG}_kernel_ab_${AB_NAME:-variants}.log ;;
    dgradprobes)   # wave-time accounting of the data-gradient kernel (mlp_dev.h Prof) and its timing probes; build the libraries first:
      #   for v in "p0:-DSP_PROF" "p1:-DSP_PROF -DSP_PROBE_NO_STORES" "p2:-DSP_PROF -DSP_PROBE_NO_DMA" "p3:-DSP_PROF -DSP_PROBE_NO_DMA -DSP_PROBE_NO_STORES" \
      #            "p4:-DSP_PROF -DSP_PROBE_NO_DMA -DSP_PROBE_NO_STORES -DSP_PROBE_NO_BARRIER" "nodeferp:-DSP_PROF -DSP_BWD_DEFER=0"; do
      #     python tools/build_flag_variant.py ${v%%:*} "${v#*:}" mlp_bwd.hip; done
      for tag in ${PROBE_TAGS:-nodeferp p0 p1 p2 p3 p4}; do
        [ -f sparf_amd/libsparf_hip_$tag.so ] || continue
        for P in ${PROBE_PRECS:-bf16x3 bf16}; do
          echo "== lib $tag prec $P"
          SPARF_LIB=$PWD/sparf_amd/libsparf_hip_$tag.so timeout 300 python tools/kernel_bench.py $P 2>&1 | grep -A1 "^dgrad"
        done
      done | tee gpurun_out/${TAG}_dgrad_lap_table.log ;;
    fwdprobes)     # the same for the bf16x3 training forward (the dominant kernel); libraries: as above with f0..f4 and mlp_fwd_x3_train.hip
      for tag in ${PROBE_TAGS:-f0 f1 f2 f3 f4}; do
        [ -f sparf_amd/libsparf_hip_$tag.so ] || continue
        echo "== lib $tag prec bf16x3"
        SPARF_LIB=$PWD/sparf_amd/libsparf_hip_$tag.so timeout 300 python tools/kernel_bench.py bf16x3 2>&1 | grep -A1 "^fwd save"
      done | tee gpurun_out/${TAG}_fwd_lap_table.log ;;
    fwdprobes6)   # round 6: what a cross-tile software pipeline of the bf16x3 training forward could hide at most -- the kernel without its
      # encoding (15 sincosf per lane half) and / or without its tile end (sigmoid + colour stores); WRONG RESULTS, timing only.  Libraries:
      #   for v in "e0:-DSP_PROBE_NO_ENCODING" "e1:-DSP_PROBE_NO_TILE_END" "e2:-DSP_PROBE_NO_ENCODING -DSP_PROBE_NO_TILE_END"; do
      #     python tools/build_flag_variant.py ${v%%:*} "${v#*:}" mlp_fwd_x3_train.hip; done
      for rep in 1 2 3; do for tag in default e0 e1 e2; do
        if [ "$tag" = default ]; then unset SPARF_LIB; else [ -f sparf_amd/libsparf_hip_$tag.so ] || continue; export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_$tag.so; fi
        echo "== rep $rep lib $tag: $(KB_ONLY='fwd save' timeout 300 python tools/kernel_bench.py bf16x3 2>&1 | grep '^fwd save')"
      done; done | tee gpurun_out/${TAG}_fwd_pipeline_probes.log
      unset SPARF_LIB ;;
    registration) # joint pose-NeRF registration, oracle (torch ops) and HIP renderer side by side (tests/tools/registration_run.py)
      timeout ${REG_TIMEOUT:-2400} python tests/tools/registration_run.py --steps ${REG_STEPS:-3000} --seeds ${REG_SEEDS:-3} ${REG_ARGS:-} --out gpurun_out/${TAG}_registration.json 2>&1 | grep -v "Warning\|warnings.warn" | tail -60 ;;
    q8halves)     # round 6 experiment: the 8-bit weight-gradient jobs as half jobs, two resident workgroups per CU (wgrad.hip SP_WG_Q8_HALVES).
      # Library:  python tools/build_flag_variant.py q8h "-DSP_WG_Q8_HALVES=1" wgrad.hip     (profiles/r06_wgrad_q8_halves.log was taken with a
      # run-time switch between the same two kernel sets, before the switch became this build flag)
      [ -f sparf_amd/libsparf_hip_q8h.so ] || { echo "no sparf_amd/libsparf_hip_q8h.so: section skipped"; continue; }
      SPARF_LIB=$PWD/sparf_amd/libsparf_hip_q8h.so timeout 600 python -m pytest tests/test_q8_saves_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -4
      for rep in 1 2 3; do for h in 0 1; do for P in bf16x3+q8 bf16+q8; do
        if [ $h = 1 ]; then export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_q8h.so; else unset SPARF_LIB; fi
        echo "== rep $rep halves $h prec $P: $(KB_ONLY=wgrad,pass timeout 300 python tools/kernel_bench.py $P 2>&1 | grep -E '^(wgrad|pass bwd)' | tr '\n' ' ')"
      done; done; done | tee gpurun_out/${TAG}_wgrad_q8_halves.log
      unset SPARF_LIB ;;
    lazyab)       # configs 3 / 4 with and without lazy batching of the back-to-back correspondence renders, same box, alternating
      for rep in 1 2 3; do for lz in 1 0; do for c in 3 4; do
        echo "== rep $rep config $c lazy $lz: $(SPARF_LAZY_BATCH=$lz timeout 400 python bench.py --config $c --steps 20 $quick --no-roofline --no-telemetry | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "rays/s", round(d["ms_per_step"],2), "ms; sustained", round(d["sustained"]["value"]))')"
      done; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_lazy_batch_ab.log ;;
    smallstep)    # kernel time line of a 512-ray step replayed as one hipGraph (a 4096-ray batch strong-scaled over 8 GPUs)
      mkdir -p gpurun_out/prof
      timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_r512 -- python bench.py --rays ${SMALL_RAYS:-512} --graph --steps 300 --warmup 20 --min-seconds 0 $quick --no-roofline --no-telemetry > gpurun_out/${TAG}_prof_r512.log 2>&1
      grep -o '"value": [0-9.]*' gpurun_out/${TAG}_prof_r512.log | head -1
      python tools/prof_summary.py gpurun_out/prof/${TAG}_r512_results.db gpurun_out/${TAG}_r512_kernel_stats.csv; head -24 gpurun_out/${TAG}_r512_kernel_stats.csv | cut -c1-110,150-
      rm -rf gpurun_out/prof ;;
    geometry)     # bf16x3 data-gradient kernel: 256-row (8 waves) vs 128-row (4 waves) workgroup tiles by row count (api.hip x3_dgrad_waves)
      for R in ${GEOM_RAYS:-512 1024 1536 2048 4096}; do for N in 64 192; do
        echo "== rays $R samples $N rows $((R*N))"
        KB_ONLY=dgrad timeout 300 python tools/kernel_bench.py bf16x3 $R $N 2>&1 | grep "^dgrad"
      done; done | tee gpurun_out/${TAG}_dgrad_geometry.log ;;
    probe)        # which clock / power sensors the box offers + the calibration kernels (bench_telemetry.py)
      timeout 300 python tools/telemetry_probe.py 2>&1 | grep -v "Warning\|warnings.warn" | tee gpurun_out/${TAG}_telemetry_probe.log | cut -c1-1500 ;;
    *) echo "unknown section $sec" ;;
  esac
done
du -sh gpurun_out
