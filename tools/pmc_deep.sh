#!/bin/bash
# Where do the waves of the fused kernels spend their cycles?  SQ counters in groups of <= 8 (one rocprofv3 pass each)
# over tools/kernel_bench.py.   [PMC_FILTER=sparf::wgrad] bash tools/pmc_deep.sh <prec> <outdir>
set -u
PREC=${1:-bf16x3}
OUT=${2:-gpurun_out/pmc_deep_$PREC}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o g$i -- python tools/kernel_bench.py $PREC > $OUT/g$i.log 2>&1
  echo "pass $i exit $?"; tail -2 $OUT/g$i.log | cut -c1-200
done
python - <<PY
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "${PMC_FILTER:-sparf::mlp_}" in r["Kernel_Name"]:
            vals[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(vals.items()):
    m = {n: sum(v[-5:]) / len(v[-5:]) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k)
    for n, v in sorted(m.items()):
        print("   %-28s %14.0f  %6.3f of wave cycles" % (n, v, v / wc))
PY
rm -f $OUT/*.db
