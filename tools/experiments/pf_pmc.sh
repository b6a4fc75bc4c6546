#!/bin/bash
# PMC passes over the GPS pre-filter chain kernel (1 000 logs): issue mix, wait shares and instruction-cache counters.
# usage (GPU box): bash tools/experiments/pf_pmc.sh  -> gpurun_out/pf_pmc/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pf_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
grep -oE "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_CACHE|INSTR)[A-Z0-9_]*)\b" $OUT/avail.txt | sort -u > $OUT/icache_counters.txt
i=0
for p in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_IFETCH_LEVEL" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1)); rm -rf $OUT/p$i
  timeout -k 10 200 rocprofv3 --pmc $p --output-format csv -d $OUT/p$i -- python3 $R/tools/experiments/prefilter_timing.py > $OUT/p$i.log 2>&1 || echo "pass $i ($p) failed" >> $OUT/summary_err.txt
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gps_prefilter_chain_kernel" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(f"{k:28s} {sum(acc[k]) / len(acc[k]):16.1f}  ({len(acc[k])} launches)")
PY
cat $OUT/summary.txt; cat $OUT/icache_counters.txt | head -20
