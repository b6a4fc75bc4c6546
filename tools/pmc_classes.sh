#!/bin/bash
# Run on the GPU box from the repo root: dynamic instruction counts of ONE 64-pose chunk of the wave kernel, by class.
# Two track lengths 64 poses apart (256 and 320: four and five full chunks) differ by exactly one chunk; the counters of the two
# launches are subtracted by tools/price_mix.py.  Counter passes only (no trace domains in the same run).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_classes
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/list_avail.txt 2>&1 || true
PASSES=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
        "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"
        "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY")
for k in ekf pipeline; do
  for n in 256 320; do
    i=0
    for p in "${PASSES[@]}"; do
      timeout -k 10 300 rocprofv3 --pmc $p --output-format csv -d $OUT/${k}_n${n}_p$i -- python3 $R/bench.py --workload c2 --poses $n --kernel $k --no-extra --no-cpu-baseline --no-other-rows --steps 3 --warmup 1 > $OUT/${k}_n${n}_p$i.json 2> $OUT/${k}_n${n}_p$i.err || echo "pass $k $n $i failed"
      i=$((i+1))
    done
  done
done
ls $OUT
