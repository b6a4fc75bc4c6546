#!/bin/bash
# the auxiliary-kernel part of tools/make_profiles.sh alone (PMC passes of tools/bench_kernels.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_raw
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
  n=$(echo $p | cut -d" " -f1)
  timeout -k 10 600 rocprofv3 --pmc $p --output-format csv -d $OUT/pmc_aux_$n -- python3 $R/tools/bench_kernels.py > $OUT/pmc_aux_$n.json 2> $OUT/pmc_aux_$n.err || echo "pmc aux $n failed"
done
ls $OUT
