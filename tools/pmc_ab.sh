#!/bin/bash
# PMC A/B of two library builds on the C2 kernels (run through gpurun from the repo root): tools/pmc_ab.sh <libA.so> <libB.so>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_ab
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  export GSF_LIBRARY=$R/gps_optimize_slam_amd/$lib
  for kern in ekf pipeline; do
    timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/${tag}_$kern -- python3 $R/bench.py --kernel $kern --no-extra --no-cpu-baseline --steps 3 --warmup 1 --set-option duo_kernel=0 > $OUT/${tag}_$kern.json 2> $OUT/${tag}_$kern.err || echo "pmc $tag $kern failed"
    echo "#### $tag $kern"; python3 $R/tools/pmc_summary.py $OUT/${tag}_$kern | grep -A9 "ekf_wave_kernel"
  done
done
