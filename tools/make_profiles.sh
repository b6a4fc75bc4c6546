#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: rocprofv3 evidence for the bench's kernels.
#   1. kernel trace + stats of the DEFAULT bench command (python3 bench.py)
#   2. PMC passes (separate runs: gpurun forbids mixing --pmc with trace domains) for HBM traffic and issue mix
# Outputs under gpurun_out/profiles_raw/; tools/collect_profiles.py turns them into the committed profiles/*.
# usage: tools/make_profiles.sh            every pass;
#        tools/make_profiles.sh trace      only pass 1 again -- run it AFTER tools/collect_profiles.py has written profiles/rNN_traffic.json
#                                          from the PMC passes, so that the committed bench line quotes that traffic / issue profile
#        tools/make_profiles.sh main,ab    only the named sections (a gpurun call is capped at 20 minutes): trace main ab aux c3ekf r4;
#                                          raw outputs of earlier calls are kept (gpurun merges gpurun_out/ back)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_raw
SECTIONS=${1:-all}
want() { [ "$SECTIONS" = "all" ] || [[ ",$SECTIONS," == *",$1,"* ]]; }
if [ "$SECTIONS" = "all" ]; then rm -rf $OUT; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if want trace; then
rm -rf $OUT/trace_bench_default; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench_default -- python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "trace failed"
fi
if [ "$SECTIONS" = "trace" ]; then ls $OUT/trace_bench_default; exit 0; fi
if want main; then
for wl in c2 c3; do
  for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
    n=$(echo $p | cut -d" " -f1)
    rm -rf $OUT/pmc_${wl}_$n; timeout -k 10 300 rocprofv3 --pmc $p --output-format csv -d $OUT/pmc_${wl}_$n -- python3 $R/bench.py --workload $wl --no-extra --no-cpu-baseline --no-other-rows --steps 3 --warmup 1 > $OUT/pmc_${wl}_$n.json 2> $OUT/pmc_${wl}_$n.err || echo "pmc $wl $n failed"
  done
done
fi
if want ab; then
# 2b. the C2 launch by the other kernels, same counters (A/B for DESIGN.md section 5): K4 alone by the wave kernel, and the
#     workgroup-per-trajectory kernel (fused pipeline and K4 alone)
for spec in "c2ekf:--kernel ekf" "c2block:--set-option block_kernel=1 --fit-rows all" "c2blockekf:--kernel ekf --set-option block_kernel=1"; do
  wl=${spec%%:*}; fl=${spec#*:}
  for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
    n=$(echo $p | cut -d" " -f1)
    rm -rf $OUT/pmc_${wl}_$n; timeout -k 10 300 rocprofv3 --pmc $p --output-format csv -d $OUT/pmc_${wl}_$n -- python3 $R/bench.py --workload c2 $fl --no-extra --no-cpu-baseline --no-other-rows --steps 3 --warmup 1 > $OUT/pmc_${wl}_$n.json 2> $OUT/pmc_${wl}_$n.err || echo "pmc $wl $n failed"
  done
done
rm -rf $OUT/trace_c2block; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2block -- python3 $R/bench.py --workload c2 --set-option block_kernel=1 --fit-rows all --no-extra --no-cpu-baseline > $OUT/bench_c2block.json 2> $OUT/bench_c2block.err || echo "trace c2block failed"
ls $OUT
fi
if want aux; then
# 3. the auxiliary kernels (tools/bench_kernels.py): kernel trace + PMC passes
rm -rf $OUT/trace_aux; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_aux -- python3 $R/tools/bench_kernels.py > $OUT/aux_kernels.json 2> $OUT/aux_kernels.err || echo "aux trace failed"
for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
  n=$(echo $p | cut -d" " -f1)
  rm -rf $OUT/pmc_aux_$n; timeout -k 10 600 rocprofv3 --pmc $p --output-format csv -d $OUT/pmc_aux_$n -- python3 $R/tools/bench_kernels.py > $OUT/pmc_aux_$n.json 2> $OUT/pmc_aux_$n.err || echo "pmc aux $n failed"
done
fi
if want c3ekf; then
# 4. K4-only PMC at C3 (the committed C3 PMC of round 1 was of the pipeline kernel only)
for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
  n=$(echo $p | cut -d" " -f1)
  rm -rf $OUT/pmc_c3ekf_$n; timeout -k 10 300 rocprofv3 --pmc $p --output-format csv -d $OUT/pmc_c3ekf_$n -- python3 $R/bench.py --workload c3 --kernel ekf --no-extra --no-cpu-baseline --no-other-rows --steps 3 --warmup 1 > $OUT/pmc_c3ekf_$n.json 2> $OUT/pmc_c3ekf_$n.err || echo "pmc c3ekf $n failed"
done
ls $OUT
fi
if want r4; then
# 5. round 4: the kernels behind the C5 figure (one chunk of the shard: 32 768 tracks x 1 000 poses, grid 2 097 152), the lane-per-trajectory
#    kernels (time-major C3), and the headline kernels under the other row rule (--fit-rows all)
PMC4=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS")
for spec in "c5chunk:--workload c3 --traj-per-gpu 32768" "c3lane:--workload c3 --layout time" "c3laneekf:--workload c3 --layout time --kernel ekf" "c2all:--workload c2 --fit-rows all" "c3all:--workload c3 --fit-rows all"; do
  wl=${spec%%:*}; fl=${spec#*:}
  for p in "${PMC4[@]}"; do
    n=$(echo $p | cut -d" " -f1)
    rm -rf $OUT/pmc_${wl}_$n; timeout -k 10 300 rocprofv3 --pmc $p --output-format csv -d $OUT/pmc_${wl}_$n -- python3 $R/bench.py $fl --no-extra --no-cpu-baseline --no-other-rows --steps 3 --warmup 1 > $OUT/pmc_${wl}_$n.json 2> $OUT/pmc_${wl}_$n.err || echo "pmc $wl $n failed"
  done
done
ls $OUT
fi
