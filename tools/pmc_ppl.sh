R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_ppl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ppl in 1 2; do
  for p in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
    n=$(echo $p | cut -d" " -f1)
    timeout -k 10 300 rocprofv3 --pmc $p --output-format csv -d $OUT/ppl${ppl}_$n -- python3 $R/bench.py --workload c3 --kernel ekf --no-extra --no-cpu-baseline --steps 3 --warmup 1 --set-option wave_ppl=$ppl > $OUT/ppl${ppl}_$n.json 2> $OUT/ppl${ppl}_$n.err || echo "pmc $ppl $n failed"
  done
done
ls $OUT
