# usage: bash tools/experiments/wave2_run.sh TAG [check]
T=gpurun_out/$1; mkdir -p $T
export W2=$PWD/gps_optimize_slam_amd/libgsf_wave2.so
if [ "$2" = check ]; then
GSF_LIBRARY=$W2 timeout -k 10 500 python tests/campaigns/wave2_check.py check 600 > $T/check.log 2>&1
echo "check rc=$?" >> $T/check.log
fi
GSF_LIBRARY=$PWD/gps_optimize_slam_amd/libgsf_wave2_timing.so python tools/chunk_timing.py 1000 1000 > $T/t.log 2>&1
for rep in 1 2; do
for lib in base w2; do
  if [ $lib = w2 ]; then export GSF_LIBRARY=$W2; else unset GSF_LIBRARY; fi
  for k in pipeline ekf; do
    timeout -k 10 120 python bench.py --workload c2 --kernel $k --no-extra --no-cpu-baseline --no-other-rows > $T/b_${lib}_${k}_$rep.json 2> $T/b_${lib}_${k}_$rep.err
  done
  timeout -k 10 120 python bench.py --workload c2 --poses 1000 --kernel ekf --no-extra --no-cpu-baseline --no-other-rows > $T/b_${lib}_ekf1000_$rep.json 2> $T/b_${lib}_ekf1000_$rep.err
done
done
echo done
