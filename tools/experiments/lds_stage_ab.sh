# same-box A/B of two builds of the library on the big-batch workloads (C3 pipeline / K4, one C5 chunk): GSF_LIBRARY_A vs the shipped libgsf.so
mkdir -p gpurun_out/r4am
for rep in 1 2; do
for lib in A shipped; do
  if [ $lib = A ]; then export GSF_LIBRARY=$GSF_LIBRARY_A; else unset GSF_LIBRARY; fi
  for spec in "c3pipe:--workload c3" "c3ekf:--workload c3 --kernel ekf" "c5chunk:--workload c3 --traj-per-gpu 32768"; do
    wl=${spec%%:*}; fl=${spec#*:}
    timeout -k 10 200 python bench.py $fl --no-extra --no-cpu-baseline --no-other-rows > gpurun_out/r4am/${lib}_${wl}_$rep.json 2> gpurun_out/r4am/${lib}_${wl}_$rep.err
    python -c "
import json
d=json.loads(open('gpurun_out/r4am/${lib}_${wl}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$lib $wl', round(r['kernel_ms'],4),'ms', r['kernel'][:36], d.get('max_abs_pos_err_m'), d.get('status_bits_equal'))"
  done
done; done
