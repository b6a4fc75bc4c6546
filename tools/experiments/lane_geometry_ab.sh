mkdir -p gpurun_out/r4ad
for lib in base b256_pf2_o2 b256_pf4_o2 b64_pf4_o2 b256_pf2_o3 b1024_pf2_o2; do
  if [ $lib = base ]; then unset GSF_LIBRARY; else export GSF_LIBRARY=$PWD/gps_optimize_slam_amd/libgsf_lane_$lib.so; fi
  for k in ekf pipeline; do
    timeout -k 10 200 python bench.py --workload c3 --layout time --kernel $k --no-extra --no-cpu-baseline --no-other-rows --steps 5 --warmup 2 > gpurun_out/r4ad/$lib.$k.json 2> gpurun_out/r4ad/$lib.$k.err
    python -c "
import json,sys
d=json.loads(open('gpurun_out/r4ad/$lib.$k.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$lib $k', round(r['kernel_ms'],3),'ms', r['kernel'], d.get('max_abs_pos_err_m'), d.get('status_bits_equal'))"
  done
done
