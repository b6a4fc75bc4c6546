set -u
# same-box A/B: the hybrid build (helper wave only for outage tracks) against the shipped library, fused pipeline on 271-pose tracks
mkdir -p gpurun_out/r4as
HY=$PWD/gps_optimize_slam_amd/libgsf_hybrid.so
for rep in 1 2; do
for T in 1000; do
for lib in shipped hybrid2 hybrid1 hybrid3; do
  unset GSF_LIBRARY GSF_HYBRID_MODE
  if [ $lib = hybrid0 ]; then export GSF_LIBRARY=$HY GSF_HYBRID_MODE=0; fi
  if [ $lib = hybrid1 ]; then export GSF_LIBRARY=$HY GSF_HYBRID_MODE=1; fi
  if [ $lib = hybrid2 ]; then export GSF_LIBRARY=$HY GSF_HYBRID_MODE=2; fi
  if [ $lib = hybrid3 ]; then export GSF_LIBRARY=$HY GSF_HYBRID_MODE=3; fi
  timeout -k 10 120 python bench.py --workload c2 --traj-per-gpu $T --no-extra --no-cpu-baseline --no-other-rows > gpurun_out/r4as/${lib}_${T}_${rep}.json 2> gpurun_out/r4as/${lib}_${T}_${rep}.err
  python -c "
import json
d=json.loads(open('gpurun_out/r4as/${lib}_${T}_${rep}.json').read().strip().splitlines()[-1]); r=d['roofline']
print('T=$T $lib', round(r['kernel_ms']*1e3,2),'us', d.get('max_abs_pos_err_m'), d.get('status_bits_equal'))"
done; done; done
