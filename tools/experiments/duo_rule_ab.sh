set -u
mkdir -p gpurun_out/r4ag
for rep in 1 2; do
for T in ${DUO_SIZES:-1000 768 512 384}; do
for duo in 0 1; do
  timeout -k 10 120 python bench.py --workload c2 --traj-per-gpu $T --set-option duo_kernel=$duo --no-extra --no-cpu-baseline --no-other-rows > gpurun_out/r4ag/d${duo}_${T}_${rep}.json 2> gpurun_out/r4ag/d${duo}_${T}_${rep}.err
  python -c "
import json
d=json.loads(open('gpurun_out/r4ag/d${duo}_${T}_${rep}.json').read().strip().splitlines()[-1]); r=d['roofline']
print('T=$T duo=$duo', round(r['kernel_ms']*1e3,2),'us', r['kernel'][:40], d.get('status_bits_equal'))"
done; done; done
