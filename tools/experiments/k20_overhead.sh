# the driver's command (K = 20 steps): step time against kernel time, graph replay vs eager launches
mkdir -p gpurun_out/r4ai
for rep in 1 2 3; do
for mode in graph eager; do
  fl=""; [ $mode = eager ] && fl="--no-graph"
  timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline $fl 2>gpurun_out/r4ai/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', 'ms_per_step', round(d['ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_ms']*1e3,2), 'us; overhead per region', round((d['ms_per_step']-d['roofline']['kernel_ms'])*20*1e3,1), 'us')"
done; done
