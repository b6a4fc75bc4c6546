#!/bin/bash
# is the pipeline's second read of its fit rows (57 B/pose on top of K4's 145) paid at the HBM rate?
# Same 1e8 poses per launch, tracks of 1 000 / 2 000 / 4 000 poses (rows in flight of the resident waves: 3 072 tracks x N x 89 B = 273 MB / 547 MB /
# 1.09 GB against 256 MB of Infinity Cache), fused pipeline vs K4 alone, same box, graph replay.  Output: one line per run under gpurun_out/$1/.
set -u
OUT=gpurun_out/${1:-r5fit}
mkdir -p $OUT
for rep in 1 2; do
for spec in "1000:100000" "2000:50000" "4000:25000"; do
  N=${spec%%:*}; T=${spec#*:}
  for k in pipeline ekf; do
    timeout -k 10 200 python bench.py --workload c3 --poses $N --traj-per-gpu $T --kernel $k --steps 10 --warmup 2 --no-extra --no-cpu-baseline --no-other-rows \
      > $OUT/${k}_${N}_${rep}.json 2> $OUT/${k}_${N}_${rep}.err || echo "run $k $N failed"
    python - <<PY
import json
d=json.loads(open('$OUT/${k}_${N}_${rep}.json').read().strip().splitlines()[-1]); r=d['roofline']
print('rep $rep N=$N T=$T $k: kernel_ms %.4f  frac %.3f  %s' % (r['kernel_ms'], r['frac'], r['kernel']))
PY
  done
done
done
