# same-box A/B at the headline shape: GSF_LIBRARY_A against the shipped libgsf.so (fused pipeline and K4, 1 000 x 271, graph replay)
mkdir -p gpurun_out/r4aw
for rep in 1 2 3; do
for lib in A shipped; do
  if [ $lib = A ]; then export GSF_LIBRARY=$GSF_LIBRARY_A; else unset GSF_LIBRARY; fi
  for k in pipeline ekf; do
  timeout -k 10 120 python bench.py --workload c2 --kernel $k --no-extra --no-cpu-baseline --no-other-rows > gpurun_out/r4aw/${lib}_${k}_$rep.json 2> gpurun_out/r4aw/${lib}_${k}_$rep.err
  python -c "
import json
d=json.loads(open('gpurun_out/r4aw/${lib}_${k}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$lib $k', round(r['kernel_ms']*1e3,2),'us', d.get('max_abs_pos_err_m'), d.get('status_bits_equal'))"
  done
done; done
