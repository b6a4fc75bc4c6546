mkdir -p gpurun_out/r4ao
for rep in 1 2; do
for lib in shipped enuocc6 enuocc8; do
  if [ $lib = shipped ]; then unset GSF_LIBRARY; else export GSF_LIBRARY=$PWD/gps_optimize_slam_amd/libgsf_$lib.so; fi
  python tools/ab_k1.py $lib 2>/dev/null | grep ENU
done
for lib in shipped k3occ7 k3occ8; do
  if [ $lib = shipped ]; then unset GSF_LIBRARY; else export GSF_LIBRARY=$PWD/gps_optimize_slam_amd/libgsf_$lib.so; fi
  echo "$lib: $(python tools/ab_k3.py 2>/dev/null | grep -i "1e8\|100000\|ms" | head -3 | tr '\n' ' ')"
done
done
