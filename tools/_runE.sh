mkdir -p gpurun_out
V=$PWD/build/variants
for v in dbg1 dbg2 dbg3; do
  TSDF_HIP_LIB=$V/libtsdf_hip_$v.so TSDF_DEBUG_INTEGRATE=1 python tools/bench_kernels.py --frames 12 --passes 4 --no-track-timing > gpurun_out/r2e_k_$v.json 2> gpurun_out/r2e_k_$v.err
  echo $v; cut -c1-120 gpurun_out/r2e_k_$v.json
done
