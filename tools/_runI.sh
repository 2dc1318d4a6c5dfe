mkdir -p gpurun_out
V=$PWD/build/variants
for d in 0 1 8 16; do
  TSDF_HIP_LIB=$V/libtsdf_hip_dbg.so TSDF_DEBUG_INTEGRATE=$d python tools/bench_kernels.py --frames 12 --passes 4 --no-track-timing > gpurun_out/r2i_k_$d.json 2> gpurun_out/r2i_k_$d.err
  echo debug=$d; cut -c1-100 gpurun_out/r2i_k_$d.json
done
