mkdir -p gpurun_out
(time python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sequence.py -m gpu -q --timeout 900) > gpurun_out/r2c_pytest.log 2>&1
V=$PWD/build/variants
run() {  # name lib env
  env TSDF_HIP_LIB=$V/libtsdf_hip_$2.so $3 python tools/bench_kernels.py --frames 12 --passes 20 --no-track-timing > gpurun_out/r2c_k_$1.json 2> gpurun_out/r2c_k_$1.err
  env TSDF_HIP_LIB=$V/libtsdf_hip_$2.so $3 python3 tools/pmc_memside.py gpurun_out/ms r2c_$1 --quick > gpurun_out/r2c_ms_$1.log 2>&1
}
run base base TSDF_X=0
run ntdw ntdw TSDF_X=0
run norec dbg TSDF_DEBUG_INTEGRATE=1
run normw dbg TSDF_DEBUG_INTEGRATE=2
tail -6 gpurun_out/r2c_pytest.log
for n in base ntdw norec normw; do echo $n; cat gpurun_out/r2c_k_$n.json | cut -c1-330; grep integrate_kernel gpurun_out/r2c_ms_$n.log | cut -c1-600; done
