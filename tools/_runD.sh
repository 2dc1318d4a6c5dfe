mkdir -p gpurun_out
V=$PWD/build/variants
(time python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_multirank_gpu.py -m gpu -q --timeout 900) > gpurun_out/r2d_pytest.log 2>&1
for v in i512 i1024; do (TSDF_HIP_LIB=$V/libtsdf_hip_$v.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q --timeout 900 -k "integrate or fuzz or checkpoint") > gpurun_out/r2d_pytest_$v.log 2>&1; done
run() {
  env TSDF_HIP_LIB=$V/libtsdf_hip_$2.so $3 python tools/bench_kernels.py --frames 12 --passes 20 --no-track-timing > gpurun_out/r2d_k_$1.json 2> gpurun_out/r2d_k_$1.err
  env TSDF_HIP_LIB=$V/libtsdf_hip_$2.so $3 python3 tools/pmc_memside.py gpurun_out/ms r2d_$1 --quick > gpurun_out/r2d_ms_$1.log 2>&1
}
run i256 i256 TSDF_X=0
run i512 i512 TSDF_X=0
run i1024 i1024 TSDF_X=0
python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r2d_bench.json 2>&1
tail -4 gpurun_out/r2d_pytest.log; tail -3 gpurun_out/r2d_pytest_i512.log; tail -3 gpurun_out/r2d_pytest_i1024.log
for n in i256 i512 i1024; do echo $n; cat gpurun_out/r2d_k_$n.json | cut -c1-200; grep integrate_kernel gpurun_out/r2d_ms_$n.log | cut -c1-400; done
tail -c 600 gpurun_out/r2d_bench.json
