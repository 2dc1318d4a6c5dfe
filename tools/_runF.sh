mkdir -p gpurun_out
(time python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -x) > gpurun_out/r2f_pytest.log 2>&1
python tools/bench_kernels.py --frames 12 --passes 20 --no-track-timing > gpurun_out/r2f_k.json 2> gpurun_out/r2f_k.err
python3 tools/pmc_memside.py gpurun_out/ms r2f --quick > gpurun_out/r2f_ms.log 2>&1
python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r2f_bench.json 2>&1
tail -4 gpurun_out/r2f_pytest.log; cut -c1-200 gpurun_out/r2f_k.json; grep "integrate_kernel\|clip_rows\|scatter" gpurun_out/r2f_ms.log | cut -c1-400; tail -c 900 gpurun_out/r2f_bench.json
