mkdir -p gpurun_out
(python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_golden.py -m gpu -q --timeout 900 -x) > gpurun_out/r2h_pytest.log 2>&1
python tools/bench_kernels.py --frames 12 --passes 20 --no-track-timing > gpurun_out/r2h_k.json 2> gpurun_out/r2h_k.err
tail -3 gpurun_out/r2h_pytest.log; cut -c1-200 gpurun_out/r2h_k.json
