mkdir -p gpurun_out
timeout 700 python3 tools/pmc_memside.py gpurun_out/ms r2g --passes=9,10,0,2,1,3 > gpurun_out/r2g_ms.log 2>&1
cat gpurun_out/r2g_ms.log | cut -c1-1200
