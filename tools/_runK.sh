mkdir -p gpurun_out/prof
(time python -m pytest tests -m gpu -q --timeout 900) > gpurun_out/r2k_pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2k_smoke.log 2>&1
timeout 900 python3 tools/collect_profiles.py gpurun_out/prof r02 > gpurun_out/r2k_collect.log 2>&1
tail -5 gpurun_out/r2k_pytest.log; tail -2 gpurun_out/r2k_smoke.log; tail -15 gpurun_out/r2k_collect.log | cut -c1-600
