mkdir -p gpurun_out
(python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_sequence.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900) > gpurun_out/r2m_pytest.log 2>&1
python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r2m_bench.json 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ktm -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
grep -E "passed|failed" gpurun_out/r2m_pytest.log | tail -2
grep "tsdf::" gpurun_out/ktm/*/*kernel_stats.csv | cut -d, -f1,2,4 | cut -c1-40,200-
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r2m_bench.json') if l.startswith('{\"metric')][-1])
print(d['value'], d['ms_per_step'], d['gn_iterations_per_frame'], d['stage_ms_per_frame'], d['tracker_gather']['avg_pass_wall_ms'])"
