mkdir -p gpurun_out/kt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2j_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2j_bench.err
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/kt/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r2j_kernel_stats.csv; head -12 $f | cut -c1-200
tail -c 400 gpurun_out/r2j_bench.json
