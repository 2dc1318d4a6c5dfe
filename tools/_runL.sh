mkdir -p gpurun_out
(time python bench.py --gpus 2 --dist-backend gloo --steps 6 --warmup 2) > gpurun_out/r2l_n2.json 2> gpurun_out/r2l_n2.err
echo rc=$?; tail -c 1500 gpurun_out/r2l_n2.err; python3 - <<'PY'
import json
t=open('gpurun_out/r2l_n2.json').read()
l=[x for x in t.splitlines() if x.startswith('{"metric')]
if l:
    d=json.loads(l[-1]); print({k:d[k] for k in ('value','n_gpus','scaling','ms_per_step')}); print(d['config']); print(d.get('weak_leg')); print([k for k in d if k.endswith('_error')])
PY
(time python bench.py --gpus 2 --dist-backend gloo --config 4 --steps 4 --warmup 1 --no-extras) > gpurun_out/r2l_c4.json 2> gpurun_out/r2l_c4.err; echo rc=$?; tail -c 600 gpurun_out/r2l_c4.err; cut -c1-400 gpurun_out/r2l_c4.json
