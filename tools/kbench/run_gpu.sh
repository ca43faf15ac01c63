mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; rc=$?; tail -15 gpurun_out/t_all.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')})
print(d['cpu_baseline'])
print(d['extra'])
print(d['roofline']['step']['whole_step'])
PY
