mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; rc=$?; tail -6 gpurun_out/t_all.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --no-extras > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')})
PY
R=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/prof_r2b
rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_r2b -- python3 /root/repo/bench.py --steps 60 --warmup 10 --no-extras > $R/prof_r2b.log 2>&1
cd /root/repo
python tools/per_step.py gpurun_out/prof_r2b "first_kernel<2, 2, 0>" > gpurun_out/prof_r2b_per_step.txt 2>&1; cat gpurun_out/prof_r2b_per_step.txt
./tools/kbench/kbench fwd 1024 16 1856 256 > gpurun_out/kbench_fwd.log 2>&1; cat gpurun_out/kbench_fwd.log
./tools/kbench/kbench fwd 512 32 3712 768 >> gpurun_out/kbench_fwd.log 2>&1; tail -3 gpurun_out/kbench_fwd.log
