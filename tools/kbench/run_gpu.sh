mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; rc=$?; tail -15 gpurun_out/t_all.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.log | cut -c1-1500
