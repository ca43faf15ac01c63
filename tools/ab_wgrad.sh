#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 -m pytest $R/tests/test_gpu_convblock.py -x -q -m gpu > $R/gpurun_out/t_wg.log 2>&1 || { tail -30 $R/gpurun_out/t_wg.log; exit 1; }
tail -2 $R/gpurun_out/t_wg.log
python3 $R/bench.py --steps 200 --warmup 20 --no-extras > $R/gpurun_out/ab_wg.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_wg_prof -o run -- python3 $R/bench.py --steps 60 --warmup 10 --no-extras > $R/gpurun_out/ab_wg_prof.log 2>&1
python3 $R/tools/per_step.py $R/gpurun_out/ab_wg_prof "first_kernel<2, 2, 0>" > $R/gpurun_out/ab_wg.txt
rm -f $R/gpurun_out/ab_wg_prof/*kernel_trace.csv
grep -h ms_per_step $R/gpurun_out/ab_wg.log | python3 -c "import sys,json; [print(json.loads(l)['ms_per_step'], json.loads(l)['config']['final_loss']) for l in sys.stdin]"
head -10 $R/gpurun_out/ab_wg.txt; tail -2 $R/gpurun_out/ab_wg.txt
