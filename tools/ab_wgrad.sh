#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 -m pytest $R/tests/test_gpu_model.py -x -q -m gpu -k "riders or graph_replayed or philox_training" > $R/gpurun_out/t_model.log 2>&1 || { tail -40 $R/gpurun_out/t_model.log; exit 1; }
tail -2 $R/gpurun_out/t_model.log
python3 $R/bench.py --steps 200 --warmup 20 --no-extras > $R/gpurun_out/ab_wg.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_wg_prof -o run -- python3 $R/bench.py --steps 60 --warmup 10 --no-extras > $R/gpurun_out/ab_wg_prof.log 2>&1
python3 $R/tools/per_step.py $R/gpurun_out/ab_wg_prof "first_" > $R/gpurun_out/ab_wg.txt
rm -f $R/gpurun_out/ab_wg_prof/*kernel_trace.csv
grep -h ms_per_step $R/gpurun_out/ab_wg.log | python3 -c "import sys,json; [print(json.loads(l)['ms_per_step'], json.loads(l)['config']['final_loss']) for l in sys.stdin]"
cat $R/gpurun_out/ab_wg.txt | cut -c1-130
