#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_convblock.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/t_cb.log 2>&1; echo rc=$?; tail -2 gpurun_out/t_cb.log
for W in cfg5 cfg2; do
for V in tiled stream; do
  EMB_CONVT_IMPL=$V timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-extras --workload $W 2>/dev/null | python3 -c "import sys,json; [print('$W', '$V', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
done; done
