#!/bin/bash
# Developer tool: end-of-round validation on the GPU box (full GPU suite, smoke, two-rank rehearsal on one GPU, evidence profile)
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/t_all.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 50 --warmup 10 --backend gloo --no-extras > gpurun_out/bench_gloo2.log 2>&1; echo "gloo2 rc=$?"; grep -h ms_per_step gpurun_out/bench_gloo2.log | python3 -c "import sys,json; [print('gloo2', json.loads(l)['ms_per_step'], json.loads(l)['value']) for l in sys.stdin]"
bash tools/round_profile.sh round2_l | tail -c 300
