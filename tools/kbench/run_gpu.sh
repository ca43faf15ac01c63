set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/t_kern.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_kern.log
tail -25 gpurun_out/t_kern.log
for wl in cfg2 cfg2_b4096 cfg5; do
  for S in 0 1 2 4; do
    EMB_BWD_S=$S timeout -k 10 300 python bench.py --roofline-only --workload $wl > gpurun_out/roofb_${wl}_S$S.log 2>&1
  done
done
for wl in cfg3 cfg4; do timeout -k 10 300 python bench.py --roofline-only --workload $wl > gpurun_out/roofb_${wl}_S0.log 2>&1; done
for f in gpurun_out/roofb_*.log; do echo $f; grep kernels $f | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l)['kernels']; print({k:round(v['us_per_launch'],2) for k,v in d.items()})
"; done
