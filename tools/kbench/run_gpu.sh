mkdir -p gpurun_out
{
for S in 1 2 4; do ./tools/kbench/kbench bwd 1024 16 1856 256 $S; done
./tools/kbench/kbench bwd 4096 16 1856 256 0
./tools/kbench/kbench bwd 512 32 3712 768 0
} > gpurun_out/kbench.log 2>&1
cat gpurun_out/kbench.log
if grep -q "Memory access fault" gpurun_out/kbench.log; then exit 1; fi
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "forward_backward" > gpurun_out/t_kern.log 2>&1; rc=$?; tail -8 gpurun_out/t_kern.log; exit $rc
