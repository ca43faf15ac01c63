mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_r2a
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_r2a -- python3 /root/repo/bench.py --steps 60 --warmup 10 --no-extras > /root/repo/gpurun_out/prof_r2a.log 2>&1
cd /root/repo
python tools/per_step.py gpurun_out/prof_r2a first_kernel > gpurun_out/prof_r2a_per_step.txt 2>&1
cat gpurun_out/prof_r2a_per_step.txt
ls gpurun_out/prof_r2a/*/ | head
