mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > gpurun_out/final_tests.log
bash tools/profile_round.sh r03 > gpurun_out/profile_round.log 2>&1
cat gpurun_out/final_tests.log
tail -3 gpurun_out/profile_round.log | cut -c1-200
