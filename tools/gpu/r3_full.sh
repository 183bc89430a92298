cd /root/repo
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 400 > gpurun_out/r3_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_gpu.log
tail -6 gpurun_out/r3_pytest_gpu.log
bash tools/collect_profiles.sh round3_b > gpurun_out/r3_collect.log 2>&1
tail -5 gpurun_out/r3_collect.log
ls gpurun_out/profiles_round3_b | head -30
