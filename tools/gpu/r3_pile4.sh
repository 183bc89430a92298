cd /root/repo
timeout 900 python -m pytest tests/test_gpu_pileup.py -x -q --timeout 300 > gpurun_out/r3_pile4.log 2>&1; echo "rc=$?" >> gpurun_out/r3_pile4.log
tail -30 gpurun_out/r3_pile4.log
