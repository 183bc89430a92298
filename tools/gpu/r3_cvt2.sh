cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 -k "cvt or model or ops or pickles or invariance or platform_configs" > gpurun_out/r3_cvt_tests2.log 2>&1; echo "rc=$?" >> gpurun_out/r3_cvt_tests2.log
tail -4 gpurun_out/r3_cvt_tests2.log
timeout 600 python tools/ab.py clairs_to_amd/libclairsto_amd.so clairs_to_amd/libnochain.so clairs_to_amd/libclairsto_amd.so clairs_to_amd/libnochain.so --reps 60 2>&1 | tail -5 > gpurun_out/r3_cvt_ab2.log
cat gpurun_out/r3_cvt_ab2.log
