# CvT: grouped-block launches vs one launch per block; parity tests of the networks
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "cvt or model or ops or pickles or invariance or platform_configs" > gpurun_out/r3_cvt_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_cvt_tests.log
tail -4 gpurun_out/r3_cvt_tests.log
for g in 4 1 4 1; do
  echo "CTO_CVT_BLOCKS_PER_LAUNCH=$g"
  CTO_CVT_BLOCKS_PER_LAUNCH=$g timeout 300 python tools/ab.py clairs_to_amd/libclairsto_amd.so --reps 60 2>&1 | tail -2
done > gpurun_out/r3_cvt_ab.log 2>&1
cat gpurun_out/r3_cvt_ab.log
