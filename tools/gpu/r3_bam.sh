cd /root/repo
for dp in 1 0; do
  CTO_DEVICE_PILEUP=$dp timeout 900 python tools/e2e_bench.py --kind bam --chunks 48 --sites 4096 --producers 16 --writers 2 --pipeline native --repeats 3 --inflate-cus 144 --inflate-jobs 8,16 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_bam_dp$dp.log
done
CTO_DEVICE_PILEUP=1 CTO_PIPE_TIMING=1 timeout 600 python tools/e2e_bench.py --kind bam --chunks 16 --sites 4096 --producers 8 --writers 2 --pipeline native --repeats 1 --inflate-cus 144 --inflate-jobs 8 2>&1 | grep "device pile-up" | head -12 > gpurun_out/r3_bam_timing.log
tail -n 4 gpurun_out/r3_bam_dp1.log gpurun_out/r3_bam_dp0.log; cat gpurun_out/r3_bam_timing.log
