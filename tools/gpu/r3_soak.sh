cd /root/repo
timeout 1200 python tools/e2e_bench.py --kind bam --chunks 48 --sites 4096 --producers 16 --writers 2 --pipeline native --repeats 40 --inflate-cus 144 --inflate-jobs 8 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 > gpurun_out/r3_soak_bam.log
timeout 1200 python tools/e2e_bench.py --kind bam --chunks 48 --sites 4096 --producers 16 --writers 2 --pipeline native --repeats 4 --inflate-cus 144 --inflate-jobs 8 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('4 repeats: rss', d['max_rss_mb'], 'rate', d['sites_per_s'])" >> gpurun_out/r3_soak_bam.log
timeout 1200 python tools/e2e_bench.py --kind text --chunks 60 --sites 10000 --producers 8 --writers 2 --pipeline native --repeats 60 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 > gpurun_out/r3_soak_text.log
cat gpurun_out/r3_soak_bam.log gpurun_out/r3_soak_text.log
