cd /root/repo
CTO_DEVICE_PILEUP=1 timeout 1500 python tools/e2e_bench.py --kind bam --chunks 64 --sites 4096 --producers 16 --writers 2 --pipeline native --repeats 3 --inflate-cus 64,96,112,128 --inflate-jobs 10,12,14,16 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_bam_sweep2.log
python - <<'PY'
import json
for ln in open('gpurun_out/r3_bam_sweep2.log'):
    if ln.startswith('{"sites_per_s"'):
        d = json.loads(ln)
        print(d["inflate_cus"], d["inflate_jobs"], d["sites_per_s"], d["device_inflated"], d["host_process"]["user_cpu_ms_per_chunk"], d["stage_thread_time"]["device_ms_per_chunk"])
PY
