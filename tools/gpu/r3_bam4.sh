cd /root/repo
for pt in 1 2; do
CTO_DEVICE_PILEUP=1 timeout 1500 python tools/e2e_bench.py --kind bam --chunks 64 --sites 4096 --producers 12,16,20,24,32 --writers 2 --pipeline native --repeats 3 --inflate-cus 144 --inflate-jobs 8,12 --pack-threads $pt 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_bam_sweep3_pt$pt.log
done
python - <<'PY'
import json
for pt in (1, 2):
    for ln in open('gpurun_out/r3_bam_sweep3_pt%d.log' % pt):
        if ln.startswith('{"sites_per_s"'):
            d = json.loads(ln)
            print("pack_threads", pt, "producers", d["producers"], "jobs", d["inflate_jobs"], d["sites_per_s"], "dev", d["device_inflated"], "cpu", d["host_process"]["user_cpu_ms_per_chunk"])
PY
