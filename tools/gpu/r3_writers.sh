cd /root/repo
for w in 2 3 4; do for p in 4 6; do
echo "== writers $w producers $p"
timeout 600 python -m clairs_to_amd.e2e --chunks 96 --kinds text --writers $w --producers $p --reference-chunk-sites 10000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
for k in ('mpileup_text_to_vcf','mpileup_text_to_vcf_reference_chunks'):
    v=d[k]; print(k, v['sites_per_s'], v['stage_thread_time'])
"
done; done
