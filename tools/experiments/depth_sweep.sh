# chunks in flight (slots) of the C pipeline against the text leg's rate: CTO_PIPELINE_DEPTH, 0 = producers + writers + 2
for d in 0 12 16 24; do CTO_PIPELINE_DEPTH=$d timeout 300 python -m clairs_to_amd.e2e --kinds text --chunks 96 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['mpileup_text_to_vcf']; print($d, r['sites_per_s'], r['producers'], r['writers'], r['stage_thread_time'])"; done
