for w in 2 3 4 6; do timeout 300 python -m clairs_to_amd.e2e --kinds text --chunks 96 --writers $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['mpileup_text_to_vcf']; print(r['sites_per_s'], r['producers'], r['writers'], r['stage_thread_time'])"; done
