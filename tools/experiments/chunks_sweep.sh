# length of the text leg's run against its rate: fill and drain of the pipeline are a fixed ~12 ms
for c in 96 192 384; do timeout 500 python -m clairs_to_amd.e2e --kinds text --chunks $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['mpileup_text_to_vcf']; print($c, r['sites_per_s'], r['seconds'], r['producers'], r['device_tokeniser']['sites_per_s'])"; done
