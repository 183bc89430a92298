#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of the two BAM file-to-file legs (clairs_to_amd.e2e --kinds bam): which kernels a BAM chunk's
# device time is made of.  Durations are wall durations of launches that OVERLAP (8-10 chunks in flight, the inflate on CU-masked streams): they rank
# the kernels, they do not add up to chip time.   bash tools/bam_kstats.sh   (on the GPU box, PYTHONPATH = the repo)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bam
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bam -- python -m clairs_to_amd.e2e --kinds bam --chunks 36 --bam-chunks 12 --rank-share 0 > /tmp/prof_bam.log 2>&1
tail -c 600 /tmp/prof_bam.log
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_bam/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][-50:]
    print('%-52s calls %6s  avg %9.1f us  total %8.1f ms  %5.1f %%' % (n, r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot))
PY
