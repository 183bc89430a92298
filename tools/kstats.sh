#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of `tools/ab.py --child` with the library given in CTO_LIB_PATH
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ab
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python $GRAFT_REPO_ROOT/tools/ab.py --child --reps ${1:-20} > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_ab/**/*kernel_stats.csv', recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'cto::' in n or 'anonymous' in n:
        short = n.replace('(anonymous namespace)::', '').split('(')[0][-60:]
        print('%-62s calls %4s  avg %9.1f us' % (short, r['Calls'], float(r['AverageNs']) / 1e3))
        if 'k_cvt_block' in n: tot += float(r['AverageNs']) / 1e3
print('sum of k_cvt_block averages: %.1f us' % tot)
PY
