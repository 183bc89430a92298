#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of tools/tokenise_gpu_bench.py: the device tokeniser on one 4096-site chunk
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tok
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tok -- python $GRAFT_REPO_ROOT/tools/tokenise_gpu_bench.py --reps ${1:-6} > /tmp/prof_tok.log 2>&1
grep -E "device:|host:" /tmp/prof_tok.log | tail -4
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_tok/**/*kernel_stats.csv', recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'k_' in n and 'rocclr' not in n:
        short = n.replace('(anonymous namespace)::', '').split('(')[0][-48:]
        print('%-50s calls %4s  avg %9.1f us' % (short, r['Calls'], float(r['AverageNs']) / 1e3))
PY
