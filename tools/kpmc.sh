#!/bin/bash
# per-kernel PMC sums (rocprofv3 --kernel-trace --pmc <counters>) of `tools/ab.py --child`; usage: tools/kpmc.sh "CTR1 CTR2 ..."
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pmc
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d /tmp/prof_pmc -- python $GRAFT_REPO_ROOT/tools/ab.py --child --reps 3 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/prof_pmc/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'k_cvt_block' in n or 'k_gru' in n:
        acc[n.replace('(anonymous namespace)::', '').split('(')[0][-48:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, '  '.join('%s=%.3g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
