R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_lds
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/prof_lds -- python $R/tools/ab.py --child --reps 5 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/prof_lds/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'cto::' in k or 'featurize' in k:
        m = {c: sum(v) / len(v) for c, v in d.items()}
        print(k.ljust(70), ' '.join('%s=%.3g' % (c.replace('SQ_', ''), v) for c, v in sorted(m.items())), 'conflict/active=%.2f' % (m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
PY
