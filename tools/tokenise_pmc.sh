#!/bin/bash
# hardware counters of the tokeniser's kernels (PMC-only passes of tools/tokenise_gpu_bench.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_BRANCH"; do
  rm -rf /tmp/prof_tpmc
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/prof_tpmc -- python $R/tools/tokenise_gpu_bench.py --reps 2 > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
fs = glob.glob('/tmp/prof_tpmc/**/*counter_collection.csv', recursive=True)
if not fs:
    print('no counters'); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    n = r['Kernel_Name']
    if 'k_rows' in n or 'k_expand' in n or 'k_row_keys' in n or 'k_apply3' in n:
        acc[n.replace('(anonymous namespace)::', '').split('(')[0][:28]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k.ljust(30), ' '.join('%s=%.4g' % (c.replace('SQ_', ''), sum(v) / len(v)) for c, v in sorted(d.items())))
PY
done
