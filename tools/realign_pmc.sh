#!/bin/bash
# hardware counters of the realigner's kernels (PMC-only passes of tools/realign_gpu_bench.py; never combined with other trace domains)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/prof_rpmc
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/prof_rpmc -- python $R/tools/realign_gpu_bench.py --reps 1 > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
fs = glob.glob('/tmp/prof_rpmc/**/*counter_collection.csv', recursive=True)
if not fs:
    print('no counters'); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    n = r['Kernel_Name']
    if 'k_sw' in n or 'k_banded' in n or 'k_fast_pass' in n:
        key = n.replace('(anonymous namespace)::', '').split('(')[0][:28] + ' grid ' + r.get('Grid_Size', r.get('Grid_Size_X', '?'))
        acc[key][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in acc.items():
    print(k.ljust(44), ' '.join('%s=%.4g' % (c.replace('SQ_', ''), v) for c, v in sorted(d.items())))
PY
done
