#!/bin/bash
# PMC passes over the split-operand layer-2 kernel (rocprofv3 --kernel-trace --pmc, counters only): bash tools/split_pmc.sh [f16|bf16]
R=${GRAFT_REPO_ROOT:-$(pwd)}
KIND=${1:-f16}
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/prof_split
  CTO_GRU_SPLIT=$KIND rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_split -- python $R/tools/ab.py --child --reps 5 > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
fs = glob.glob('/tmp/prof_split/**/*counter_collection.csv', recursive=True)
if not fs:
    print('no counter file'); raise SystemExit
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'k_gru_split<256' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('  '.join('%s=%.5g (n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(acc.items())))
PY
done
