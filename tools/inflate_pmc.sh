#!/bin/bash
# PMC passes over the inflate kernel with 8 launches in flight (rocprofv3 --kernel-trace --pmc, counters only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
BAM=$(python $R/tools/inflate_ab.py --make-bam /tmp/infpmc_run | tail -1)
export CTO_AB_K=${CTO_AB_K:-8}
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES"; do
  rm -rf /tmp/prof_inf
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_inf -- python $R/tools/inflate_ab.py --child $BAM > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/prof_inf/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_bgzf_inflate' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('  '.join('%s=%.4g (n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(acc.items())))
PY
done
