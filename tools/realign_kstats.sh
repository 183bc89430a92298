#!/bin/bash
# per-launch durations of the realigner's kernels (rocprofv3 --kernel-trace) for tools/realign_gpu_bench.py
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_rl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rl -- python $GRAFT_REPO_ROOT/tools/realign_gpu_bench.py --reps ${1:-2} > /tmp/prof_rl.log 2>&1
grep '^{' /tmp/prof_rl.log | tail -1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_rl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[-6:]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '')
    print('%-40s grid %7s wg %4s lds %6s  %9.3f ms' % (n[:40], r['Grid_Size_X'], r['Workgroup_Size_X'], r['LDS_Block_Size'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
PY
