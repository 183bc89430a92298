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
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = [i for i, r in enumerate(rows) if 'k_fast_pass' in r['Kernel_Name']][-1]          # the last call's launches, on one time axis
t0 = int(rows[last]['Start_Timestamp'])
for r in rows[last:]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '')
    print('%-34s grid %7s wg %4s lds %6s  start %8.3f  end %8.3f ms' % (n[:34], r['Grid_Size_X'], r['Workgroup_Size_X'], r['LDS_Block_Size'],
          (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6))
PY
