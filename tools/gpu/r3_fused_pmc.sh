cd /root/repo
export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
i=$((i+1))
timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/fspmc_$i -o x -- python tools/feat_ab.py --reps 5 > gpurun_out/fspmc_$i.log 2>&1
python - $i <<'PY'
import csv, glob, sys, collections
f = glob.glob('gpurun_out/fspmc_%s/**/x_counter_collection.csv' % sys.argv[1], recursive=True)
if not f: print('no counter file'); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0][-30:]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, 'launches', len(next(iter(d.values()))))
PY
done
