cd /root/repo
export TMPDIR=/tmp


timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ops.py -m gpu -x -q --timeout 300 2>&1 | tail -4
for lib in clairs_to_amd/libclairsto_amd.so clairs_to_amd/libfs_*.so; do
[ -f $lib ] || continue
tag=$(basename $lib .so)
CTO_LIB_PATH=$PWD/$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fsprof3_$tag -o x -- python tools/feat_ab.py > gpurun_out/fsprof3.log 2>&1
python - $tag <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/fsprof3_%s/**/x_kernel_trace.csv' % sys.argv[1], recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_featurize_sites' in r['Kernel_Name']]
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
print(sys.argv[1], len(d), 'with stores', sum(d[5:55]) / 50, 'without', sum(d[60:110]) / 50)
PY
done
