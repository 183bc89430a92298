cd /root/repo
timeout 600 python -m pytest tests/test_gpu_pileup.py -x -q --timeout 200 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q -k "native_pipeline" --timeout 300 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pileprof2 -- python $R/tools/pileup_bench.py 2000 6 > $R/gpurun_out/r3_pileprof2.log 2>&1
grep -v amdgpu.ids $R/gpurun_out/r3_pileprof2.log | grep "device:\|host reader" | tail -4
f=$(find $R/gpurun_out/pileprof2 -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r3_pileprof2_kernel_stats.csv
cut -d, -f1-4 $f | sed 's/(anonymous namespace):://g' | cut -c1-60,400- | head -14
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r3_pileprof2_kernel_stats.csv')))
for r in rows[:14]:
    print(r['Name'][:48].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
