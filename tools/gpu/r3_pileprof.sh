cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pileprof -- python $R/tools/pileup_bench.py 2000 6 > $R/gpurun_out/r3_pileprof.log 2>&1
cat $R/gpurun_out/r3_pileprof.log | grep -v amdgpu.ids | tail -10
f=$(find $R/gpurun_out/pileprof -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r3_pileprof_kernel_stats.csv
head -25 $f
