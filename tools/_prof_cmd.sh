cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_f -- python $GRAFT_REPO_ROOT/tools/kbench.py --reps 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT/gpurun_out/prof_f && find . -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} ../kstats_f.csv
