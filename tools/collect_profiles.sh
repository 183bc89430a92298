#!/bin/bash
# Collect the rocprofv3 evidence that profiles/ holds, on the GPU box:  bash tools/collect_profiles.sh <tag>
#   1. kernel stats of the exact default bench command (+ its JSON line)
#   2. kernel trace of a short run (one step's launch timeline)
#   3. HBM traffic per kernel: separate --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with other trace domains)
# Everything lands in gpurun_out/profiles_<tag>/ ; tools/profiles_digest.py turns it into the files committed under profiles/.
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -- python $R/bench.py > $OUT/default_bench.json 2> $OUT/default_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/short -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-sustained --no-postfilters --no-live-traffic > $OUT/short_bench.json 2> $OUT/short_bench.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-sustained --no-split --no-postfilters --no-live-traffic > /dev/null 2> $OUT/pmc_$c.err
done
#   4. matrix-pipe occupancy: MFMA instruction count and wave cycles per kernel (two more PMC-only passes)
for c in "SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-sustained --no-split --no-postfilters --no-live-traffic > /dev/null 2> $OUT/pmc_$n.err
done
python $R/tools/profiles_digest.py $OUT $TAG
#   5. the realigner's and the tokeniser's launches (their own small drivers)
ROUND=${CTO_ROUND:-round5}
mkdir -p $OUT/digest
CTO_REALIGN_TRACE=1 bash $R/tools/realign_kstats.sh 3 > $OUT/digest/${ROUND}_${TAG}_realign_kernels.txt 2>&1
grep '^\[realign\]' /tmp/prof_rl.log | tail -22 >> $OUT/digest/${ROUND}_${TAG}_realign_kernels.txt
bash $R/tools/tokenise_kstats.sh 6 > $OUT/digest/${ROUND}_${TAG}_tokenise_kernels.txt 2>&1
ls $OUT
