#!/bin/bash
# CvT block kernels: the product build beside the ablation builds (CTO_CVT_ABL=1: no GEMMs; =2: no LayerNorm / depth-wise / softmax / GELU)
# and the half-height-tile builds (CTO_CVT_TS2=8 CTO_CVT_TS3=8, classifier unfused for both sides of that comparison).
# per-kernel averages by rocprofv3 --kernel-trace --stats (tools/kstats.sh).  Build the variants first (DESIGN.md 7.1 has the commands).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in amd abl1 abl2; do
  echo "== lib$v (classifier fused)"; CTO_LIB_PATH=$R/clairs_to_amd/libclairsto_$v.so bash $R/tools/kstats.sh 20 | grep -i "cvt_block\|sum of"
done
for v in amd abl2 ts8 ts8_abl2; do
  echo "== lib$v, CTO_CVT_NO_HEAD_FUSE=1"; CTO_CVT_NO_HEAD_FUSE=1 CTO_LIB_PATH=$R/clairs_to_amd/libclairsto_$v.so bash $R/tools/kstats.sh 20 | grep -i "cvt_block\|sum of"
done
