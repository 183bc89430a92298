cd /root/repo
for v in 0 1 0 1; do
  echo "CTO_CVT_NO_HEAD_FUSE=$v"
  CTO_CVT_NO_HEAD_FUSE=$v timeout 300 python tools/ab.py clairs_to_amd/libclairsto_amd.so --reps 60 2>&1 | tail -1
done
