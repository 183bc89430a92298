cd /root/repo
timeout 600 python tools/ab.py clairs_to_amd/libclairsto_amd.so clairs_to_amd/libgate_even.so clairs_to_amd/libclairsto_amd.so clairs_to_amd/libgate_even.so --reps 80 2>&1 | tail -4
