set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q -k "native_pipeline" 2>&1 | tail -8
for ts in 1 0; do
  CTO_TILE_STREAM=$ts timeout 600 python tools/e2e_bench.py --kind text --chunks 24 --sites 10000 --producers 8 --writers 2 --pipeline native --repeats 3 2>&1 | tail -3 > gpurun_out/e2e_10k_ts$ts.log
  CTO_TILE_STREAM=$ts timeout 600 python tools/e2e_bench.py --kind text --chunks 48 --sites 4096 --producers 8 --writers 2 --pipeline native --repeats 3 2>&1 | tail -3 > gpurun_out/e2e_4096_ts$ts.log
  CTO_TILE_STREAM=$ts timeout 600 python tools/e2e_bench.py --kind text --chunks 40 --sites 5000 --producers 8 --writers 2 --pipeline native --repeats 3 2>&1 | tail -3 > gpurun_out/e2e_5000_ts$ts.log
done
tail -n 3 gpurun_out/e2e_*ts*.log
