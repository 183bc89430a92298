# full GPU suite + default bench line, outputs under gpurun_out/
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_pytest_gpu.log
tail -5 gpurun_out/r3_pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r3_bench.json
