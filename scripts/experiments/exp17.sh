mkdir -p gpurun_out/exp17
timeout 600 python -m pytest tests/test_gpu_sba.py -x -q -m gpu -k "fused_path" > gpurun_out/exp17/pytest.log 2>&1; tail -5 gpurun_out/exp17/pytest.log
bash scripts/pmc_waits_sba.sh 2>&1 | grep "sba" | cut -c1-600
rm -rf gpurun_out/waits_sba/p1 gpurun_out/waits_sba/p2
