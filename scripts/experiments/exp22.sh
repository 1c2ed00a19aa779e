mkdir -p gpurun_out/exp22
timeout 900 python -m pytest tests/test_gpu_sba.py -x -q -m gpu > gpurun_out/exp22/pytest.log 2>&1; tail -8 gpurun_out/exp22/pytest.log | cut -c1-300
