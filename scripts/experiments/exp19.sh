mkdir -p gpurun_out/exp19
O=gpurun_out/exp19
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_side_initial" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python scripts/e2e_phases.py > $O/e2e.log 2>&1; tail -4 $O/e2e.log
