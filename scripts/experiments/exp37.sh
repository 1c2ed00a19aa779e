O=gpurun_out/exp37; mkdir -p $O
ACINO_SWEEP=3 timeout 120 python scripts/sweep_stamps.py 100 3 > $O/stamps_1.log 2>&1; grep -v amdgpu.ids $O/stamps_1.log | head -18
