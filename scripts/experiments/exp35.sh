O=gpurun_out/exp35; mkdir -p $O
for i in 1 2; do ACINO_SWEEP=3 timeout 120 python scripts/sweep_stamps.py 100 3 > $O/stamps_$i.log 2>&1; grep -v amdgpu.ids $O/stamps_$i.log | head -22; done
