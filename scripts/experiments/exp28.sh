O=gpurun_out/exp28; mkdir -p $O
for mask in 184 56 0; do
  ACINO_SWEEP=2 timeout 120 python scripts/sweep_stamps.py 100 3 $mask > $O/stamps_$mask.log 2>&1
  echo "=== mask $mask"; grep -v amdgpu.ids $O/stamps_$mask.log | grep "spike\|end of node\|parallel part\|G in LDS\|SIMD\|pivots 4"
done
