# round 5: who slows whom in the T = G F sweep - stamped workgroup with subsets of the strip waves idle (skip mask by role)
O=gpurun_out/exp27; mkdir -p $O
for mask in 0 48 8 56 192 248; do
  ACINO_SWEEP=2 timeout 120 python scripts/sweep_stamps.py 100 3 $mask > $O/stamps_$mask.log 2>&1
  echo "=== mask $mask"; grep -v amdgpu.ids $O/stamps_$mask.log | grep "chain\|helper\|end of node\|parallel part\|G in LDS\|SIMD"
done
