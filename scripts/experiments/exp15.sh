mkdir -p gpurun_out/exp15
for mask in 0 8 16 32 64 128 0xf8; do
  echo "=== skip mask $mask" >> gpurun_out/exp15/stamps.txt
  timeout 300 python scripts/sweep_stamps.py 100 5 $mask >> gpurun_out/exp15/stamps.txt 2>&1
done
grep -n "SIMD of\|=== \|end of node\|pivots\|panel\|T stored" gpurun_out/exp15/stamps.txt
