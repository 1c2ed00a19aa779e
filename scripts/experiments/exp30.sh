# what slows the factor trio beside the strip waves: their matrix instructions (mode 256: no LDS reads) or their LDS reads (mode 512: no matrix instructions)?
O=gpurun_out/exp30; mkdir -p $O
for mask in 0 256 512 248; do
  ACINO_SWEEP=2 timeout 120 python scripts/sweep_stamps.py 100 3 $mask > $O/stamps_$mask.log 2>&1
  echo "=== mask $mask"; grep -v amdgpu.ids $O/stamps_$mask.log | grep "G in\|arrived\|tiles of G\|written\|parallel part\|chain\|helper 1\|end of node"
done
