mkdir -p gpurun_out/exp24
O=gpurun_out/exp24
run() { python -c "from acinoset_amd import _lib; _lib.build(force=True, verbose=False)" > $O/build_$1.log 2>&1; timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" > $O/solver_$1.log 2>&1; echo "== $1: $(grep -o "'assemble': [0-9.]*" $O/solver_$1.log | tail -1) $(grep -o "[0-9.]* us/step" $O/solver_$1.log | tail -1)"; }
run base
sed -i 's/^__global__ void __launch_bounds__(256)\nk_fte_assemble/X/' acinoset_amd/csrc/fte_assemble.hip
python - <<'PY'
import re
p='acinoset_amd/csrc/fte_assemble.hip'; s=open(p).read()
s=s.replace("__global__ void __launch_bounds__(256)\nk_fte_assemble(","__global__ void __launch_bounds__(256, ASM_OCC)\nk_fte_assemble(",1)
s=s.replace("namespace acino {\n__global__","#ifndef ASM_OCC\n#define ASM_OCC 3\n#endif\nnamespace acino {\n__global__",1)
open(p,'w').write(s)
PY
for cfg in "8 3" "6 4" "6 3" "4 4" "4 5" "5 4" "10 2" "12 2"; do set -- $cfg
  sed -i "s/^constexpr int FPB = [0-9]*;/constexpr int FPB = $1;/" acinoset_amd/csrc/fte_kernels.hpp
  sed -i "s/^#define ASM_OCC [0-9]*/#define ASM_OCC $2/" acinoset_amd/csrc/fte_assemble.hip
  run "fpb$1_occ$2"
done
