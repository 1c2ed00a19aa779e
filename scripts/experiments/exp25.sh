# sweep micro-variants, rebuilt on the box: trailing-product group size, helper priority, spin sleep
mkdir -p gpurun_out/exp25
O=gpurun_out/exp25
F=acinoset_amd/csrc/chunk.hip
cp $F /tmp/chunk_orig.hip
run() { python -c "from acinoset_amd import _lib; _lib.build(force=True, verbose=False)" > $O/build_$1.log 2>&1; timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver_$1.log 2>&1; echo "== $1: $(grep -o "'chunk_sweep': [0-9.]*" $O/solver_$1.log | tr '\n' ' ') $(grep -o "[0-9.]* us/step" $O/solver_$1.log | tr '\n' ' ') $(grep -o "cost23=[-0-9.]*" $O/solver_$1.log | tail -1)"; }
run base
# trailing products two at a time
python - <<'PY'
p='acinoset_amd/csrc/chunk.hip'; s=open('/tmp/chunk_orig.hip').read()
s=s.replace("  for (int q0 = 0; q0 < NTOT; q0 += 4) {\n    d4 a[4];","  for (int q0 = 0; q0 < NTOT; q0 += 2) {\n    d4 a[2];")
a=s.index("  for (int q0 = 0; q0 < NTOT; q0 += 2) {"); b=s.index("// One 16-column strip (tile column jb)")
s=s[:a]+s[a:b].replace("for (int q = 0; q < 4; ++q)","for (int q = 0; q < 2; ++q)")+s[b:]
open(p,'w').write(s)
PY
run trail2
python - <<'PY'
p='acinoset_amd/csrc/chunk.hip'; s=open('/tmp/chunk_orig.hip').read()
s=s.replace("  for (int q0 = 0; q0 < NTOT; q0 += 4) {\n    d4 a[4];","  for (int q0 = 0; q0 < NTOT; q0 += 7) {\n    d4 a[7];")
a=s.index("  for (int q0 = 0; q0 < NTOT; q0 += 7) {"); b=s.index("// One 16-column strip (tile column jb)")
s=s[:a]+s[a:b].replace("for (int q = 0; q < 4; ++q)","for (int q = 0; q < 7; ++q)")+s[b:]
open(p,'w').write(s)
PY
run trail7
python - <<'PY'
p='acinoset_amd/csrc/chunk.hip'; s=open('/tmp/chunk_orig.hip').read()
s=s.replace("    __builtin_amdgcn_s_setprio(2);                     // a helper's short bursts","    __builtin_amdgcn_s_setprio(3);                     // a helper's short bursts")
open(p,'w').write(s)
PY
run helper_prio3
python - <<'PY'
p='acinoset_amd/csrc/chunk.hip'; s=open('/tmp/chunk_orig.hip').read()
s=s.replace("    __builtin_amdgcn_s_setprio(2);                     // a helper's short bursts","    __builtin_amdgcn_s_setprio(0);                     // a helper's short bursts")
s=s.replace("    __builtin_amdgcn_s_setprio(3);                     // the chain's VALU wins","    __builtin_amdgcn_s_setprio(0);                     // the chain's VALU wins")
open(p,'w').write(s)
PY
run no_prio
python - <<'PY'
p='acinoset_amd/csrc/chunk.hip'; s=open('/tmp/chunk_orig.hip').read()
s=s.replace("< target) __builtin_amdgcn_s_sleep(1);","< target) __builtin_amdgcn_s_sleep(0);")
open(p,'w').write(s)
PY
run sleep0
python - <<'PY'
p='acinoset_amd/csrc/chunk.hip'; s=open('/tmp/chunk_orig.hip').read()
s=s.replace("< target) __builtin_amdgcn_s_sleep(1);","< target) __builtin_amdgcn_s_sleep(4);")
open(p,'w').write(s)
PY
run sleep4
cp /tmp/chunk_orig.hip $F
