cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp11
O=$GRAFT_REPO_ROOT/gpurun_out/exp11
(timeout 300 python scripts/solver_sweep.py 5000 "-1,0,0" "0,3,3" "0,0,0") > $O/n5000.log 2>&1
(timeout 300 python scripts/solver_sweep.py 2500 "-1,0,0" "0,4,3" "0,0,0") > $O/n2500.log 2>&1
(timeout 300 python scripts/solver_sweep.py 1250 "-1,0,0" "0,4,3" "0,0,0") > $O/n1250.log 2>&1
grep -v amdgpu $O/n5000.log $O/n2500.log $O/n1250.log | cut -c1-300
