cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp9
O=$GRAFT_REPO_ROOT/gpurun_out/exp9
for map in 75436210 76514320 76541320 75416320 76543210 65743210 47651320; do
  echo "== roles $map" >> $O/roles.log
  (ACINO_SWEEP_ROLES=$map timeout 200 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3") 2>&1 | grep -v amdgpu >> $O/roles.log
done
cat $O/roles.log | cut -c1-260
