cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp12
O=$GRAFT_REPO_ROOT/gpurun_out/exp12
(timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sharded or multiprocess or rccl or window or size_sweep or edge_cases or separator_chain") > $O/pytest_shard.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q -k "not soak") > $O/pytest_chunk.log 2>&1
(timeout 900 python scripts/shard_model.py) > $O/shard_model.txt 2>&1
tail -n 12 $O/pytest_shard.log; tail -n 3 $O/pytest_chunk.log; grep -v amdgpu $O/shard_model.txt | cut -c1-400
