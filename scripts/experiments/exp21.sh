mkdir -p gpurun_out/exp21
O=gpurun_out/exp21
export TMPDIR=/tmp
prof() { ( cd /tmp; PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$1 -o p -- python $GRAFT_REPO_ROOT/scripts/solver_sweep.py 10000 "0,2,3" > /dev/null 2>&1 ); f=$(find $O/prof_$1 -name "*kernel_stats.csv" | head -1); echo "== $1"; grep "k_chunk_backsub\|k_chunk_sweep" "$f" | cut -c1-200; rm -rf $O/prof_$1; }
prof base
sed -i 's/^constexpr int BK_T = 512;/#define EXP_NOFETCH 1\nconstexpr int BK_T = 512;/' acinoset_amd/csrc/chunk.hip
md5sum acinoset_amd/libacinoset_hip.so; T0=$SECONDS; python -c "from acinoset_amd import _lib; _lib.build(force=True, verbose=False)" > $O/build.log 2>&1; echo "build $((SECONDS-T0)) s"; md5sum acinoset_amd/libacinoset_hip.so; grep -c EXP_NOFETCH acinoset_amd/csrc/chunk.hip
prof nofetch
