#!/bin/bash
# One GPU-box round: tests, bench, rocprof kernel stats, PMC counters.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/round
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_stats -o stats -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/rocprof_pmc_fetch -o fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/rocprof_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/rocprof_pmc_write -o write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/rocprof_pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/rocprof_pmc_sq -o sq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/rocprof_pmc_sq.err
cd $R
python scripts/summarize_pmc.py $OUT $OUT/summary > $OUT/summary.log 2>&1
tail -3 $OUT/pytest_gpu.log; cat $OUT/smoke.log | tail -2; cat $OUT/bench.json | cut -c1-600; ls -R $OUT | head -40
