#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + HBM counters of the judged bench command.
# Summaries land in gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/<tag>_*.{csv,json}.
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
# PMC passes: own runs, counters only (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950)
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $CMD > $OUT/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_sq -o bench -- $CMD > $OUT/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_profile.py $OUT $TAG > $OUT/summary.log 2>&1
tail -40 $OUT/summary.log
