cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_split
mkdir -p $OUT
AICG_PRECISION=bf16x3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -30
