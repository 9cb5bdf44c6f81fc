O=$GRAFT_REPO_ROOT/gpurun_out/r6v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-step --precision f16 > $O/trace.log 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/f16_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete
head -40 $O/f16_kernel_stats.csv | cut -c1-150
