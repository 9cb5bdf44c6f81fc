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
# the raw kernel_stats of the trace pass rides along; the large per-dispatch CSVs stay on the box
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/summary/${TAG}_rocprofv3_kernel_stats_raw.csv 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
# the round's A/B tables on this box (round-robin): stride-2 extractor layers, the vocoder's 1-D Winograd layers
python tools/kbench_g1s.py 1,0 5 > $OUT/kbench_g1s_policy.txt 2>&1
python tools/kbench_g1w.py -1,0 5 1,3,5 > $OUT/kbench_g1w_policy.txt 2>&1
# the judged bench lines of this round
python bench.py --conv-shapes $OUT/conv_shapes_c3.json > $OUT/bench_c3_default.json 2> $OUT/bench_c3_default.err
python bench.py --config C5 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2>/dev/null
python bench.py --precision bf16x3 --no-cpu-baseline > $OUT/bench_c3_split.json 2> $OUT/bench_c3_split.err
python bench.py --precision f16 --no-cpu-baseline > $OUT/bench_c3_f16.json 2> $OUT/bench_c3_f16.err
python tools/kbench_half.py > $OUT/kbench_half_layers.txt 2>&1
python bench.py --config C2 --no-cpu-baseline > $OUT/bench_c2.json 2>/dev/null
python bench.py --config C4 --no-cpu-baseline > $OUT/bench_c4.json 2>/dev/null
python bench.py --preset fp32 --no-cpu-baseline > $OUT/bench_c3_fp32preset.json 2>/dev/null
python bench.py --mdx-models 3 --no-cpu-baseline > $OUT/bench_c3_3mdx.json 2>/dev/null
for f in $OUT/bench_*.json; do echo $f; cut -c1-200 $f; done
