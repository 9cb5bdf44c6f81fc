#!/bin/bash
# round 4, GPU call 3: 2-D Winograd forms, compile-time ablation variants, SQ counters of the eight-wave form
mkdir -p gpurun_out/r4c3
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 300 python tools/kbench_w2d.py > gpurun_out/r4c3/kbench_w2d.log 2>&1
timeout 400 python tools/kbench_w2d_ablate.py 48 256 3072 > gpurun_out/r4c3/ablate_L0.log 2>&1
timeout 400 python tools/kbench_w2d_ablate.py 144 64 768 > gpurun_out/r4c3/ablate_L2.log 2>&1
cd /tmp && export TMPDIR=/tmp
for lvl in "48 256 3072" "144 64 768"; do
  tag=$(echo $lvl | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d /root/repo/gpurun_out/r4c3/pmcA_$tag -o k -- python /root/repo/tools/kbench_w2d_one.py $lvl > /root/repo/gpurun_out/r4c3/pmcA_$tag.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d /root/repo/gpurun_out/r4c3/pmcB_$tag -o k -- python /root/repo/tools/kbench_w2d_one.py $lvl > /root/repo/gpurun_out/r4c3/pmcB_$tag.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_LDS --output-format csv -d /root/repo/gpurun_out/r4c3/pmcC_$tag -o k -- python /root/repo/tools/kbench_w2d_one.py $lvl > /root/repo/gpurun_out/r4c3/pmcC_$tag.log 2>&1
done
cd /root/repo
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r4c3/pmc*_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if "conv_w2d" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
        print(d, {k: (v[0], v[1] / max(1, v[0])) for k, v in agg.items()})
PY
find gpurun_out/r4c3 -name "*.csv" -size +200k -delete
tail -n 6 gpurun_out/r4c3/kbench_w2d.log
grep -v amdgpu gpurun_out/r4c3/ablate_L0.log; grep -v amdgpu gpurun_out/r4c3/ablate_L2.log
