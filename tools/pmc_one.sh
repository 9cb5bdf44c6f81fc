#!/bin/bash
# PMC counter passes over one conv shape (tools/kbench_one.py); prints per-kernel counter averages.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_one
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$@"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH" ; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -o one -- python $GRAFT_REPO_ROOT/tools/kbench_one.py $ARGS > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python - <<'P'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_one"
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "conv" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kname, d in acc.items():
        print(kname)
        for c, v in d.items():
            print(f"   {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
P
