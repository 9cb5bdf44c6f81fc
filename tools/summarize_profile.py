"""Condense rocprofv3 CSV output (kernel trace + PMC passes of bench.py) into small files for profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
# third argument "split": the opt-in split-precision command's passes (trace_split / pmc_sq_split) -> <tag>_split_*
SPLIT = len(sys.argv) > 3 and sys.argv[3] == "split"
SUF = "_split" if SPLIT else ""
out_dir = os.path.join(src, "summary")
os.makedirs(out_dir, exist_ok=True)


def find(sub, pat):
    hits = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def is_conv(k):
    """the implicit-GEMM convolution family: conv_ws_kernel / conv_ws16_kernel (wave-specialised), conv_mfma_kernel /
    conv_mfma16_kernel (small-problem fallback)"""
    return k.startswith(("conv_ws_kernel", "conv_ws3_kernel", "conv_ws3s_kernel", "conv_ws3m16_kernel", "conv_ws3m16h_kernel", "conv_ws3w_kernel", "conv_w2d_kernel", "conv_g1_kernel", "conv_g1s_kernel", "conv_g1w_kernel", "conv_ws16_kernel", "conv_mfma_kernel", "conv_mfma16_kernel", "conv_pointwise_kernel"))


def short(name):
    name = name.replace("void ", "").replace("aicg::", "")
    return name[:name.index("(")] if "(" in name else name


def conv_calls_ratio(agg, summary):
    """conv launches in the PMC pass over conv launches in the trace pass (same command: 1.0)."""
    return 1.0


summary = {}
kt = find("trace" + SUF, "*kernel_trace.csv")
if kt:
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(kt)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += d
    total = sum(v[1] for v in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(os.path.join(out_dir, "%s%s_kernel_stats.csv" % (tag, SUF)), "w") as f:
        f.write("kernel,calls,total_us,avg_us,percent\n")
        for k, (c, t) in rows[:40]:
            f.write('"%s",%d,%.1f,%.2f,%.2f\n' % (k, c, t, t / c, 100 * t / total))
    summary["kernel_time_us_total"] = total
    summary["top_kernels"] = [{"kernel": k, "calls": c, "total_us": t, "avg_us": t / c} for k, (c, t) in rows[:12]]
    conv = [(c, t) for k, (c, t) in agg.items() if is_conv(k)]
    summary["conv_kernels"] = {"calls": sum(c for c, _ in conv), "total_us": sum(t for _, t in conv),
                                   "avg_us": sum(t for _, t in conv) / max(1, sum(c for c, _ in conv))}

for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    cc = None if SPLIT else find(sub, "*counter_collection.csv")
    if not cc:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(cc)):
        if r.get("Counter_Name") != counter:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    tot = {k: v for k, v in agg.items()}
    conv_calls = sum(c for k, (c, v) in tot.items() if is_conv(k))
    conv_val = sum(v for k, (c, v) in tot.items() if is_conv(k))
    summary[counter] = {"conv_kernels_calls": conv_calls, "conv_kernels_sum": conv_val,
                        "all_kernels_sum": sum(v for _, v in tot.values()),
                        "note": "rocprofv3 units: KiB; FETCH_SIZE under-reports wide coalesced reads 2x on gfx950 (MI355X_MICROARCH HBM)"}

cc = find("pmc_sq" + SUF, "*counter_collection.csv")
if cc:
    agg = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(cc)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    conv = defaultdict(float)
    for k, d in agg.items():
        if is_conv(k):
            for c, v in d.items():
                conv[c] += v
    summary["sq_conv_kernels"] = dict(conv)
    if conv.get("SQ_BUSY_CYCLES"):
        summary["sq_conv_kernels"]["mfma_busy_over_sq_busy"] = conv.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / conv["SQ_BUSY_CYCLES"]
        if summary.get("conv_kernels", {}).get("total_us"):
            # SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs; kernel time is the
            # trace pass' (an un-profiled pass of the same command: clocks differ by a few per cent between passes)
            t = summary["conv_kernels"]["total_us"] * 1e-6 * conv_calls_ratio(agg, summary)
            clk = conv["SQ_BUSY_CYCLES"] / 32.0 / t
            summary["sq_conv_kernels"]["shader_clock_ghz"] = clk / 1e9
            summary["sq_conv_kernels"]["mfma_pipe_busy_frac"] = conv.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / t / clk
            summary["sq_conv_kernels"]["wait_inst_any_over_wave_cycles"] = conv.get("SQ_WAIT_INST_ANY", 0) / max(1.0, conv.get("SQ_WAVE_CYCLES", 0))

# ---- per kernel (the six largest by time): HBM traffic per launch against nothing but itself, matrix-pipe occupancy
if kt and not SPLIT:
    per = {}
    top = [k for k, _ in rows[:8]]
    trace = {k: agg_t for k, agg_t in rows}
    fetch = defaultdict(lambda: [0, 0.0])
    write = defaultdict(lambda: [0, 0.0])
    for sub, counter, dst in (("pmc_fetch", "FETCH_SIZE", fetch), ("pmc_write", "WRITE_SIZE", write)):
        cc2 = find(sub, "*counter_collection.csv")
        if cc2:
            for r in csv.DictReader(open(cc2)):
                if r.get("Counter_Name") == counter:
                    a = dst[short(r["Kernel_Name"])]
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
    sq = defaultdict(lambda: defaultdict(float))
    cc3 = find("pmc_sq", "*counter_collection.csv")
    if cc3:
        for r in csv.DictReader(open(cc3)):
            sq[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in top:
        c, t = trace[k]
        row = {"calls": c, "avg_us": t / c, "total_us": t}
        if fetch[k][0] and write[k][0]:
            row["hbm_bytes_per_launch_2xfetch_plus_write"] = (2.0 * fetch[k][1] / fetch[k][0] + write[k][1] / write[k][0]) * 1024.0
            row["hbm_tb_per_s"] = row["hbm_bytes_per_launch_2xfetch_plus_write"] / (t / c * 1e-6) / 1e12
        d = sq.get(k)
        if d and d.get("SQ_BUSY_CYCLES"):
            clk = d["SQ_BUSY_CYCLES"] / 32.0 / (t * 1e-6)
            row["shader_clock_ghz"] = clk / 1e9
            row["mfma_pipe_busy_frac"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / (t * 1e-6) / clk
            row["executed_tflops_f32_mfma"] = d.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0 / (t * 1e-6) / 1e12
            row["wait_inst_any_over_wave_cycles"] = d.get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, d.get("SQ_WAVE_CYCLES", 0.0))
        per[k] = row
    summary["per_kernel_top"] = per

json.dump(summary, open(os.path.join(out_dir, "%s%s_summary.json" % (tag, SUF)), "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
