"""Who runs when: the timed step of `bench.py` as a rocprofv3 kernel trace, per hardware queue.

  (GPU box)  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline
             python tools/timeline.py reduce <dir>/.../bench_kernel_trace.csv profiles/r05_kernel_trace_step.csv.gz
  (anywhere) python tools/timeline.py summary profiles/r05_kernel_trace_step.csv.gz > profiles/r05_timeline.json

`reduce` keeps (start ns, end ns, queue, kernel name[:70]) of every dispatch; `summary` cuts out the second step (the timed one: the
first is the warm-up, the third carries the per-launch events of the stage table) at the MDX STFT launches and reports, per phase (MDX
separation | HuBERT with the f0 branch underneath | synthesis) and per queue: kernels, busy time, first / last timestamp; the union
busy time of the device and its idle gaps; the f0 branch's recurrence launches; the bursts of the encoder-half stream and how long the
main stream stood idle behind HuBERT.  Queue ids are the profiler's: the main stream is the one that runs conv_w2d, the f0 stream the
one that runs gru4, the encoder-half stream the one that runs attn_fwd<96, .>.
"""
import collections
import csv
import gzip
import json
import sys


def reduce_trace(src, dst):
    with open(src) as f, gzip.open(dst, "wt") as g:
        for r in csv.DictReader(f):
            g.write("%s,%s,%s,%s\n" % (r["Start_Timestamp"], r["End_Timestamp"], r.get("Queue_Id", ""), r["Kernel_Name"][:70].replace(",", ";")))


def load(path):
    rows = []
    for line in gzip.open(path, "rt"):
        a = line.rstrip("\n").split(",")          # names carry no commas (reduce replaces them); an older form has a stream id column
        rows.append((int(a[0]), int(a[1]), a[2], a[-1]))
    rows.sort()
    return rows


def union(ev):
    ev = sorted(ev)
    busy, gaps = 0, []
    cs, ce = ev[0]
    for s, e in ev[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append((s - ce, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + ce - cs, gaps


def summary(path, step=1):
    rows = load(path)
    stft = [r for r in rows if "stft_reg" in r[3] and "istft" not in r[3] and "Plan<16" in r[3]]
    per_step = len(stft) // 3 if len(stft) >= 3 else len(stft)
    a = stft[step * per_step][0]
    b = stft[(step + 1) * per_step][0] if len(stft) > (step + 1) * per_step else rows[-1][1] + 1
    S = [r for r in rows if a <= r[0] < b]
    t0 = S[0][0]
    ms = lambda t: round((t - t0) / 1e6, 2)
    q_main = next(r[2] for r in S if "conv_w2d" in r[3])
    q_f0 = next((r[2] for r in S if "gru4" in r[3]), None)
    q_enc = next((r[2] for r in S if "attn_fwd_kernel<96" in r[3] and r[2] != q_main), None)
    hub_end = max(r[1] for r in S if r[2] == q_main and "attn_fwd_kernel<64" in r[3])
    mdx_end = max(r[1] for r in S if "istft" in r[3])
    voc = [r for r in S if r[2] == q_main and "sine_frame" in r[3]]
    out = {"trace": path, "step_ms": ms(max(r[1] for r in S)), "kernels": len(S),
           "queues": {"main": q_main, "f0": q_f0, "encoder_half": q_enc}}
    busy, gaps = union([(r[0], r[1]) for r in S])
    out["device_busy_ms"] = round(busy / 1e6, 2)
    out["device_idle_ms"] = round(sum(g for g, _ in gaps) / 1e6, 2)
    out["largest_idle_gaps_us_at_ms"] = [(round(g / 1e3, 1), ms(t)) for g, t in sorted(gaps, reverse=True)[:8]]
    phases = {"mdx": (t0, mdx_end), "hubert_f0": (mdx_end, voc[0][0] if voc else hub_end), "synth": (voc[0][0] if voc else hub_end, b)}
    if q_enc is not None:
        first_enc = min(r[0] for r in S if r[2] == q_enc)
        phases["hubert_f0"] = (mdx_end, first_enc)
        phases["synth"] = (first_enc, b)
    out["phases"] = {}
    for name, (lo, hi) in phases.items():
        P = [r for r in S if lo <= r[0] < hi]
        d = {"from_ms": ms(lo), "to_ms": ms(max(r[1] for r in P)), "kernel_ms_sum": round(sum(r[1] - r[0] for r in P) / 1e6, 2), "queues": {}}
        for q in sorted(set(r[2] for r in P)):
            Q = [r for r in P if r[2] == q]
            top = collections.Counter()
            for r in Q:
                top[r[3].split("(")[0].replace("void aicg::", "").replace("aicg::", "")[:44]] += r[1] - r[0]
            d["queues"][q] = {"kernels": len(Q), "busy_ms": round(sum(r[1] - r[0] for r in Q) / 1e6, 2), "first_ms": ms(Q[0][0]),
                              "last_ms": ms(max(r[1] for r in Q)), "top": {k: round(v / 1e6, 2) for k, v in top.most_common(6)}}
        out["phases"][name] = d
    if q_f0 is not None:
        out["f0_recurrence_launches_ms"] = [(ms(r[0]), ms(r[1])) for r in S if "gru4" in r[3]]
        out["hubert_last_attention_ends_ms"] = ms(hub_end)
    if q_enc is not None:
        E = [r for r in S if r[2] == q_enc]
        bursts = [[E[0]]]
        for r in E[1:]:
            if r[0] - bursts[-1][-1][1] > 1e6:
                bursts.append([])
            bursts[-1].append(r)
        out["encoder_half_bursts"] = [{"from_ms": ms(g[0][0]), "to_ms": ms(g[-1][1]), "kernels": len(g), "busy_ms": round(sum(r[1] - r[0] for r in g) / 1e6, 2)} for g in bursts]
        prev, idle, big = None, 0, []
        for r in [r for r in S if r[2] == q_main and r[0] > hub_end]:
            if prev and r[0] > prev:
                idle += r[0] - prev
                if r[0] - prev > 300e3:
                    big.append((round((r[0] - prev) / 1e3), ms(prev)))
            prev = max(prev or 0, r[1])
        out["main_stream_idle_behind_hubert_ms"] = round(idle / 1e6, 2)
        out["main_stream_gaps_over_300us"] = big
        out["vocoders_start_ms"] = [ms(r[0]) for r in voc]
    return out


if __name__ == "__main__":
    if sys.argv[1] == "reduce":
        reduce_trace(sys.argv[2], sys.argv[3])
    else:
        print(json.dumps(summary(sys.argv[2]), indent=1))
