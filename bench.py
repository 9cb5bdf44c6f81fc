#!/usr/bin/env python
"""Headline benchmark: real-time factor (audio-seconds / wall-seconds) of MDX-Net separation + RVC voice conversion
on a 4-minute 44.1 kHz stereo track per GPU (BASELINE.json metric; configs of SURVEY 8d).

  python bench.py --gpus N --steps K --warmup W [--config C3]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

--config  C3 (default, the headline): MDX (1 model) -> separated vocals -> 44.1 kHz stereo to 16 kHz mono on the device ->
              VC.pipeline with rmvpe f0.  One "step" = one synthetic track of 240 s x N through both stages.
          C2: the MDX stage alone.          C4: C3 with f0_method = mangio-crepe (hop 128).
          C5: one 1800 s track in total, sharded over the N ranks (strong scaling).
--mdx-models 3  runs main.py's chain of three separations (Voc_FT -> KARA-class -> Reverb_HQ-class geometries) instead of one.

MDX: peak-normalised stereo wave resident in HBM -> framed STFT -> TFC-TDF U-Net (Voc_FT-class geometry: dim_f 3072, dim_t 256,
n_fft 7680, denoise on => 2 network passes per window) -> iSTFT -> window join.  RVC: 48 Hz zero-phase high-pass, cut search,
f0 on the whole track, per chunk HuBERT-base -> nearest x2 + protect -> SynthesizerTrnMs768NSFsid 40 kHz, RMS mix, int16.
The hand-over between the stages is the opt-in device path (aicg_resample_poly, scipy.signal.resample_poly semantics); the
reference goes through a PCM-16 WAV and ffmpeg (not installed here).  Model loading, file I/O and main.py's mixing / effects
are outside the timed region.  At N > 1 the window list (MDX) and the chunk list (RVC) are sharded across ranks and joined by
RCCL all-gathers.  All network parameters are seeded random tensors of the real architectures (no checkpoints exist offline);
data is synthetic.  Compute dtype fp32 (MFMA f32).
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRACK_S = 240.0
PEAK_F32_MFMA = 157.3   # TFLOP/s, MI355X_MICROARCH.md
PEAK_BF16_MFMA = 2516.6  # TFLOP/s dense (16 x the fp32 MFMA rate, MI355X_MICROARCH.md "1/16 of BF16 MFMA")
MFMA_PEAK = PEAK_F32_MFMA  # of the arithmetic in use: --precision bf16x3 spends three bf16 MFMAs per fp32 product -> bf16 / 3
PEAK_HBM = 8000.0       # GB/s spec (6290 measured float4 copy)


def build_models(device, config, n_mdx, tiny=False, preset="fp16", half=False):
    """`tiny` (tests/test_bench_launch.py only, never on a GPU box): the miniature networks of the CPU suite, so that the launch /
    sharding / reporting logic of this file can be exercised end to end on the kernel emulator."""
    from aicovergen_amd.hubert import HubertModel
    from aicovergen_amd.infer_pack.models import SynthesizerTrnMs768NSFsid
    from aicovergen_amd.mdx import MDX, MDXModel
    from aicovergen_amd.rmvpe import RMVPE
    from aicovergen_amd.rvc import Config
    from aicovergen_amd.vc_infer_pipeline import VC
    from synthetic import weights
    mdxs = []
    for i, mcfg in enumerate([weights.MDX_TINY] if tiny else [weights.MDX_VOC_FT, weights.MDX_KARA2, weights.MDX_REVERB_HQ][:n_mdx]):
        model = MDXModel(device, mcfg["dim_f"], mcfg["dim_t"], mcfg["n_fft"], hop=64 if tiny else 1024, stem_name="Vocals",
                         compensation=1.021)
        mdxs.append(MDX(None, model, state_dict=weights.mdx_state_dict(mcfg, 1234 + i)))
    if config == "C2":
        return mdxs, None, None, None
    if tiny:
        nets = weights.small_model_set(1234)
        cfg = type("TinyCfg", (), dict(x_pad=1, x_query=1, x_center=1, x_max=2, is_half=False, device=device))()
        vc = VC(nets["synth_cfg"][-1], cfg)
        hub = HubertModel(nets["hubert_sd"], nets["hubert_cfg"]).to(device)
        vc.model_rmvpe = RMVPE(None, False, device, state_dict=nets["rmvpe_sd"])
        net_g = SynthesizerTrnMs768NSFsid(*nets["synth_cfg"], is_half=False)
        del net_g.enc_q
        net_g.load_state_dict(nets["synth_sd"], strict=False)
        net_g.eval().to(device)
        return mdxs, vc, hub, net_g
    # Config's is_half only selects the chunk geometry here (all kernels compute in fp32): True = what main.py:196 passes, the
    # "fp16" preset x_pad, x_query, x_center, x_max = 3, 10, 60, 65; False = the fp32 preset 1, 6, 38, 41 (src/rvc.py:76-93)
    cfg = Config(str(device), preset == "fp16")
    cfg.device = device
    vc = VC(40000, cfg)
    hub = HubertModel(weights.hubert_state_dict(weights.HUBERT_BASE, 1234), weights.HUBERT_BASE).to(device)
    if config == "C4":
        from aicovergen_amd.crepe import Crepe
        vc.model_crepe = {"full": Crepe(weights.crepe_state_dict(weights.CREPE_FULL, 1237), device)}
    else:
        vc.model_rmvpe = RMVPE(None, False, device, state_dict=weights.rmvpe_state_dict(weights.RMVPE_FULL, 1235))
    net_g = SynthesizerTrnMs768NSFsid(*weights.SYNTH_CFG_40K_V2, is_half=False)
    del net_g.enc_q
    net_g.load_state_dict(weights.synth_state_dict(weights.SYNTH_CFG_40K_V2, 1236), strict=False)
    net_g.eval().to(device)
    if half:   # --precision f16 (AICG_HALF=1 is set): what load_hubert / get_vc do when is_half is passed (src/rvc.py:103-104,137-138)
        hub, net_g = hub.half(), net_g.half()
    return mdxs, vc, hub, net_g


def one_step(config, mdxs, vc, hub, net_g, wave44_dev, group, emu=False):
    from aicovergen_amd import dist as adist
    from aicovergen_amd import ops
    t0 = time.perf_counter()
    adist.last_join.clear()
    sep = wave44_dev
    for i, m in enumerate(mdxs):   # main.py feeds each separation the previous one's stem; run_mdx peak-normalises its input
        if i > 0:
            sep = sep / ops.absmax(sep.reshape(-1)).clamp_min(1e-12)   # (mdx.py:258-259)
        sep = adist.mdx_separate(m, sep, True, 2, group)
    if not emu:
        torch.cuda.synchronize()  # stage boundary (the reference writes the stems to disk here)
    mdx_s = time.perf_counter() - t0
    if config == "C2":
        return sep, None, [0, 0, 0], dict(adist.last_join, mdx_s=mdx_s)
    t1 = time.perf_counter()
    wave16 = ops.resample_poly_mono(sep, 44100, 16000)      # stereo 44.1 kHz -> mono 16 kHz, stays in HBM
    if not emu:
        torch.cuda.synchronize()
    times = [0, 0, 0]
    method = "mangio-crepe" if config == "C4" else "rmvpe"
    out = vc.pipeline(hub, net_g, 0, wave16, "synthetic.wav", times, 0, method, "", 0.5, 1, 3, vc.t_pad_tgt // vc.x_pad, 0, 0.25,
                      "v2", 0.33, 128, group=group, noise_seed=1234)
    return sep, out, times, dict(vc.last_profile, mdx_s=mdx_s, **adist.last_join, resample_s=time.perf_counter() - t1 - sum(
        vc.last_profile[k] for k in ("plan_s", "f0_s", "chunks_s", "post_s")))


def cpu_baseline():
    """The oracle (CPU restatement of the reference, "port") timed on this box's host cores on a bounded sample: one MDX window pair
    (denoise => 2 U-Net passes, 5.57 s of audio) + the RVC pipeline on 30 s (BASELINE C1's length: one 576 000-sample chunk, so
    the chunk padding is amortised as in a long track).  Threads: min(host cores, 16) -- torch's intra-op pool degrades badly
    beyond that on these layer sizes (256 threads measured 100x slower than 16 on the GPU box's EPYC)."""
    from oracle import mdxnet
    from oracle import pipeline as opipe
    from synthetic import weights
    from synthetic.inputs import song_like, vocal_like
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    mcfg = weights.MDX_VOC_FT
    sd = weights.mdx_state_dict(mcfg, 1234)
    chunk = 1024 * (mcfg["dim_t"] - 1)
    x = torch.from_numpy(song_like(chunk / 44100.0 + 0.01, 44100, 1)[:, :chunk]).unsqueeze(0)
    gen_s = (chunk - mcfg["n_fft"]) / 44100.0
    t0 = time.time()
    with torch.no_grad():
        for sgn in (1.0, -1.0):
            mdxnet.istft(mdxnet.unet(sd, mcfg, mdxnet.stft(sgn * x, mcfg["n_fft"], 1024, mcfg["dim_f"])), mcfg["n_fft"], 1024)
    mdx_cost = (time.time() - t0) / gen_s          # CPU seconds per audio second
    nets = weights.full_model_set(1234)
    geo = opipe.Geometry(40000, 3, 10, 60, 65)
    seconds_rvc = 30.0
    a = vocal_like(seconds_rvc, 16000, 2)
    t0 = time.time()
    opipe.vc_pipeline(nets, geo, a, tgt_sr=40000)
    rvc_cost = (time.time() - t0) / seconds_rvc
    return {"value": 1.0 / (mdx_cost + rvc_cost), "unit": "x real-time", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "oracle port: 1 MDX window pair (5.57 s, denoise) + VC.pipeline (rmvpe) on %.0f s; CPU seconds per audio second: "
                      "%.2f (MDX) + %.2f (RVC)" % (seconds_rvc, mdx_cost, rvc_cost)}


def pmc_traffic_per_launch():
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes of this same command (profiles/r03_summary.json, falling
    back to earlier rounds): FETCH_SIZE (x 2: the gfx950 under-report for wide reads, MI355X_MICROARCH.md) + WRITE_SIZE, KiB units, summed
    over the conv kernels; None when no summary is committed.  The counters cannot be collected inside the timed run."""
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_summary.json" % tag)
        try:
            s = json.load(open(path))
            f, w = s["FETCH_SIZE"], s["WRITE_SIZE"]
            raw = (f["conv_kernels_sum"] / f["conv_kernels_calls"] + w["conv_kernels_sum"] / w["conv_kernels_calls"]) * 1024.0
            cor = (2.0 * f["conv_kernels_sum"] / f["conv_kernels_calls"] + w["conv_kernels_sum"] / w["conv_kernels_calls"]) * 1024.0
            return {"raw": raw, "fetch_x2": cor, "source": "profiles/%s_summary.json" % tag}
        except Exception:
            continue
    return None


def stage_table(conv, stages, steps):
    """Per-stage roofline fractions (report per stage; STFT / iSTFT stand-alone)."""
    rows = [{"stage": "conv family (implicit GEMM + Winograd)", "bound": "mfma", "ms_per_step": conv["ms"] / steps,
             "achieved": conv["tflops"], "executed": conv["tflops_executed"], "unit": "TFLOP/s",
             "frac": conv["tflops_executed"] / MFMA_PEAK, "frac_algorithmic": conv["tflops"] / MFMA_PEAK}]
    for name, r in sorted(stages.items(), key=lambda kv: -kv[1]["ms"]):
        if r["ms"] <= 0:
            continue
        if r["flops"] > 0:
            a = r["flops"] / (r["ms"] * 1e-3) / 1e12
            rows.append({"stage": name, "bound": "mfma", "ms_per_step": r["ms"] / steps, "achieved": a, "unit": "TFLOP/s",
                         "frac": a / (MFMA_PEAK if name == "tdf_gemm_nt" else PEAK_F32_MFMA)})
        else:
            a = r["bytes"] / (r["ms"] * 1e-3) / 1e9
            rows.append({"stage": name, "bound": "hbm", "ms_per_step": r["ms"] / steps, "achieved": a, "unit": "GB/s",
                         "frac": a / PEAK_HBM})
    return rows


def reference_cpu_record():
    """The reference's OWN VC.pipeline (its unmodified code, CPU fp32) on BASELINE C1's 30 s input, as timed by
    tests/golden/make_golden.py when it produced the committed golden -- in the BUILD container (8 vCPU), not on this box:
    /root/reference does not travel.  Reported next to the live port timing, never mixed into it."""
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "pipeline_c1_30s.npz"))
        sec, thr, dur = float(g["ref_cpu_seconds"][0]), int(g["ref_threads"][0]), float(g["seconds"][0])
        return {"ref_cpu_seconds": sec, "ref_audio_seconds": dur, "ref_threads": thr, "ref_rtf_rvc_only": dur / sec,
                "ref_where": "reference src/vc_infer_pipeline.py VC.pipeline (rmvpe, fp32, C1 30 s): a SINGLE run in the build container "
                             "(another host than this box), recorded in tests/golden/pipeline_c1_30s.npz; the round-4 judge measured "
                             "22-28 s for the same run on a 2.6 GHz Xeon; the round-5 builder 16-20 s per 30 s input on 8 threads when "
                             "the other C1 inputs were generated -- a few times real time on a CPU either way"}
    except Exception:
        return {}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves, exactly the way the
    driver does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), one rank per GPU
    over RCCL, and pass rank 0's single JSON line through.  Returns the exit code, or None when this process is a rank itself
    (RANK / WORLD_SIZE present) or N == 1."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return None
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's only working mode on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["C2", "C3", "C4", "C5"], default="C3")
    ap.add_argument("--mdx-models", type=int, default=1, choices=[1, 3])
    ap.add_argument("--preset", choices=["fp16", "fp32"], default="fp16",
                    help="RVC chunk geometry (src/rvc.py:76-93): fp16 = x_pad,x_query,x_center,x_max 3,10,60,65 (what main.py "
                         "selects, the headline), fp32 = 1,6,38,41.  Geometry only: the arithmetic is fp32 either way")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-step", action="store_true", help="skip the extra instrumented step behind the timed region")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "f16"], default=None,
                    help="arithmetic of the conv / TDF GEMM family: fp32 MFMA (default, the headline) or the opt-in split precision "
                         "(bf16 hi + lo operands, 3 bf16 MFMAs per product, fp32 accumulation); f16: the reference's is_half mode on "
                         "the RVC half -- HuBERT's and the synthesizer's LDS-DMA staged layers take fp16 operands on the matrix pipe (AICG_HALF=1 + "
                         ".half(); fp32 activations and accumulation; MDX-Net, f0, attention stay fp32) -- a SEPARATE line, never the headline; "
                         "default: $AICG_PRECISION or fp32")
    ap.add_argument("--track-seconds", type=float, default=None)
    ap.add_argument("--dump", type=str, default=None, help="write the last step's outputs (npz) for cross-checking runs")
    ap.add_argument("--conv-shapes", type=str, default=None, help="write the instrumented step's per-layer-shape conv table (JSON)")
    args = ap.parse_args()

    rc = self_launch(args)
    if rc is not None:
        sys.exit(rc)

    # stdout carries exactly ONE line (rank 0's JSON): everything else this process or its native libraries (gloo / RCCL banners,
    # the synthesizer constructors' prints) write to file descriptor 1 goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as td
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to report a line for a "
                 "different job size" % (args.gpus, world))
    # TEST HOOK (tests/test_bench_launch.py): the kernel emulator on host tensors with the miniature networks -- exercises
    # the launch, sharding and reporting logic of this file without a GPU.  Never set on a GPU box; the line says so.
    emu = os.environ.get("AICG_BENCH_EMU") == "1"
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import conftest
        conftest._bind("emu")
        device = torch.device("cpu")
    else:
        if os.environ.get("AICG_FORCE_DEVICE") is not None:   # debugging aid: several ranks on one GPU (with gloo)
            local = int(os.environ["AICG_FORCE_DEVICE"])
        if local >= torch.cuda.device_count():
            sys.exit("bench.py: rank %d needs cuda:%d but only %d device(s) are visible" % (rank, local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    group = None
    # One rank normally has no process group.  AICG_DIST_BACKEND set explicitly under a launcher (RANK in the environment) creates the
    # one-rank group anyway; together with AICG_FORCE_COLLECTIVES=1 every join of the step then runs through RCCL on a one-GPU box
    # (profiles/r05_bench_c3_one_rank_rccl.json) -- the line says so in config.collectives
    if world > 1 or (os.environ.get("AICG_DIST_BACKEND") and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("AICG_DIST_BACKEND", "gloo" if emu else "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            td.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            td.init_process_group(backend=backend, rank=rank, world_size=world)

    from aicovergen_amd import ops
    from synthetic.inputs import song_like
    global MFMA_PEAK
    if args.precision is not None:
        ops.split_precision = args.precision == "bf16x3"
    half_mode = args.precision == "f16"
    if half_mode:
        os.environ["AICG_HALF"] = "1"
    split_mode = bool(ops.split_precision)
    MFMA_PEAK = PEAK_BF16_MFMA / 3.0 if split_mode else PEAK_F32_MFMA
    strong = args.config == "C5"
    if strong:
        seconds = args.track_seconds or 1800.0        # one track in total, sharded over the ranks
    else:
        seconds = (args.track_seconds or TRACK_S) * world   # weak scaling: 240 s per GPU
    with contextlib.redirect_stdout(sys.stderr):      # the synthesizer constructors print like the reference's do
        mdxs, vc, hub, net_g = build_models(device, args.config, args.mdx_models, tiny=emu, preset=args.preset, half=half_mode)
    wave44 = song_like(seconds, 44100, 1234)
    wave44 = wave44 / max(np.max(wave44), abs(np.min(wave44)))
    wave44_dev = torch.from_numpy(wave44).to(device)     # inputs resident in HBM before the timed region

    def sync():
        if not emu:
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            td.barrier()
        sync()

    with contextlib.redirect_stdout(sys.stderr):
        for _ in range(args.warmup):
            one_step(args.config, mdxs, vc, hub, net_g, wave44_dev, group, emu)
        # ---- the timed region: K steps, no instrumentation (no HIP events, no per-launch accounting) ----
        barrier()
        t0 = time.perf_counter()
        stage = [0.0, 0.0, 0.0]
        split = {}
        for _ in range(args.steps):
            sep, out, times, prof_s = one_step(args.config, mdxs, vc, hub, net_g, wave44_dev, group, emu)
            stage = [a + b for a, b in zip(stage, times)]
            for k, v in prof_s.items():
                split[k] = split.get(k, 0.0) + v / args.steps
        barrier()
        dt = time.perf_counter() - t0
        # ---- one more step of the same work with a HIP event pair around every conv / staged launch (roofline + stage table):
        # ~2 700 event records per step, kept OUT of the timed region; its own wall time is reported as profiled_step_ms ----
        prof = sprof = None
        prof_ms = None
        if not args.no_profile_step and not emu:
            if rank == 0:
                prof, sprof = ops.ConvProfile(), ops.StageProfile()
                ops.conv_profile, ops.stage_profile = prof, sprof
            from aicovergen_amd import dist as adist
            adist.profile_joins = True          # the collectives' own time (device drained around each): this step only
            barrier()
            tp = time.perf_counter()
            _, _, _, prof_split = one_step(args.config, mdxs, vc, hub, net_g, wave44_dev, group, emu)
            barrier()
            prof_ms = (time.perf_counter() - tp) * 1e3
            adist.profile_joins = False
            ops.conv_profile = ops.stage_profile = None
            for k, v in prof_split.items():
                if k.endswith("allgather_s"):
                    split[k] = v
    dts = torch.tensor([dt], dtype=torch.float64, device=device)
    per_rank = [dt]
    have_group = td.is_available() and td.is_initialized()
    if have_group:
        allt = [torch.empty_like(dts) for _ in range(world)]
        td.all_gather(allt, dts)
        per_rank = [float(t.item()) for t in allt]
    dt = max(per_rank)                                   # MAX over ranks
    # where each rank's step went (seconds per step): own MDX windows incl. the stem all-gather, the all-gather alone, plan, HuBERT
    # with the f0 branch underneath, the wait for f0 behind it, the chunk loop, the chunk join, post -- so that a scaling curve says
    # which term grew
    # (mdx_allgather_s, rvc_lengths_allgather_s, rvc_pieces_allgather_s: the collective alone, device drained on both sides --
    #  RCCL time as opposed to waiting for the slowest rank, which is in the stage walls around it; measured in the ONE instrumented
    #  step behind the timed region (dist.profile_joins), the timed steps only count them; `collectives` = how many ran per step)
    split_keys = ["mdx_s", "mdx_allgather_s", "plan_s", "f0_s", "f0_wait_s", "chunks_s", "join_s", "rvc_lengths_allgather_s",
                  "rvc_pieces_allgather_s", "post_s", "collectives"]
    per_rank_split = None
    if have_group:
        mine_split = torch.tensor([split.get(k, 0.0) for k in split_keys], dtype=torch.float64, device=device)
        alls = [torch.empty_like(mine_split) for _ in range(world)]
        td.all_gather(alls, mine_split)
        per_rank_split = [dict(zip(split_keys, [round(float(v), 5) for v in t.cpu().tolist()])) for t in alls]
    if rank == 0:
        ms = dt / args.steps * 1e3
        x = (vc.x_pad, vc.x_query, vc.x_center, vc.x_max) if vc is not None else None
        workload = {
            "C2": "C2: %d s 44.1 kHz stereo -> MDX-Net (%d model(s), Voc_FT-class 3072/256/7680, denoise) only",
            "C3": "C3: %d s 44.1 kHz stereo -> MDX-Net (%d model(s), Voc_FT-class 3072/256/7680, denoise) -> vocals resampled on the "
                  "device to 16 kHz mono -> RVC (HuBERT-base, RMVPE, SynthesizerTrnMs768NSFsid 40k)",
            "C4": "C4: %d s 44.1 kHz stereo -> MDX-Net (%d model(s)) -> device resample -> RVC with mangio-crepe f0 (CREPE-full, hop 128)",
            "C5": "C5: one %d s 44.1 kHz stereo track -> MDX-Net (%d model(s)) -> device resample -> RVC (rmvpe), sharded over the ranks",
        }[args.config] % (int(seconds), args.mdx_models) + ", seeded random weights"
        if emu:
            workload = "TEST HOOK (kernel emulator, miniature networks, host tensors) -- not a measurement: " + workload
        res = {
            "metric": "real-time factor (audio-sec/wall-sec) for MDX+RVC on 4-min 44.1 kHz track",
            "value": seconds * args.steps / dt, "unit": "x real-time", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "bf16x3 (bf16 hi+lo operands, f32 accumulate; f0 / kNN f32)" if split_mode
            else "f16 operands on the matrix pipe for HuBERT's and the synthesizer's conv_g1 / conv_g1w layers (f32 activations, f32 accumulate); MDX-Net, f0, attention, everything else f32" if half_mode else "f32",
            "data": "synthetic",
            "config": {"workload": workload, "config_id": args.config, "mdx_models": args.mdx_models,
                       "audio_seconds_total": seconds,
                       "rvc_preset": None if x is None else "x_pad,x_query,x_center,x_max=%d,%d,%d,%d" % x,
                       "stage_handover": "device (aicg_resample_poly); excluded like in the reference's metric: model load, WAV "
                                         "read/write, main.py mixing",
                       "sharding": "mdx windows + rvc chunks + rmvpe u-net time segments over %d rank(s), all-gather joins; the rmvpe bigru is "
                                   "one recurrence over the track, computed on every rank" % world,
                       "per_rank_seconds_per_step": [t / args.steps for t in per_rank],
                       "per_rank_wall_split_seconds_per_step": per_rank_split,
                       "collectives": None if not have_group else "backend %s, world %d%s" % (
                           td.get_backend(), world, ", one-rank joins forced through the collectives (AICG_FORCE_COLLECTIVES)"
                           if world == 1 else ""),
                       "stage_seconds_per_step": {"hubert": stage[0] / args.steps, "f0": stage[1] / args.steps,
                                                  "synth": stage[2] / args.steps},
                       "wall_split_seconds_per_step": split,
                       "profiled_step_ms": prof_ms},
        }
        if prof is not None:
            conv = prof.summary()
            traffic = None if (split_mode or half_mode) else pmc_traffic_per_launch()   # the committed PMC passes are of the default (fp32) command
            res["roofline"] = {
                # `achieved` is what SURVEY 8(d) defines: every layer's DIRECT-form flops over the family's kernel time.  The Winograd
                # layers execute 4/9 (two-dimensional form) or 2/3 (row form) of those multiply-adds, so `achieved` is an effective
                # rate and may exceed the peak; the fraction of the matrix pipe's roofline is `frac` = executed / peak.
                **({"mixed_precision_note": "--precision f16: the marked layers of the RVC half execute on the fp16 matrix pipe (a quarter of the MFMA "
                    "cycles per multiply-add); `frac` prices ALL executed flops against the fp32 peak and is NOT a pipe utilisation on this line -- "
                    "`dominant_kernel` (conv_w2d, fp32 either way) is"} if half_mode else {}),
                "bound": "mfma", "achieved": conv["tflops"], "peak": MFMA_PEAK, "unit": "TFLOP/s",
                "frac": conv["tflops_executed"] / MFMA_PEAK,
                "frac_algorithmic": conv["tflops"] / MFMA_PEAK,
                "traffic": None if traffic is None else traffic["fetch_x2"],
                "traffic_unit": None if traffic is None else
                "HBM bytes per launch: rocprofv3 2 x FETCH_SIZE + WRITE_SIZE (%s); uncorrected %.4g" % (traffic["source"],
                                                                                                       traffic["raw"]),
                "executed": conv["tflops_executed"],
                "executed_note": "TFLOP/s the matrix pipe was given: the 3x3 TFC layers of MDX-Net run the Winograd F(2x2,3x3) kernel "
                                 "(conv_w2d: 16 instead of 36 multiply-adds per 2x2 output block; AICG_WINOGRAD=1: the F(2,3)-along-rows "
                                 "kernel conv_ws3w, 2/3), the vocoder's k = 3 / 7 / 11 ResBlock layers the one-dimensional F(2,3) kernel "
                                 "(conv_g1w: 4 / 10 / 15 instead of 6 / 14 / 22 per output pair); `achieved` counts every layer's "
                                 "direct-form flops, `frac` = executed / peak",
                "kernel": "conv family (fp32 MFMA: conv_ws3 / conv_ws3m16h implicit GEMM, conv_g1 / conv_g1s LDS-DMA staged 1x1 and stride-2 GEMMs, "
                          "conv_w2d Winograd F(2x2,3x3), conv_g1w one-dimensional Winograd F(2,3))" if not split_mode
                else "conv family (split precision: conv_ws3s on the bf16 MFMA, 3 MFMAs per product -- achieved and peak are "
                     "in fp32-equivalent TFLOP/s, peak = 2516.6 / 3; the f0 models' layers run the fp32 kernels)",
                "measured": "HIP events around every launch of one extra step of the same work, taken right behind the timed "
                            "region (profiled_step_ms; the timed steps carry no events)",
                "launches_per_step": conv["launches"],
                "algorithmic_tflop_per_step": conv["flops"] / 1e12,
                "algorithmic_bytes_per_launch": conv["bytes"] / max(1, conv["launches"]),
                "kernel_ms_per_step": conv["ms"]}
            # the DOMINANT kernel of the step on its own: conv_w2d (Winograd F(2x2,3x3): MDX-Net's 3x3 layers; it executes 16 of the direct
            # form's 36 multiply-adds), from the same live HIP events -- what profiles/rNN_summary.json states from the rocprofv3 trace
            w2 = [r for r in prof.by_shape() if r["shape"].endswith(" wino2")]
            if w2:
                w2_ms, w2_gf, w2_n = sum(r["ms"] for r in w2), sum(r["gflop"] for r in w2), sum(r["launches"] for r in w2)
                res["roofline"]["dominant_kernel"] = {
                    "kernel": "conv_w2d_kernel (Winograd F(2x2,3x3), fp32 MFMA 16x16x4, pair fragments)", "launches_per_step": w2_n,
                    "ms_per_step": w2_ms, "share_of_step": w2_ms / ms, "achieved": w2_gf / w2_ms, "executed": w2_gf / w2_ms * 16.0 / 36.0,
                    "peak": MFMA_PEAK, "unit": "TFLOP/s", "frac": w2_gf / w2_ms * 16.0 / 36.0 / MFMA_PEAK,
                    "note": "achieved = direct-form flops / event time; executed = 16/36 of them; frac = executed / peak"}
            res["stages"] = stage_table(conv, sprof.summary(), 1)
            if args.conv_shapes:
                with open(args.conv_shapes, "w") as f:
                    json.dump(prof.by_shape(), f, indent=0)
        else:
            res["roofline"] = None
        if world == 1 and not args.no_cpu_baseline and not emu:
            with contextlib.redirect_stdout(sys.stderr):
                res["cpu_baseline"] = dict(cpu_baseline(), **reference_cpu_record())
        if args.dump:
            np.savez_compressed(args.dump, sep=sep.cpu().numpy()[:, ::7], out=out if out is not None else np.zeros(1))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if have_group:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
