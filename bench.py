#!/usr/bin/env python
"""Headline benchmark: real-time factor (audio-seconds / wall-seconds) of MDX-Net separation + RVC voice conversion
on a 4-minute 44.1 kHz stereo track per GPU (BASELINE.json metric, config C3 of SURVEY 8d).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one synthetic track of 240 s x N:
  (1) MDX: peak-normalised S44 stereo wave (resident in HBM) -> framed STFT -> TFC-TDF U-Net (Voc_FT-class geometry:
      dim_f 3072, dim_t 256, n_fft 7680, denoise on => 2 network passes per window) -> iSTFT -> window join;
  (2) RVC: S16 mono wave -> VC.pipeline (48 Hz high-pass, cut search, RMVPE f0 on the whole track, per chunk:
      HuBERT-base -> nearest x2 + protect -> SynthesizerTrnMs768NSFsid 40 kHz) -> int16.
At N > 1 the window list (MDX) and the chunk list (RVC) are sharded across ranks and joined by RCCL all-gathers
(weak scaling: per-GPU audio is fixed at 240 s).  All network parameters are seeded random tensors of the real
architectures (no checkpoints exist offline); data is synthetic.  Compute dtype fp32 (MFMA f32).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRACK_S = 240.0


def build_models(device):
    from aicovergen_amd.hubert import HubertModel
    from aicovergen_amd.infer_pack.models import SynthesizerTrnMs768NSFsid
    from aicovergen_amd.mdx import MDX, MDXModel
    from aicovergen_amd.rmvpe import RMVPE
    from aicovergen_amd.rvc import Config
    from aicovergen_amd.vc_infer_pipeline import VC
    from synthetic import weights
    mcfg = weights.MDX_VOC_FT
    model = MDXModel(device, mcfg["dim_f"], mcfg["dim_t"], mcfg["n_fft"], stem_name="Vocals", compensation=1.021)
    mdx = MDX(None, model, state_dict=weights.mdx_state_dict(mcfg, 1234))
    cfg = Config(str(device), True)  # what main.py selects: the "6G" fp16 preset x = (3, 10, 60, 65)
    cfg.device = device
    vc = VC(40000, cfg)
    hub = HubertModel(weights.hubert_state_dict(weights.HUBERT_BASE, 1234), weights.HUBERT_BASE).to(device)
    vc.model_rmvpe = RMVPE(None, False, device, state_dict=weights.rmvpe_state_dict(weights.RMVPE_FULL, 1235))
    net_g = SynthesizerTrnMs768NSFsid(*weights.SYNTH_CFG_40K_V2, is_half=False)
    del net_g.enc_q
    net_g.load_state_dict(weights.synth_state_dict(weights.SYNTH_CFG_40K_V2, 1236), strict=False)
    net_g.eval().to(device)
    return mdx, vc, hub, net_g


def one_step(mdx, vc, hub, net_g, wave44_dev, wave16, group):
    from aicovergen_amd import dist as adist
    t0 = time.perf_counter()
    sep = adist.mdx_separate(mdx, wave44_dev, True, 2, group)
    torch.cuda.synchronize()  # stage boundary (the reference writes the stems to disk here)
    mdx_s = time.perf_counter() - t0
    times = [0, 0, 0]
    out = vc.pipeline(hub, net_g, 0, wave16, "synthetic.wav", times, 0, "rmvpe", "", 0.5, 1, 3, 40000, 0, 0.25, "v2", 0.33,
                      128, group=group, noise_seed=1234)
    return sep, out, times, dict(vc.last_profile, mdx_s=mdx_s)


def cpu_baseline(seconds_rvc=4.0):
    """The oracle (CPU restatement of the reference, "port") timed on this box's host cores on a bounded sample:
    one MDX window pair (denoise => 2 U-Net passes, 5.57 s of audio) + the RVC pipeline on `seconds_rvc` s.
    Threads: min(host cores, 16) -- torch's intra-op pool degrades badly beyond that on these layer sizes (256
    threads measured 100x slower than 16 on the GPU box's 256-core EPYC), so that is what is used and reported."""
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
    a = vocal_like(seconds_rvc, 16000, 2)
    t0 = time.time()
    opipe.vc_pipeline(nets, geo, a, tgt_sr=40000)
    rvc_cost = (time.time() - t0) / seconds_rvc
    return {"value": 1.0 / (mdx_cost + rvc_cost), "unit": "x real-time", "cores": cores, "kind": "port",
            "sample": "1 MDX window pair (5.57 s, denoise) + VC.pipeline (rmvpe) on %.0f s; CPU seconds per audio second: %.2f (MDX) + %.2f (RVC)"
                      % (seconds_rvc, mdx_cost, rvc_cost)}


def pmc_traffic_per_launch():
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes of this same command (profiles/r01_summary.json:
    FETCH_SIZE + WRITE_SIZE, KiB units, summed over the conv kernels, uncorrected -- see DESIGN.md section 5); None when the
    summary is absent.  The counters cannot be collected inside the timed run."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_summary.json")
    try:
        s = json.load(open(path))
        f, w = s["FETCH_SIZE"], s["WRITE_SIZE"]
        return (f["conv_kernels_sum"] / f["conv_kernels_calls"] + w["conv_kernels_sum"] / w["conv_kernels_calls"]) * 1024.0
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--track-seconds", type=float, default=TRACK_S)
    ap.add_argument("--dump", type=str, default=None, help="write the last step's outputs (npz) for cross-checking runs")
    args = ap.parse_args()

    import torch.distributed as td
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if os.environ.get("AICG_FORCE_DEVICE") is not None:   # debugging aid: several ranks on one GPU (with gloo)
        local = int(os.environ["AICG_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("AICG_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            td.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            td.init_process_group(backend=backend, rank=rank, world_size=world)

    from aicovergen_amd import ops
    from synthetic.inputs import song_like, vocal_like
    seconds = args.track_seconds * world
    mdx, vc, hub, net_g = build_models(device)
    wave44 = song_like(seconds, 44100, 1234)
    wave44 = wave44 / max(np.max(wave44), abs(np.min(wave44)))
    wave44_dev = torch.from_numpy(wave44).to(device)     # inputs resident in HBM before the timed region
    wave16 = vocal_like(seconds, 16000, 1234)             # the RVC stage's own input format: host float32 @16 kHz

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(mdx, vc, hub, net_g, wave44_dev, wave16, group)
    prof = None
    if rank == 0:
        prof = ops.ConvProfile()
        ops.conv_profile = prof
    barrier()
    t0 = time.perf_counter()
    stage = [0.0, 0.0, 0.0]
    split = {}
    for _ in range(args.steps):
        sep, out, times, prof_s = one_step(mdx, vc, hub, net_g, wave44_dev, wave16, group)
        stage = [a + b for a, b in zip(stage, times)]
        for k, v in prof_s.items():
            split[k] = split.get(k, 0.0) + v / args.steps
    barrier()
    dt = time.perf_counter() - t0
    ops.conv_profile = None
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        ms = dt / args.steps * 1e3
        conv = prof.summary()
        res = {
            "metric": "real-time factor (audio-sec/wall-sec) for MDX+RVC on 4-min 44.1 kHz track",
            "value": seconds * args.steps / dt, "unit": "x real-time", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3: %d s 44.1 kHz stereo -> MDX-Net (Voc_FT-class 3072/256/7680, denoise) + RVC (HuBERT-base, "
                                   "RMVPE, SynthesizerTrnMs768NSFsid 40k), seeded random weights" % int(seconds),
                       "audio_seconds_per_gpu": args.track_seconds, "rvc_preset": "x_pad,x_query,x_center,x_max=3,10,60,65",
                       "sharding": "mdx windows + rvc chunks over %d rank(s), all-gather join" % world,
                       "stage_seconds_per_step": {"hubert": stage[0] / args.steps, "f0": stage[1] / args.steps,
                                                  "synth": stage[2] / args.steps},
                       "wall_split_seconds_per_step": split},
            "roofline": {"bound": "mfma", "achieved": conv["tflops"], "peak": 157.3, "unit": "TFLOP/s",
                         "frac": conv["tflops"] / 157.3, "traffic": pmc_traffic_per_launch(),
                         "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE + WRITE_SIZE, profiles/r01_summary.json)",
                         "kernel": "conv_ws_kernel family (fp32 MFMA implicit GEMM: conv_ws / conv_ws16 / conv_mfma / conv_mfma16)",
                         "launches_per_step": conv["launches"] / args.steps,
                         "algorithmic_tflop_per_step": conv["flops"] / args.steps / 1e12,
                         "algorithmic_bytes_per_launch": conv["bytes"] / max(1, conv["launches"]),
                         "kernel_ms_per_step": conv["ms"] / args.steps},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        if args.dump:
            np.savez_compressed(args.dump, sep=sep.cpu().numpy()[:, ::7], out=out)
        print(json.dumps(res))
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
