"""Every join of the multi-GPU path through the REAL torch.distributed collectives on a box with ONE GPU (VERDICT r4 "missing" #1).

A one-rank process group is created the way bench.py creates the N-rank one (backend "nccl" = RCCL, device_id = this rank's GPU,
127.0.0.1 rendezvous) and `dist.force_collectives` removes the world == 1 shortcuts, so that

  * dist.mdx_separate            (MDX stems: one all_gather of equal (per, 2, gen) blocks),
  * dist.gather_pieces           (RVC chunks: length exchange + padded all_gather),
  * rmvpe.E2E.features_sharded   (RMVPE U-Net time segments: all_gather of (per, 384) blocks, queued on the f0 side stream),
  * crepe.predict                (CREPE posteriors: all_gather of (per, 360) blocks)

each run `td.all_gather` on device tensors.  Asserted: the forced run equals the shortcut run BIT FOR BIT (an all_gather over one rank is
a copy), and the number of collectives that really ran.  Timed: each join's collective at the sizes of the 240 s and 1800 s tracks.
Not a scaling number -- the point is that the first 8-GPU run cannot die in init_process_group or in the first all_gather.

  python tools/rccl_one_rank.py --out profiles/r05_rccl_one_rank.json            (GPU box: nccl on cuda:0)
  python tools/rccl_one_rank.py --emu --backend gloo                            (this container: same code on the kernel emulator)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as td


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--emu", action="store_true", help="kernel emulator on host tensors (CPU container), miniature sizes")
    ap.add_argument("--out", default=None)
    ap.add_argument("--port", type=int, default=29533)
    ap.add_argument("--quick", action="store_true", help="skip the two RMVPE pipeline runs (the CPU suite's budget)")
    args = ap.parse_args()
    import conftest
    conftest._bind("emu" if args.emu else "hip")
    dev = conftest.Dev("emu" if args.emu else "hip")
    device = dev.device
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(args.port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.perf_counter()
    if args.backend == "nccl":
        torch.cuda.set_device(device)
        td.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)     # bench.py's call
    else:
        td.init_process_group(backend=args.backend, rank=0, world_size=1)
    res = {"backend": td.get_backend(), "world_size": td.get_world_size(), "init_process_group_s": round(time.perf_counter() - t0, 3),
           "device": str(device), "torch": torch.__version__, "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    if args.backend == "nccl":
        try:
            res["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:   # noqa: BLE001
            res["rccl_version"] = "unavailable: %s" % e

    from aicovergen_amd import crepe, dist as adist
    from aicovergen_amd.mdx import MDX, MDXModel
    from aicovergen_amd.rmvpe import RMVPE
    from synthetic import weights
    from synthetic.inputs import song_like, vocal_like
    import test_pipeline as tp

    calls = {"n": 0}
    real_all_gather = td.all_gather

    def counting_all_gather(outs, t, group=None, **kw):
        calls["n"] += 1
        if args.backend == "nccl":
            assert t.is_cuda and all(o.is_cuda for o in outs), "a host tensor reached an RCCL collective"
        return real_all_gather(outs, t, group=group, **kw)
    td.all_gather = counting_all_gather

    def sync():
        if not args.emu:
            torch.cuda.synchronize()

    def both(fn):
        """fn() with the one-rank shortcut, then through the collectives -> (shortcut result, forced result, collectives run)."""
        adist.force_collectives = False
        n0 = calls["n"]
        a = fn()
        assert calls["n"] == n0, "the shortcut path ran a collective"
        adist.force_collectives = True
        b = fn()
        adist.force_collectives = False
        return a, b, calls["n"] - n0

    G = td.group.WORLD
    checks = {}
    # ---- MDX stems -------------------------------------------------------------------------------------------------------------
    cfg = weights.MDX_TINY
    sess = MDX(None, MDXModel(device, cfg["dim_f"], cfg["dim_t"], cfg["n_fft"], hop=64), state_dict=weights.mdx_state_dict(cfg, 1234))
    wave = torch.from_numpy(song_like(0.2, 44100, seed=3)[:, :7000]).to(device)
    a, b, n = both(lambda: adist.mdx_separate(sess, wave, True, 2, G).cpu())
    assert torch.equal(a, b) and n == 1
    checks["mdx_separate"] = {"bit_equal": True, "collectives": n}
    # ---- RVC chunks + RMVPE U-Net segments (whole pipeline, progressive f0 schedule, 6 chunks) -----------------------------------
    nets = weights.small_model_set(1234)
    if not args.quick:
        audio = vocal_like(6.3, 16000, 1239)
        a, b, n = both(lambda: tp.run(dev, nets, audio, group=G)[0])
        assert np.array_equal(a, b) and n >= 2, n
        checks["pipeline_rmvpe_6_chunks"] = {"bit_equal": True, "collectives": n}
    # the U-Net join needs a track of at least two context margins: the small RMVPE on a 1 952-frame mel (reach 320)
    r = RMVPE(None, False, device, state_dict=nets["rmvpe_sd"])
    g = torch.Generator().manual_seed(11)
    mel = (torch.randn(1, 128, 1952, generator=g) * 2 - 4).to(device)
    a, b, n = both(lambda: r.model.features_sharded(mel, G).cpu())
    assert torch.equal(a, b) and n == 1 and torch.equal(a, r.model.features(mel).cpu())
    checks["rmvpe_features_sharded"] = {"bit_equal": True, "collectives": n}
    # ... and from pipeline(), on the f0 side stream under the HuBERT pass (20.2 s: 2 048 padded frames, one chunk)
    if not args.quick:
        audio20 = vocal_like(20.2, 16000, 77)
        a, b, n = both(lambda: tp.run(dev, nets, audio20, x=(1, 10, 60, 65), group=G)[0])
        assert np.array_equal(a, b) and n >= 3, n          # features + chunk lengths + chunk pieces
        checks["pipeline_rmvpe_unet_join_on_side_stream"] = {"bit_equal": True, "collectives": n}
    # ---- CREPE posteriors ----------------------------------------------------------------------------------------------------------
    vc, hub, net_g, tgt_sr = tp.build(dev, nets, (1, 1, 1, 2))
    vc.model_crepe = {"full": crepe.Crepe(weights.crepe_state_dict(weights.CREPE_MICRO, 5), device)}
    crepe.DITHER = lambda k: torch.zeros(k)
    audio2 = vocal_like(1.995, 16000, 1240)
    a, b, n = both(lambda: vc.pipeline(hub, net_g, 0, audio2, "x.wav", [0, 0, 0], 0, "mangio-crepe", "", 0.5, 1, 3, tgt_sr, 0, 0.25,
                                       "v2", 0.33, 64, noise_fn=tp.noise_fn_for(nets), group=G))
    assert np.array_equal(a, b) and n >= 3, n          # posteriors + chunk lengths + chunk pieces
    checks["pipeline_mangio_crepe"] = {"bit_equal": True, "collectives": n}
    res["checks"] = checks

    # ---- each join's collective at the sizes of the bench tracks (240 s per rank; one 1800 s track) ------------------------------
    sizes = {"mdx_stems_240s (44, 2, 261120) f32": (44, 2, 261120), "mdx_stems_1800s (314, 2, 261120) f32": (314, 2, 261120),
             "rmvpe_unet_blocks_240s (24608, 384) f32": (24608, 384), "crepe_posteriors_240s (30751, 360) f32": (30751, 360),
             "rvc_pieces_240s (4, 2640000) f32": (4, 2640000)}
    if args.emu:
        sizes = {k: tuple(max(1, d // 64) for d in v) for k, v in sizes.items()}
    adist.force_collectives = True
    timing = {}
    for name, shape in sizes.items():
        x = torch.randn(shape, device=device)
        y = adist.all_gather_equal(x, G)
        assert torch.equal(x, y)
        sync()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            adist.all_gather_equal(x, G)
        sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        timing[name] = {"ms": round(ms, 3), "MB": round(x.numel() * 4 / 1e6, 1), "GB_per_s_incl_concat": round(x.numel() * 4 / ms / 1e6, 1)}
        del x, y
    adist.force_collectives = False
    res["collective_timing_one_rank"] = timing
    res["note"] = ("one rank: the all_gather is a device copy through RCCL's launch path plus torch.cat; the figures bound the fixed cost of "
                   "each join, not xGMI bandwidth")
    res["all_gather_calls_total"] = calls["n"]
    td.destroy_process_group()
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
