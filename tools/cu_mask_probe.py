"""Does a static CU partition between the f0 branch (a chain of ~100 small, latency-bound launches + the BiGRU) and the HuBERT branch (chip-
filling GEMMs) shorten the HuBERT || f0 phase of VC.pipeline?  The f0 side stream / the main stream are created with
hipExtStreamCreateWithCUMask; masks as bit lists over the 256 CUs.  Reports the phase (f0_s) and the whole pipeline per variant."""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench  # noqa: E402
from synthetic.inputs import vocal_like  # noqa: E402
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = [0] * 8
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    arr = (ctypes.c_uint32 * 8)(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


mdxs, vc, hub, net_g = bench.build_models(dev, "C3", 1, tiny=False, preset="fp16")
audio = torch.from_numpy(vocal_like(240.0, 16000, 1234)).to(dev)
ALL = list(range(256))


def run(label, side_bits, main_bits, n=3):
    vc._f0_stream = masked_stream(side_bits) if side_bits is not None else None
    main = masked_stream(main_bits) if main_bits is not None else torch.cuda.current_stream(dev)
    ts = []
    with torch.cuda.stream(main):
        for _ in range(n + 1):
            times = [0, 0, 0]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            vc.pipeline(hub, net_g, 0, audio, "x.wav", times, 0, "rmvpe", "", 0.5, 1, 3, 40000, 0, 0.25, "v2", 0.33, 128, noise_seed=1)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0, dict(vc.last_profile), list(times)))
    t, prof, times = sorted(ts[1:], key=lambda x: x[0])[len(ts[1:]) // 2]
    print(f"{label:58s} pipeline {t*1e3:7.1f} ms  phase {prof['f0_s']*1e3:6.1f}  chunks {prof['chunks_s']*1e3:6.1f}  times[hubert,f0,synth] = "
          + ", ".join("%.1f" % (x * 1e3) for x in times), flush=True)


run("as shipped (priority side stream, no masks)", None, None)
run("f0 on CUs 0..31, main on 32..255", ALL[:32], ALL[32:])
run("f0 on CUs 0..63, main on 64..255", ALL[:64], ALL[64:])
run("f0 on every 8th CU (32), main on the rest", ALL[::8], [b for b in ALL if b % 8])
run("f0 on every 4th CU (64), main on the rest", ALL[::4], [b for b in ALL if b % 4])
run("f0 on every 4th CU (64), main unmasked", ALL[::4], None)
run("f0 unmasked (default priority stream), main on 3 of 4 CUs", None, [b for b in ALL if b % 4])
run("as shipped again", None, None)
