"""In-kernel timeline of the wave-specialised conv kernel (GPU box only).  Builds a private copy of the library with
-DAICG_CONV_TRACE -fgpu-rdc-free (tools/build/libaicg_trace.so), runs one layer and prints, per traced workgroup: prologue
(entry -> first stage ready), K loop, epilogue, in shader cycles, against the MFMA-issue floor of the tile."""
import ctypes, glob, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "build"); os.makedirs(OUT, exist_ok=True)
LIB = os.path.join(OUT, "libaicg_trace.so")
srcs = sorted(glob.glob(os.path.join(ROOT, "aicovergen_amd", "csrc", "*.hip")))
if not os.path.exists(LIB) or "--rebuild" in sys.argv:
    import concurrent.futures
    def cc(src):
        o = os.path.join(OUT, os.path.basename(src) + ".o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fgpu-rdc", "-DAICG_CONV_TRACE", "-c", src, "-o", o])
        return o
    with concurrent.futures.ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(cc, srcs))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-fgpu-rdc", "--hip-link", "-shared", "-fPIC", "-o", LIB] + objs, cwd=OUT)
if "--build-only" in sys.argv:
    sys.exit(0)
from aicovergen_amd import _lib, ops
_lib._use_library_for_tests(LIB, "hip")
lib = _lib.get()
dev = torch.device("cuda:0")
NW, NS = 8192, 8


def run(name, n, ci, co, H, W, k=3, d=1, mfma_per_kstep=None):
    is1d = H == 1
    x = torch.randn(n, ci, W, device=dev) if is1d else torch.randn(n, ci, H, W, device=dev)
    w = torch.randn(co, ci, k) * 0.05 if is1d else torch.randn(co, ci, k, k) * 0.05
    pc = ops.PackedConv(w, torch.zeros(co), padding=(k - 1) * d // 2, dilation=d, device=dev)
    out = torch.empty_like(x) if ci == co else None
    buf = np.zeros(NW * NS, dtype=np.uint64)
    for _ in range(2):
        ops.conv(x, pc, out=out, act=ops.ACT_RELU)
    lib.aicg_conv_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.aicg_conv_trace_read(buf.ctypes.data, 1)
    ops.conv(x, pc, out=out, act=ops.ACT_RELU)
    lib.aicg_conv_trace_read(buf.ctypes.data, 1)
    b2 = np.zeros(64 * 2 * 32 * 4, dtype=np.uint64)
    lib.aicg_conv_trace2_read.argtypes = [ctypes.c_void_p]
    lib.aicg_conv_trace2_read(b2.ctypes.data)
    b2 = b2.reshape(64, 2, 32, 4).astype(np.int64)
    for wg in (0, 8):
        base = b2[wg, 0, 0, 0]
        print(f"  WG {wg} per stage: consumer (barrier arrive, release) | producer (pre-commit, loads landed, barrier arrive, release), cycles since the consumer's first barrier arrival")
        for st in range(min(12, int(buf.reshape(NW, NS)[wg, 4]))):
            c = [int(v - base) for v in b2[wg, 0, st, :2]]
            pr = [int(v - base) for v in b2[wg, 1, st, :4]]
            print(f"     st {st:2d}: cons {c}  prod {pr}")
    t = buf.reshape(NW, NS).astype(np.int64)
    ok = t[:, 3] > 0
    t = t[ok]
    pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    nst = t[:, 4]
    hw = t[:, 5]
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3  # gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    print(f"{name}: {ok.sum()} traced WGs, stages {nst[0]}")
    for nm, v in (("entry->setup", t[:, 6] - t[:, 0]), ("setup->bias done", t[:, 7] - t[:, 6]), ("bias->stage0", t[:, 1] - t[:, 7]),
                  ("prologue", pro), ("k-loop", loop), ("epilogue", epi), ("total", t[:, 3] - t[:, 0])):
        print(f"  {nm:9s} median {np.median(v):9.0f}  p10 {np.percentile(v,10):9.0f}  p90 {np.percentile(v,90):9.0f} cycles")
    span = t[:, 3].max() - t[:, 0].min()
    print(f"  traced span {span} cycles; sum(total)/span = {np.sum(t[:,3]-t[:,0])/span:.1f} WGs in flight (of those traced)")
    # per-CU timelines: s_memtime bases differ per XCD, so cluster by epoch (gaps > 1e8) and HW_ID (se, sh, cu)
    epoch = np.zeros(len(t), dtype=np.int64)
    order = np.argsort(t[:, 0])
    e = 0
    for a, b in zip(order[:-1], order[1:]):
        if t[b, 0] - t[a, 0] > 100000000:
            e += 1
        epoch[b] = e
    key = epoch * 4096 + ((hw >> 8) & 0xff)
    shown = 0
    for k in np.unique(key):
        sel = np.nonzero(key == k)[0]
        if len(sel) < 6:
            continue
        sel = sel[np.argsort(t[sel, 0])]
        base = t[sel[0], 0]
        print(f"  CU key {k}: {len(sel)} traced WGs; (start, loop-start, loop-end, end, wave-slot) relative to the CU's first start:")
        for i in sel[:10]:
            print("     ", [int(v - base) for v in t[i, :4]], int(hw[i] & 0xf))
        shown += 1
        if shown == 2:
            break


if __name__ == "__main__":
    run("mdx_L1_c96 (96x128 tile, 3 MFMA/kstep, 432 ksteps -> floor 82944 cyc)", 16, 96, 96, 128, 1536)
    run("rb_c256_k7 (1-D, 128x128 tile)", 1, 256, 256, 1, 400000, k=7)
    if len(sys.argv) > 1 and sys.argv[1] == "all":
        run("mdx_L2_c144 (160x128 tile)", 16, 144, 144, 64, 768)
        run("rb_c128_k7_d3 (1-D)", 1, 128, 128, 1, 660000, k=7, d=3)
