"""TEST INFRASTRUCTURE: compile the unmodified csrc/*.hip kernel sources for the host CPU against the
fiber-based HIP stand-in under tests/emu/include, producing tests/emu/libaicg_emu.so.  The product never
loads this library (see aicovergen_amd/_lib.py)."""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "aicovergen_amd", "csrc")
OUT = os.path.join(HERE, "libaicg_emu.so")
OBJ = os.path.join(HERE, "build")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


ASAN_RT = None


def asan_runtime():
    """The AddressSanitizer runtime a python process must LD_PRELOAD to load the sanitized emulator."""
    r = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = r.stdout.strip()
    if not os.path.isabs(path) or not os.path.exists(path):
        raise RuntimeError("no AddressSanitizer runtime next to " + CLANG)
    return path


def build_emu(force=False, sanitize=False):
    """sanitize=True: $AICG_EMU_ASAN_DIR/libaicg_emu_asan.so (default /tmp/aicg_emu_asan), the same sources under -fsanitize=address,undefined: every kernel's
    indexing runs on the host, where an out-of-bounds LDS or global access lands in a red zone -- dynamic LDS is an exact-size
    allocation in that build, global buffers are torch's (malloc, intercepted).  Run with tests/emu/run_sanitized.sh."""
    global OUT, OBJ
    if sanitize:
        old = OUT, OBJ
        # outside the repository (the sanitized objects are ~0.5 GB: they must not travel with gpurun snapshots)
        d = os.environ.get("AICG_EMU_ASAN_DIR", "/tmp/aicg_emu_asan")
        OUT, OBJ = os.path.join(d, "libaicg_emu_asan.so"), os.path.join(d, "build")
        try:
            return _build(force, True)
        finally:
            OUT, OBJ = old
    return _build(force, False)


def _build(force, sanitize):
    if not os.path.exists(CLANG):
        raise RuntimeError("host clang++ not found at " + CLANG)
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(HERE, "emu_rt.cpp")]
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + \
        glob.glob(os.path.join(HERE, "include", "hip", "*.h"))
    flags = ["-x", "c++", "-std=c++17", "-O2", "-g", "-fPIC", "-I", os.path.join(HERE, "include"),
             "-Wno-unused-value", "-Wno-unknown-attributes", "-DAICG_DEV_SWITCHES", "-ffp-contract=fast-honor-pragmas", "-mfma"]  # hipcc's default contraction mode for device code
    san = ["-fsanitize=address,undefined", "-fno-sanitize=vptr,function", "-fno-omit-frame-pointer", "-shared-libsan"] if sanitize else []
    flags = flags + san
    if sanitize:
        flags[flags.index("-O2")] = "-O1"
    jobs, objs = [], []
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_hdr):
            jobs.append([CLANG] + flags + ["-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r.returncode, r.stdout + r.stderr

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        for cmd, rc, log in ex.map(run, jobs):
            if rc != 0:
                raise RuntimeError("emu compile failed: %s\n%s" % (" ".join(cmd), log))
    if jobs or not os.path.exists(OUT):
        cmd = [CLANG, "-shared", "-fPIC", "-o", OUT] + san + objs + ["-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emu link failed:\n" + r.stdout + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build_emu(force="--force" in sys.argv, sanitize="--sanitize" in sys.argv))
