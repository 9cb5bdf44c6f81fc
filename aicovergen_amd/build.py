"""Build libaicg_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libaicg_hip.so")
OBJ_DIR = os.path.join(HERE, "build")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libaicg_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.exists(d) and os.path.getmtime(d) <= t for d in deps)


def _deps(obj, src, hdrs):
    """What `obj` was compiled from: the compiler's own dependency list (obj.d, written by -MD) restricted to this repository, or --
    before the first build -- the source plus every header."""
    dfile = obj + ".d"
    if not os.path.exists(dfile):
        return [src] + hdrs
    text = open(dfile).read().replace("\\\n", " ")
    names = text.split(":", 1)[1].split() if ":" in text else []
    root = os.path.realpath(os.path.join(HERE, ".."))
    mine = [n for n in names if os.path.realpath(n).startswith(root + os.sep)]
    return mine or [src] + hdrs


def build_hip(force=False, verbose=True, dev=False):
    """Compile every csrc/*.hip for gfx950 and link libaicg_hip.so next to this file.
    dev=True builds libaicg_hip_dev.so instead, with -DAICG_DEV_SWITCHES: the environment-controlled A/B switches and the kernel
    generations only they select (csrc/common.h AICG_SWITCH) -- for tools/, bound through _lib._use_library_for_tests; the product
    never loads it."""
    global OUT, OBJ_DIR
    if dev:
        return _with_paths(os.path.join(HERE, "libaicg_hip_dev.so"), os.path.join(HERE, "build_dev"), force, verbose)
    return _build(force, verbose, False)


def _with_paths(out, obj_dir, force, verbose):
    global OUT, OBJ_DIR
    old = OUT, OBJ_DIR
    OUT, OBJ_DIR = out, obj_dir
    try:
        return _build(force, verbose, True)
    finally:
        OUT, OBJ_DIR = old


def _build(force, verbose, dev):
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-constant-logical-operand"] + (["-DAICG_DEV_SWITCHES", "-DAICG_CONV_ABLATION"] if dev else [])
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not _newer(obj, _deps(obj, src, hdrs)):
            jobs.append([hipcc] + flags + ["-MD", "-MF", obj + ".d", "-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r.returncode, r.stdout + r.stderr

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for cmd, rc, log in ex.map(run, jobs):
            if verbose and log.strip():
                print(log, file=sys.stderr)
            if rc != 0:
                raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), log))
    if jobs or force or not os.path.exists(OUT):
        # link inside build/ (clang-offload-bundler drops its per-object temporaries next to the output), then move
        tmp = os.path.join(OBJ_DIR, os.path.basename(OUT))
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ_DIR)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout + r.stderr))
        os.replace(tmp, OUT)
        for f in glob.glob(os.path.join(OBJ_DIR, os.path.basename(OUT) + ".*")) + glob.glob(OUT + ".*"):
            os.remove(f)
    return OUT


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, dev="--dev" in sys.argv))
