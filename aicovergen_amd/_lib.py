"""ctypes binding of libaicg_hip.so (C ABI: include/aicg.h).

The product path has exactly one backend: the hipcc-built gfx950 library that sits next to this file.
If it is missing the import of any op raises -- there is no CPU fallback.  (tests/ may point the binding
at tests/emu/libaicg_emu.so, the same kernel sources compiled for a CPU emulator, through
`_use_library_for_tests`; nothing in the package does.)
"""
import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_HEADER = os.path.join(_HERE, "..", "include", "aicg.h")
_DEFAULT = os.path.join(_HERE, "libaicg_hip.so")

_lock = threading.Lock()
_path = None
_lib = None
_backend = None  # "hip" | "emu"

_CTYPES = {
    "const float*": ctypes.c_void_p, "float*": ctypes.c_void_p, "void*": ctypes.c_void_p,
    "const void*": ctypes.c_void_p, "const int*": ctypes.c_void_p, "int*": ctypes.c_void_p,
    "const int64_t*": ctypes.c_void_p, "int64_t*": ctypes.c_void_p, "const double*": ctypes.c_void_p,
    "double*": ctypes.c_void_p, "const int16_t*": ctypes.c_void_p, "int16_t*": ctypes.c_void_p, "uint16_t*": ctypes.c_void_p,
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "double": ctypes.c_double,
    "const aicg_conv_desc*": ctypes.c_void_p,
}


def parse_header(path=_HEADER):
    """Return {function name: [ctypes arg types]} for every `int aicg_*(...)` prototype in aicg.h."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(aicg_\w+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(1), m.group(2).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                # "const float* x" -> "const float*", "int64_t o_sig" -> "int64_t"
                ty = a[: a.rfind("*") + 1] if "*" in a else a.rsplit(" ", 1)[0]
                ty = ty.replace(" *", "*")
                if ty not in _CTYPES:
                    raise RuntimeError("aicg.h: unknown C type %r in %s" % (ty, name))
                types.append(_CTYPES[ty])
        protos[name] = types
    return protos


def _load(path, backend):
    global _lib, _backend
    if not os.path.exists(path):
        raise RuntimeError(
            "aicovergen_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    protos = parse_header()
    for name, argtypes in protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise RuntimeError("aicovergen_amd: %s does not export %s declared in include/aicg.h" % (path, name))
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.aicg_last_error.restype = ctypes.c_char_p
    lib.aicg_last_error.argtypes = []
    lib.aicg_last_launch.restype = ctypes.c_char_p
    lib.aicg_last_launch.argtypes = []
    global _path
    _lib, _backend, _path = lib, backend, path
    return lib


def get():
    with _lock:
        if _lib is None:
            _load(_DEFAULT, "hip")
        return _lib


def backend():
    get()
    return _backend


def get_path():
    """File the bound library was loaded from (tests: development-only kernels exist in libaicg_hip_dev.so, not in the product)."""
    get()
    return _path


def _use_library_for_tests(path, backend="emu"):
    """TEST HOOK: bind to another build of the same C ABI (the CPU kernel emulator)."""
    with _lock:
        _load(path, backend)


def _reset_for_tests():
    global _lib, _backend
    with _lock:
        _lib, _backend = None, None


def last_launch():
    """Name of the kernel family the calling thread's most recent entry point launched (diagnostic; tests assert routing with it)."""
    return (get().aicg_last_launch() or b"").decode()


def call(name, *args):
    lib = get()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.aicg_last_error()
        raise RuntimeError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else "?"))
