#!/bin/bash
# TEST INFRASTRUCTURE: the kernel suites on the AddressSanitizer + UndefinedBehaviorSanitizer build of the CPU emulator.
# Every kernel's indexing runs on the host: an LDS access past the size the launch requested or a global access past a tensor lands
# in a red zone (dynamic LDS is an exact-size allocation in this build; global buffers are torch's own mallocs, intercepted through
# LD_PRELOAD).  Usage: tests/emu/run_sanitized.sh [pytest arguments]   (default: the conv / FFT / misc kernel suites, CPU only)
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
python tests/emu/build_emu.py --sanitize || exit 1
RT=$(python -c "import sys; sys.path.insert(0, 'tests/emu'); import build_emu; print(build_emu.asan_runtime())") || exit 1
ARGS=("$@")
if [ ${#ARGS[@]} -eq 0 ]; then
  ARGS=(tests/test_conv.py tests/test_conv_g1.py tests/test_conv_g1s.py tests/test_conv_g1w.py tests/test_conv_fuzz.py tests/test_stft.py tests/test_kernels_misc.py)
fi
# detect_leaks=0: the interpreter and torch keep their arenas; detect_stack_use_after_return=0: the emulator's work-items are fibers on
# mmap'ed stacks (announced to the sanitizer in emu_rt.cpp) and need no fake stacks
AICG_EMU_SANITIZE=1 LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0 \
  UBSAN_OPTIONS=print_stacktrace=1 python -m pytest "${ARGS[@]}" -m "not gpu" -q -p no:cacheprovider
