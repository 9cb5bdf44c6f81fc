// Error plumbing and version entry points of the C ABI (include/aicg.h).
#include "common.h"

#include <cstdint>
#include <mutex>
#include <unordered_map>

namespace aicg {
void allow_dynamic_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return;
    static std::mutex mu;
    static std::unordered_map<unsigned long long, size_t> granted;  // (device, kernel): the attribute is per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    size_t& g = granted[((unsigned long long)(uintptr_t)kernel) ^ ((unsigned long long)(dev + 1) << 56)];
    if (g >= bytes) return;
    // grant the whole 160 KB of a gfx950 CU at once: one runtime call per kernel for the lifetime of the process
    const size_t want = 160 * 1024;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) == hipSuccess) g = want;
    else (void)hipGetLastError();  // the launch itself reports the failure
}
}  // namespace aicg


namespace aicg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace aicg

namespace aicg {
static thread_local const char* g_last_launch = "";
void note_launch(const char* what) { g_last_launch = what; }
}  // namespace aicg

extern "C" const char* aicg_last_error(void) { return aicg::g_err; }
extern "C" const char* aicg_last_launch(void) { return aicg::g_last_launch; }
extern "C" int aicg_abi_version(void) { return 5; }  // 2: aicg_conv_desc gained shuffle / res_mul; 3: gemm_tile, aicg_last_launch; 4: aicg_rownorm_act_ld; 5: aicg_conv_desc.split == 2 means fp16 operands (before: any nonzero value = the bf16 split)

// Diagnostic: issue-bound fp32 MFMA loop (no memory traffic) to calibrate the attainable v_mfma_f32_32x32x2_f32 rate
// of the device the benchmarks run on (clock under load is power-dependent).  Returns nothing useful in `out` beyond
// keeping the accumulators alive; time it from the host.
namespace aicg {
typedef float pf32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) mfma_probe_kernel(float* out, int iters, float seed) {
    pf32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = seed; a1[r] = seed + 1.f; a2[r] = seed + 2.f; a3[r] = seed + 3.f; }
    float x = seed * (float)(threadIdx.x & 7) + 0.5f, y = 0.25f + seed;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace aicg

extern "C" int aicg_mfma_probe(float* out, int n_blocks, int iters, float seed, void* stream) {
    if (!out) return aicg::fail(AICG_E_ARG, "aicg_mfma_probe: null pointer");
    hipLaunchKernelGGL(aicg::mfma_probe_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, out, iters, seed);
    return aicg::check_launch("mfma_probe_kernel");
}
