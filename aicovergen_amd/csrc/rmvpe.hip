// RMVPE-specific kernels (reference src/rmvpe.py): log-mel helpers, 2x2 average pooling, the bidirectional GRU
// recurrence, and the salience -> f0 decode; plus the mel-scale coarse-pitch quantiser of VC.get_f0
// (src/vc_infer_pipeline.py:361-368).  The convolutions of the DeepUnet run through conv.hip.
#include "common.h"

#include <cstdlib>

namespace aicg {

// |re + i im| elementwise (rmvpe.py:314: magnitude = sqrt(real^2 + imag^2))
__global__ void __launch_bounds__(256) cabs_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                                   float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = sqrtf(re[i] * re[i] + im[i] * im[i]);
}

// y[n,c,:] = act(x[n,c,:] * scale[c] + shift[c])   (eval-mode BatchNorm2d on the network input, rmvpe.py:92)
__global__ void __launch_bounds__(256) channel_affine_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ out,
                                                             int C, long HW, long total, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % C);
        out[i] = apply_act(x[i] * scale[c] + shift[c], act, 0.f);
    }
}

// AvgPool2d(kernel 2x2, stride 2) (rmvpe.py:111); x may be a channel slice of a larger buffer
__global__ void __launch_bounds__(256) avgpool2x2_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int C,
                                                         int Ho, int Wo, long x_sn, long x_sc, long x_sh) {
    const long total = (long)N * C * Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int wo = (int)(i % Wo);
        long t = i / Wo;
        const int ho = (int)(t % Ho);
        t /= Ho;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        const float* p = x + (long)n * x_sn + (long)c * x_sc + (long)(2 * ho) * x_sh + 2 * wo;
        out[i] = (p[0] + p[1] + p[x_sh] + p[x_sh + 1]) * 0.25f;
    }
}

// ---- bidirectional GRU recurrence (torch.nn.GRU, 1 layer; rmvpe.py:11-20) --------------------------------------
// gi: (2*3*Hd, T) = W_ih x + b_ih for [forward r,z,n ; reverse r,z,n], channel-major (computed by the conv kernel)
// whh_t: (2, Hd, 3*Hd) = W_hh transposed per direction (k-major, so lane j reads unit-stride)
// out: (2*Hd, T).  One workgroup per direction, one thread per gate row; the T steps are strictly sequential, so
// the cost per step is the cost of reading W_hh (3*Hd x Hd fp32 = 786 KB for Hd = 256, more than one CU's register
// file): each thread keeps the first KR columns of its row in registers, the next KL columns of all rows sit in
// LDS, and only the remaining Hd - KR - KL columns stream from L2 every step.
//   r = s(gi_r + W_hr h + b_hr); z = s(gi_z + W_hz h + b_hz); n = tanh(gi_n + r*(W_hn h + b_hn)); h = (1-z)*n + z*h
template <int HD, int KR, int KL>
__global__ void __launch_bounds__(3 * HD / 2) gru_kernel(const float* __restrict__ gi, const float* __restrict__ whh_t,
                                                         const float* __restrict__ bhh, float* __restrict__ out, long T, long s0, long s1,
                                                         float* hstate) {
    // steps s0 .. s1 - 1 of the recurrence (forward: frame s, backward: frame T - 1 - s); hstate (2, HD): the hidden state a later
    // segment resumes from -- read when s0 > 0, written at the end.  A recurrence cut into segments this way is the same arithmetic
    // step by step: bit-identical to one launch over 0 .. T.
    // 3*HD/2 threads, two gate rows each (j and j + 3*HD/2): 6 waves for HD = 256 leave each wave a 256-register
    // budget, of which 2*KR hold weights.
    constexpr int NT = 3 * HD / 2;
    HIP_DYNAMIC_SHARED(float, smem)
    float* h = smem;               // HD
    float* gh = smem + HD;         // 3*HD
    float* wl = smem + 4 * HD;     // KL x 3*HD
    const int dir = blockIdx.x;
    const int j0 = threadIdx.x, j1 = threadIdx.x + NT;  // gate rows of this thread
    const float* W = whh_t + (long)dir * HD * 3 * HD;
    const float b0 = bhh[dir * 3 * HD + j0], b1 = bhh[dir * 3 * HD + j1];
    const float* gid = gi + (long)dir * 3 * HD * T;
    float* od = out + (long)dir * HD * T;
    float w0[KR], w1[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        w0[k] = W[(long)k * 3 * HD + j0];
        w1[k] = W[(long)k * 3 * HD + j1];
    }
    for (int k = 0; k < KL; ++k) {
        wl[k * 3 * HD + j0] = W[(long)(KR + k) * 3 * HD + j0];
        wl[k * 3 * HD + j1] = W[(long)(KR + k) * 3 * HD + j1];
    }
    if (j0 < HD) h[j0] = s0 > 0 ? hstate[dir * HD + j0] : 0.f;
    __syncthreads();
    for (long s = s0; s < s1; ++s) {
        const long t = dir == 0 ? s : T - 1 - s;
        // the input projections of this step are independent of h: issue their loads before the dot products
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (j0 < HD) {
            g0 = gid[(long)j0 * T + t];
            g1 = gid[(long)(HD + j0) * T + t];
            g2 = gid[(long)(2 * HD + j0) * T + t];
        }
        float a0 = b0, a1 = b1;
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const float hk = h[k];
            a0 = fmaf(w0[k], hk, a0);
            a1 = fmaf(w1[k], hk, a1);
        }
        const float* wlp = wl;
        asm volatile("" : "+v"(wlp));  // keep LICM from hoisting the LDS-resident weights into registers
#pragma unroll 4
        for (int k = 0; k < KL; ++k) {
            const float hk = h[KR + k];
            a0 = fmaf(wlp[k * 3 * HD + j0], hk, a0);
            a1 = fmaf(wlp[k * 3 * HD + j1], hk, a1);
        }
#pragma unroll 4
        for (int k = KR + KL; k < HD; ++k) {
            const float hk = h[k];
            a0 = fmaf(W[(long)k * 3 * HD + j0], hk, a0);
            a1 = fmaf(W[(long)k * 3 * HD + j1], hk, a1);
        }
        gh[j0] = a0;
        gh[j1] = a1;
        __syncthreads();
        if (j0 < HD) {
            const float r = 1.f / (1.f + expf(-(g0 + gh[j0])));
            const float z = 1.f / (1.f + expf(-(g1 + gh[HD + j0])));
            const float n = tanhf(g2 + r * gh[2 * HD + j0]);
            const float hn = (1.f - z) * n + z * h[j0];
            h[j0] = hn;
            od[(long)j0 * T + t] = hn;
        }
        __syncthreads();
    }
    if (hstate && j0 < HD) hstate[dir * HD + j0] = h[j0];
}

// ---- two-workgroup GRU: all of W_hh on chip ------------------------------------------------------------------------
// Each direction is split over two workgroups (hidden units [0, HD/2) and [HD/2, HD)); a workgroup keeps the 3*HD/2
// gate rows of its units entirely on chip (KR columns of each row in registers, HD-KR in LDS), so a step costs one
// dot product plus one exchange of HD/2 new h values with the partner.  The exchange follows the tagged-granule
// recipe of the CDNA guide (G16 / R2): one 8-byte {tag = step+1, value} agent-scope store per unit, polled with
// relaxed agent-scope loads; two parity slots because the partner may run one step ahead.  The four workgroups of a
// launch are co-resident by construction (4 working blocks on a 256-CU device); spins are bounded and report through `err`.
template <int HD, int KR>
__global__ void __launch_bounds__(3 * HD / 2) gru2_kernel(const float* __restrict__ gi, const float* __restrict__ whh_t,
                                                          const float* __restrict__ bhh, float* __restrict__ out, long T,
                                                          unsigned long long* xbuf, int* err, int force_agent) {
    constexpr int HH = HD / 2;      // units per workgroup
    constexpr int NT = 3 * HH;      // threads = gate rows per workgroup
    constexpr int KL = HD - KR;
    constexpr int PW = (HH + 63) / 64 * 64;  // first thread of the polling group: wave-aligned
    static_assert(PW + HH <= NT, "polling group must fit the workgroup");
    HIP_DYNAMIC_SHARED(float, smem)
    float* h = smem;                // HD
    float* gh = smem + HD;          // NT
    float* wl = smem + HD + NT;     // KL x NT
    // 32 blocks are launched and 4 work (0, 8, 16, 24): blocks b and b + 8 are observed to land on the same XCD (b % 8), so the
    // two halves of a direction can hand h over through that XCD's L2.  (Partners are adjacent in dispatch order among the
    // working blocks, so two host threads suffice to run the pair concurrently in the CPU emulator.)  Placement is not a contract: the pair compares HW_REG_XCC_ID at
    // start and only then publishes with plain stores (which keep the line in the local L2; sc1 / atomic stores drop it and
    // the partner reads at the cross-XCD rate -- MI355X_MICROARCH "stores of each flavour"); otherwise agent-scope stores.
    if (blockIdx.x & 7) return;
    const int dir = blockIdx.x >> 4, part = (blockIdx.x >> 3) & 1;
    const int t_ = threadIdx.x;
    __shared__ int same_xcd_s;
    if (t_ == 0) {
        int* xcc = err + 4;  // 4 ints inside the zeroed 64-byte tail of the scratch buffer
        const int me = (int)(__builtin_amdgcn_s_getreg(6164) & 0xF) + 1;  // HW_REG_XCC_ID (id 20), bits [3:0]
        __hip_atomic_store(xcc + dir * 2 + part, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int other = 0, spins = 0;
        while ((other = __hip_atomic_load(xcc + dir * 2 + (1 - part), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
            if (++spins > (1 << 22)) { *err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        same_xcd_s = (other == me) && !force_agent;
    }
    __syncthreads();
    const bool same_xcd = same_xcd_s != 0;
    const int gate = t_ / HH, u = t_ - gate * HH;
    const int row = gate * HD + part * HH + u;          // row of W_hh (3*HD x HD), [r ; z ; n]
    const float* W = whh_t + (long)dir * HD * 3 * HD;   // k-major: W[k * 3*HD + row]
    const float brow = bhh[dir * 3 * HD + row];
    const float* gid = gi + (long)dir * 3 * HD * T;
    float* od = out + (long)dir * HD * T;
    unsigned long long* mine = xbuf + ((long)(dir * 2 + part) * 2) * HH;        // [parity][HH]
    unsigned long long* theirs = xbuf + ((long)(dir * 2 + (1 - part)) * 2) * HH;
    float wr[KR];
#pragma unroll
    for (int kb = 0; kb < KR; kb += 16) {
#pragma unroll
        for (int q = 0; q < 16; ++q) wr[kb + q] = W[(long)(kb + q) * 3 * HD + row];
        asm volatile("" ::: "memory");  // 16 loads (and their 64-bit addresses) in flight at a time, not KR
    }
#pragma unroll 4
    for (int k = 0; k < KL; ++k) wl[k * NT + t_] = W[(long)(KR + k) * 3 * HD + row];
    if (t_ < HD) h[t_] = 0.f;
    __syncthreads();
    const int my_unit = part * HH + u;                  // valid for t_ < HH
    for (long s = 0; s < T; ++s) {
        const long t = dir == 0 ? s : T - 1 - s;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (t_ < HH) {
            g0 = gid[(long)my_unit * T + t];
            g1 = gid[(long)(HD + my_unit) * T + t];
            g2 = gid[(long)(2 * HD + my_unit) * T + t];
        }
        float a0 = brow;
#pragma unroll
        for (int kb = 0; kb < KR; kb += 16) {
            // h is read 16 values (4 x ds_read_b128 broadcasts) at a time; the compiler barrier keeps it from hoisting
            // all KR reads above the FMAs, which would evict the weights from the register file
            float hv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) hv[q] = h[kb + q];
            float c0 = 0.f, c1 = 0.f;  // two short chains per chunk, folded before the next chunk: the scheduler cannot
#pragma unroll                             // postpone one chain (and keep its 8 h values per chunk alive) past the loop
            for (int q = 0; q < 16; q += 2) {
                c0 = fmaf(wr[kb + q], hv[q], c0);
                c1 = fmaf(wr[kb + q + 1], hv[q + 1], c1);
            }
            a0 += c0 + c1;
            asm volatile("" : "+v"(a0));
        }
        // opaque copy of the LDS weight pointer: without it LICM hoists these loop-invariant LDS reads out of the
        // step loop into registers and the register-resident part of the row gets spilled to scratch instead
        const float* wlp = wl + t_;
        asm volatile("" : "+v"(wlp));
#pragma unroll 4
        for (int k = 0; k < KL; ++k) a0 = fmaf(wlp[k * NT], h[KR + k], a0);
        gh[t_] = a0;
        __syncthreads();
        const int par = (int)(s & 1);
        if (t_ < HH) {
            const float r = 1.f / (1.f + expf(-(g0 + gh[u])));
            const float z = 1.f / (1.f + expf(-(g1 + gh[HH + u])));
            const float n = tanhf(g2 + r * gh[2 * HH + u]);
            const float hn = (1.f - z) * n + z * h[my_unit];
            h[my_unit] = hn;
            od[(long)my_unit * T + t] = hn;
            const unsigned long long gran = ((unsigned long long)(unsigned)(s + 1) << 32) | (unsigned)__float_as_int(hn);
            if (same_xcd) mine[par * HH + u] = gran;  // one 8-byte store: the poller's sc1 load finds it in the shared L2
            else __hip_atomic_store(mine + par * HH + u, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (t_ >= PW && t_ < PW + HH) {
            // a different WAVE than the publishers (a wave that polled first while its own lanes still had to publish
            // would deadlock against the partner doing the same): fetch the partner's new h values for this step
            const int v = t_ - PW;
            unsigned long long gran = 0;
            const unsigned want = (unsigned)(s + 1);
            int spins = 0;
            for (;;) {
                gran = __hip_atomic_load(theirs + par * HH + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(gran >> 32) == want) break;
                if (++spins > (1 << 22)) { *err = 1; break; }  // bounded: never hang the device
                __builtin_amdgcn_s_sleep(1);
            }
            h[(1 - part) * HH + v] = __int_as_float((int)(unsigned)gran);
        }
        __syncthreads();
        if (*((volatile int*)err)) return;
    }
}

// ---- four-workgroup GRU (round 3): the exchange, not the dot product, bounds a step --------------------------------------------------
// r2 measured 2.9 us per step for the two-workgroup form: ~0.5 us of FMAs, ~1 us for the partner hand-over through L2, ~0.3 us for
// a volatile error-flag load on the critical path of every step, the rest barriers and gate math.  Here a direction is split over
// FOUR workgroups (HD / 4 units each): the 3 HD / 4 gate rows of a workgroup are held ENTIRELY in registers (two threads per row, HD / 2
// columns = 128 registers each, no LDS-resident weights), a step's dot product is half as long, the three partners' quarters arrive
// concurrently (one poll round, three waves), and the error flag is looked at every 32nd step.  Same tagged-granule protocol,
// same bounded spins, same XCD test for the plain-store fast path.  Working blocks: hardware launches 32 and uses b with
// (b & 7) < 2 -- direction b & 7, part b >> 3: the four parts of a direction are b, b + 8, b + 16, b + 24, observed on one XCD; the
// emulator launches 8 (direction b >> 2, part b & 3) so that a quad is contiguous in its dispatch order.
template <int HD>
__global__ void __launch_bounds__(3 * HD / 2) gru4_kernel(const float* __restrict__ gi, const float* __restrict__ whh_t,
                                                          const float* __restrict__ bhh, float* __restrict__ out, long T,
                                                          unsigned long long* xbuf, int* err, int force_agent, long s0, long s1,
                                                          float* hstate) {
    // steps s0 .. s1 - 1 (see gru_kernel); the granule tags stay the GLOBAL step number + 1, so the exchange buffer of the previous
    // segment (tags <= s0) can never satisfy a poll of this one
    constexpr int HQ = HD / 4;      // units per workgroup
    constexpr int NR = 3 * HQ;      // gate rows per workgroup
    constexpr int NT = 2 * NR;      // threads: two per row
    constexpr int KH = HD / 2;      // columns per thread
    static_assert(HQ % 64 == 0 && HQ + 3 * HQ <= NT, "publishers = wave 0 .. HQ / 64 - 1, pollers = the following 3 HQ threads");
    __shared__ __attribute__((aligned(16))) float h[HD];
    __shared__ float gh[NT];
    __shared__ int same_xcd_s;
#ifdef AICG_EMULATED
    if (blockIdx.x >= 8) return;
    const int dir = blockIdx.x >> 2, part = blockIdx.x & 3;
#else
    if ((blockIdx.x & 7) >= 2) return;
    const int dir = blockIdx.x & 7, part = blockIdx.x >> 3;
#endif
    const int t_ = threadIdx.x;
    if (t_ == 0) {
        int* xcc = err + 4;  // 8 ints inside the zeroed 64-byte tail of the scratch buffer
        const int me = (int)(__builtin_amdgcn_s_getreg(6164) & 0xF) + 1;  // HW_REG_XCC_ID (id 20), bits [3:0]
        __hip_atomic_store(xcc + dir * 4 + part, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        for (int q = 0; q < 4; ++q) {
            int other = 0, spins = 0;
            while ((other = __hip_atomic_load(xcc + dir * 4 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                if (++spins > (1 << 22)) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            same = same && other == me;
        }
        same_xcd_s = same && !force_agent;
    }
    __syncthreads();
    const bool same_xcd = same_xcd_s != 0;
    const int rl = t_ % NR, kh = t_ / NR;               // row of this workgroup, column half
    const int gate = rl / HQ, u = rl - gate * HQ;
    const int row = gate * HD + part * HQ + u;          // row of W_hh (3*HD x HD), [r ; z ; n]
    const float* W = whh_t + (long)dir * HD * 3 * HD;   // k-major: W[k * 3*HD + row]
    const float brow = kh == 0 ? bhh[dir * 3 * HD + row] : 0.f;
    const float* gid = gi + (long)dir * 3 * HD * T;
    float* od = out + (long)dir * HD * T;
    unsigned long long* mine = xbuf + ((long)(dir * 4 + part) * 2) * HQ;        // [parity][HQ]
    float wr[KH];
#pragma unroll
    for (int kb = 0; kb < KH; kb += 16) {
#pragma unroll
        for (int q = 0; q < 16; ++q) wr[kb + q] = W[(long)(kh * KH + kb + q) * 3 * HD + row];
        asm volatile("" ::: "memory");  // 16 loads (and their 64-bit addresses) in flight at a time, not KH
    }
    if (t_ < HD) h[t_] = s0 > 0 ? hstate[dir * HD + t_] : 0.f;
    __syncthreads();
    const int my_unit = part * HQ + u;                  // valid for t_ < HQ (gate 0, column half 0)
    const float* hk = h + kh * KH;
    for (long s = s0; s < s1; ++s) {
        const long t = dir == 0 ? s : T - 1 - s;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (t_ < HQ) {
            g0 = gid[(long)my_unit * T + t];
            g1 = gid[(long)(HD + my_unit) * T + t];
            g2 = gid[(long)(2 * HD + my_unit) * T + t];
        }
        float a0 = brow;
#pragma unroll
        for (int kb = 0; kb < KH; kb += 16) {
            float hv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) hv[q] = hk[kb + q];
            float c0 = 0.f, c1 = 0.f;
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                c0 = fmaf(wr[kb + q], hv[q], c0);
                c1 = fmaf(wr[kb + q + 1], hv[q + 1], c1);
            }
            a0 += c0 + c1;
            asm volatile("" : "+v"(a0));
        }
        gh[t_] = a0;
        __syncthreads();
        const int par = (int)(s & 1);
        if (t_ < HQ) {
            const float r = 1.f / (1.f + expf(-(g0 + (gh[u] + gh[NR + u]))));
            const float z = 1.f / (1.f + expf(-(g1 + (gh[HQ + u] + gh[NR + HQ + u]))));
            const float n = tanhf(g2 + r * (gh[2 * HQ + u] + gh[NR + 2 * HQ + u]));
            const float hn = (1.f - z) * n + z * h[my_unit];
            h[my_unit] = hn;
            od[(long)my_unit * T + t] = hn;
            const unsigned long long gran = ((unsigned long long)(unsigned)(s + 1) << 32) | (unsigned)__float_as_int(hn);
            if (same_xcd) mine[par * HQ + u] = gran;
            else __hip_atomic_store(mine + par * HQ + u, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (t_ < 4 * HQ) {
            // other WAVES than the publishers (see gru2_kernel): one lane per value of the three partners' quarters
            const int v = t_ - HQ;
            const int pp = (part + 1 + v / HQ) & 3, pu = v % HQ;
            const unsigned long long* theirs = xbuf + ((long)(dir * 4 + pp) * 2) * HQ;
            unsigned long long gran = 0;
            const unsigned want = (unsigned)(s + 1);
            int spins = 0;
            for (;;) {
                gran = __hip_atomic_load(theirs + par * HQ + pu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(gran >> 32) == want) break;
                if (++spins > (1 << 22)) { *err = 1; break; }  // bounded: never hang the device
                __builtin_amdgcn_s_sleep(1);
            }
            h[pp * HQ + pu] = __int_as_float((int)(unsigned)gran);
        }
        __syncthreads();
        if ((s & 31) == 31 && *((volatile int*)err)) return;   // a timed-out partner: everybody leaves within 32 steps
    }
    if (hstate && t_ < HQ) hstate[dir * HD + my_unit] = h[my_unit];   // this workgroup's quarter of the state
}

// ---- salience decode: RMVPE.to_local_average_cents + decode (rmvpe.py:359-364, 385-409) -----------------------
// salience: (T, NB=360) row-major fp32.  One wave per frame: argmax by xor-shuffles (first maximum wins, like
// np.argmax), then lane 0 forms the 9-bin float64 local average in numpy's pairwise order for n = 9:
// ((a0+a1)+(a2+a3)) + ((a4+a5)+(a6+a7)) + a8  (np.sum over the contiguous last axis).
__global__ void __launch_bounds__(256) salience_decode_kernel(const float* __restrict__ sal, double* __restrict__ cents,
                                                              double* __restrict__ f0, int* __restrict__ center, long T,
                                                              int NB, float thred) {
    const int lane = threadIdx.x & 63;
    const long frame = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= T) return;
    const float* s = sal + frame * NB;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int k = lane; k < NB; k += 64) {
        const float v = s[k];
        if (v > best || (v == best && k < bidx)) { best = v; bidx = k; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bidx, o);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane == 0) {
#pragma clang fp contract(off)  // numpy rounds every product and sum separately: no fma fusion here
        // np.sum(todo_salience * todo_cents_mapping, 1) is float64 (fp32 x fp64 products); np.sum(todo_salience, 1)
        // stays float32; the division promotes the fp32 weight sum to float64.
        double prod[9];
        float wgt[9];
        for (int i = 0; i < 9; ++i) {
            const int k = bidx - 4 + i;  // index into the un-padded salience; outside -> padded zeros
            const bool in = k >= 0 && k < NB;
            const float sv = in ? s[k] : 0.f;
            const double cm = in ? (20.0 * (double)k + 1997.3794084376191) : 0.0;  // cents_mapping, zero in the pad
            prod[i] = (double)sv * cm;
            wgt[i] = sv;
        }
        const double ps = (((prod[0] + prod[1]) + (prod[2] + prod[3])) + ((prod[4] + prod[5]) + (prod[6] + prod[7]))) + prod[8];
        const float ws = (((wgt[0] + wgt[1]) + (wgt[2] + wgt[3])) + ((wgt[4] + wgt[5]) + (wgt[6] + wgt[7]))) + wgt[8];
        double c = ps / (double)ws;
        if (best <= thred) c = 0.0;  // devided[maxx <= thred] = 0
        cents[frame] = c;
        double f = 10.0 * pow(2.0, c / 1200.0);  // numpy: 10 * (2 ** (cents_pred / 1200))
        if (f == 10.0) f = 0.0;  // f0[f0 == 10] = 0
        f0[frame] = f;
        if (center) center[frame] = bidx;
    }
}

// ---- VC.get_f0 tail (vc_infer_pipeline.py:346,361-368): f0 *= 2^(key/12); mel quantise to 1..255, rint (half-even)
__global__ void __launch_bounds__(256) f0_coarse_kernel(const double* __restrict__ f0_in, double factor, double* __restrict__ f0_out,
                                                        long* __restrict__ coarse, long n, double mel_min, double mel_max) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)  // bit-exact with numpy's separately rounded float64 ops
        const double f = f0_in[i] * factor;
        f0_out[i] = f;
        double mel = 1127.0 * log(1.0 + f / 700.0);
        if (mel > 0.0) mel = (mel - mel_min) * 254.0 / (mel_max - mel_min) + 1.0;
        if (mel <= 1.0) mel = 1.0;
        if (mel > 255.0) mel = 255.0;
        coarse[i] = (long)rint(mel);
    }
}

static unsigned ew_grid2(long total) { return (unsigned)lmax(1, lmin((total + 255) / 256, 256L * 16)); }

}  // namespace aicg

using namespace aicg;

extern "C" int aicg_complex_abs(const float* re, const float* im, float* out, int64_t n, void* stream) {
    if (!re || !im || !out) return fail(AICG_E_ARG, "aicg_complex_abs: null pointer");
    if (n == 0) return AICG_OK;
    hipLaunchKernelGGL(cabs_kernel, dim3(ew_grid2(n)), dim3(256), 0, (hipStream_t)stream, re, im, out, (long)n);
    return check_launch("cabs_kernel");
}

extern "C" int aicg_channel_affine(const float* x, const float* scale, const float* shift, float* out, int N, int C,
                                   int64_t HW, int act, void* stream) {
    if (!x || !scale || !shift || !out) return fail(AICG_E_ARG, "aicg_channel_affine: null pointer");
    const long total = (long)N * C * HW;
    if (total == 0) return AICG_OK;
    hipLaunchKernelGGL(channel_affine_kernel, dim3(ew_grid2(total)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, out, C,
                       (long)HW, total, act);
    return check_launch("channel_affine_kernel");
}

extern "C" int aicg_avgpool2x2(const float* x, float* out, int N, int C, int H, int W, int64_t x_sn, int64_t x_sc,
                               int64_t x_sh, void* stream) {
    if (!x || !out) return fail(AICG_E_ARG, "aicg_avgpool2x2: null pointer");
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * C * Ho * Wo;
    if (total == 0) return AICG_OK;
    hipLaunchKernelGGL(avgpool2x2_kernel, dim3(ew_grid2(total)), dim3(256), 0, (hipStream_t)stream, x, out, N, C, Ho, Wo,
                       (long)x_sn, (long)x_sc, (long)x_sh);
    return check_launch("avgpool2x2_kernel");
}

extern "C" int aicg_gru_bidir_seg(const float* gi, const float* whh_t, const float* bhh, float* out, int hidden, int64_t T,
                                  int64_t s_begin, int64_t s_end, float* h_state, void* stream) {
    if (!gi || !whh_t || !bhh || !out) return fail(AICG_E_ARG, "aicg_gru_bidir: null pointer");
    if (s_begin < 0 || s_end > T || s_begin > s_end || (s_begin > 0 && !h_state))
        return fail(AICG_E_ARG, "aicg_gru_bidir_seg: steps %ld .. %ld of %ld (a segment that does not start at 0 needs h_state)", (long)s_begin,
                    (long)s_end, (long)T);
    if (T == 0 || s_begin == s_end) return AICG_OK;
    if (hidden == 256) {
        constexpr int KR = 96, KL = 48;
        const size_t lds = (size_t)(4 * 256 + KL * 768) * sizeof(float);
        auto kern = gru_kernel<256, KR, KL>;
        allow_dynamic_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, dim3(2), dim3(384), lds, (hipStream_t)stream, gi, whh_t, bhh, out, (long)T, (long)s_begin, (long)s_end, h_state);
    } else if (hidden == 64) {
        constexpr int KR = 32, KL = 16;
        const size_t lds = (size_t)(4 * 64 + KL * 192) * sizeof(float);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gru_kernel<64, KR, KL>), dim3(2), dim3(96), lds, (hipStream_t)stream, gi, whh_t, bhh,
                           out, (long)T, (long)s_begin, (long)s_end, h_state);
    } else {
        return fail(AICG_E_SHAPE, "aicg_gru_bidir: hidden size %d not instantiated (256, 64)", hidden);
    }
    return check_launch("gru_kernel");
}

extern "C" int aicg_gru_bidir(const float* gi, const float* whh_t, const float* bhh, float* out, int hidden, int64_t T,
                              void* stream) {
    return aicg_gru_bidir_seg(gi, whh_t, bhh, out, hidden, T, 0, T, nullptr, stream);
}

extern "C" int aicg_gru_bidir_2wg(const float* gi, const float* whh_t, const float* bhh, float* out, int hidden, int64_t T,
                                  void* xchg_scratch, void* stream) {
    // xchg_scratch: 2 dirs x 2 parts x 2 parities x hidden/2 granules of 8 bytes (= 32*hidden bytes) + an int error word
    if (!gi || !whh_t || !bhh || !out || !xchg_scratch) return fail(AICG_E_ARG, "aicg_gru_bidir_2wg: null pointer");
    if (T == 0) return AICG_OK;
    const size_t xbytes = (size_t)2 * 2 * 2 * (hidden / 2) * 8;
    (void)hipMemsetAsync(xchg_scratch, 0, xbytes + 64, (hipStream_t)stream);
    unsigned long long* xb = (unsigned long long*)xchg_scratch;
    int* err = (int*)((char*)xchg_scratch + xbytes);
    AICG_SWITCH(force_agent, "AICG_GRU_AGENT_STORES", 0);  // A/B switch
    if (hidden == 256) {
        constexpr int KR = 208;
        const size_t lds = (size_t)(256 + 384 + (256 - KR) * 384) * sizeof(float);
        auto kern = gru2_kernel<256, KR>;
        allow_dynamic_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, dim3(32), dim3(384), lds, (hipStream_t)stream, gi, whh_t, bhh, out, (long)T, xb, err, (int)force_agent);
    } else if (hidden == 64) {
        constexpr int KR = 48;
        const size_t lds = (size_t)(64 + 96 + (64 - KR) * 96) * sizeof(float);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gru2_kernel<64, KR>), dim3(32), dim3(96), lds, (hipStream_t)stream, gi, whh_t, bhh, out,
                           (long)T, xb, err, force_agent);
    } else {
        return fail(AICG_E_SHAPE, "aicg_gru_bidir_2wg: hidden size %d not instantiated (256, 64)", hidden);
    }
    return check_launch("gru2_kernel");
}

extern "C" int aicg_gru_bidir_4wg_seg(const float* gi, const float* whh_t, const float* bhh, float* out, int hidden, int64_t T,
                                      int64_t s_begin, int64_t s_end, float* h_state, void* xchg_scratch, void* stream) {
    // xchg_scratch: 2 dirs x 4 parts x 2 parities x hidden/4 granules of 8 bytes (= 32*hidden bytes) + 64 zeroed bytes (error word,
    // XCC ids) -- the same size as aicg_gru_bidir_2wg's.  The segments of one recurrence share it: the first (s_begin == 0) clears it,
    // the following ones only the XCC ids (every launch re-establishes where its workgroups run); the error word accumulates.
    if (!gi || !whh_t || !bhh || !out || !xchg_scratch) return fail(AICG_E_ARG, "aicg_gru_bidir_4wg: null pointer");
    if (hidden != 256) return fail(AICG_E_SHAPE, "aicg_gru_bidir_4wg: hidden size %d not instantiated (256)", hidden);
    if (s_begin < 0 || s_end > T || s_begin > s_end || (s_begin > 0 && !h_state))
        return fail(AICG_E_ARG, "aicg_gru_bidir_4wg_seg: steps %ld .. %ld of %ld (a segment that does not start at 0 needs h_state)",
                    (long)s_begin, (long)s_end, (long)T);
    if (T == 0 || s_begin == s_end) return AICG_OK;
    const size_t xbytes = (size_t)2 * 4 * 2 * (hidden / 4) * 8;
    if (s_begin == 0) (void)hipMemsetAsync(xchg_scratch, 0, xbytes + 64, (hipStream_t)stream);
    else (void)hipMemsetAsync((char*)xchg_scratch + xbytes + 16, 0, 32, (hipStream_t)stream);
    unsigned long long* xb = (unsigned long long*)xchg_scratch;
    int* err = (int*)((char*)xchg_scratch + xbytes);
    AICG_SWITCH(force_agent, "AICG_GRU_AGENT_STORES", 0);  // A/B switch
#ifdef AICG_EMULATED
    const unsigned nblocks = 8;
#else
    const unsigned nblocks = 32;
#endif
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gru4_kernel<256>), dim3(nblocks), dim3(384), 0, (hipStream_t)stream, gi, whh_t, bhh, out, (long)T,
                       xb, err, (int)force_agent, (long)s_begin, (long)s_end, h_state);
    return check_launch("gru4_kernel");
}

extern "C" int aicg_gru_bidir_4wg(const float* gi, const float* whh_t, const float* bhh, float* out, int hidden, int64_t T,
                                  void* xchg_scratch, void* stream) {
    return aicg_gru_bidir_4wg_seg(gi, whh_t, bhh, out, hidden, T, 0, T, nullptr, xchg_scratch, stream);
}

extern "C" int aicg_salience_decode(const float* salience, double* cents, double* f0, int* center, int64_t T, int n_bins,
                                    float thred, void* stream) {
    if (!salience || !cents || !f0) return fail(AICG_E_ARG, "aicg_salience_decode: null pointer");
    if (n_bins < 9) return fail(AICG_E_SHAPE, "aicg_salience_decode: need at least 9 bins");
    if (T == 0) return AICG_OK;
    hipLaunchKernelGGL(salience_decode_kernel, dim3((unsigned)ldiv_up(T, 4)), dim3(256), 0, (hipStream_t)stream, salience,
                       cents, f0, center, (long)T, n_bins, thred);
    return check_launch("salience_decode_kernel");
}

extern "C" int aicg_f0_coarse(const double* f0_in, double factor, double* f0_out, int64_t* coarse, int64_t n, double mel_min,
                              double mel_max, void* stream) {
    if (!f0_in || !f0_out || !coarse) return fail(AICG_E_ARG, "aicg_f0_coarse: null pointer");
    if (n == 0) return AICG_OK;
    hipLaunchKernelGGL(f0_coarse_kernel, dim3(ew_grid2(n)), dim3(256), 0, (hipStream_t)stream, f0_in, factor, f0_out,
                       (long*)coarse, (long)n, mel_min, mel_max);
    return check_launch("f0_coarse_kernel");
}

// ---- CREPE decode: softmax over bins + Viterbi (torchcrepe.decode.viterbi -> librosa.sequence.viterbi) ------------
// probs: (n_seq, 360, n_steps_max) fp32 sigmoid outputs with bins outside [lo, hi) already masked by the caller's
// convention (handled here: masked bins get -inf before the softmax).  One workgroup per sequence (the reference
// decodes each 2*hop-frame batch independently); thread j owns state j.  Arithmetic in float64 like numpy:
//   value[0][j] = log_prob[0][j] + log(1/360 + eps);  value[t][j] = log_prob[t][j] + max_i(value[t-1][i] + log_trans[i][j])
// with first-index argmax, eps = float32 tiny, transition = normalised max(12 - |i-j|, 0).
namespace aicg {
__global__ void __launch_bounds__(384) crepe_viterbi_kernel(const float* __restrict__ probs, const int* __restrict__ seq_len,
                                                            float* __restrict__ logp, unsigned short* __restrict__ ptr,
                                                            long* __restrict__ bins_out, int NB, int max_steps, int lo, int hi) {
    HIP_DYNAMIC_SHARED(double, dsm)
    double (*val)[384] = reinterpret_cast<double (*)[384]>(dsm);  // [2][384]
    double* ltrans_band = dsm + 2 * 384;  // log_trans[i][j] for |i - j| <= 11, indexed [i][j - i + 11]; 23 of 24 used
    const int seq = blockIdx.x, j = threadIdx.x;
    const int T = seq_len[seq];
    const float* pr = probs + (long)seq * NB * max_steps;
    float* lp = logp + (long)seq * NB * max_steps;              // [t][j]
    unsigned short* pt = ptr + (long)seq * NB * max_steps;      // [t][j]
    const float eps32 = 1.17549435e-38f;
    const double eps = (double)eps32;
    // phase 1: per-frame softmax over bins (fp32, like torch.nn.functional.softmax), log(prob + eps) in fp32
    for (int t = j; t < T; t += 384) {
        float mx = -INFINITY;
        for (int b = lo; b < hi; ++b) mx = fmaxf(mx, pr[(long)b * max_steps + t]);
        float sum = 0.f;
        for (int b = lo; b < hi; ++b) sum += expf(pr[(long)b * max_steps + t] - mx);
        for (int b = 0; b < NB; ++b) {
            const float p = (b >= lo && b < hi) ? expf(pr[(long)b * max_steps + t] - mx) / sum : 0.f;
            lp[(long)t * NB + b] = logf(p + eps32);
        }
    }
    // transition band of row j (rows are normalised: sum_i max(12 - |i - j|, 0))
    if (j < NB) {
        double rs = 0.0;
        for (int i = 0; i < NB; ++i) { const int d = i > j ? i - j : j - i; if (d < 12) rs += (double)(12 - d); }
        for (int d = -11; d <= 11; ++d) {
            const int i = j + d;
            ltrans_band[j * 24 + d + 11] = (i >= 0 && i < NB) ? log((double)(12 - (d < 0 ? -d : d)) / rs + eps) : 0.0;
        }
    }
    const double lfar = log(eps);                      // log(0 + eps) for |i - j| >= 12
    const double lpinit = log(1.0 / (double)NB + eps);
    __syncthreads();
    if (T <= 0) return;
    if (j < NB) val[0][j] = (double)lp[j] + lpinit;
    __syncthreads();
    int cur = 0;
    for (int t = 1; t < T; ++t) {
        if (j < NB) {
            // trans_out[j][i] = value[t-1][i] + log_trans[i][j]; the matrix is symmetric in |i - j| but rows are normalised
            // per source state i, so the band entry must be taken from row i
            double best = -INFINITY;
            int bi = 0;
            for (int i = 0; i < NB; ++i) {
                const int d = j - i;  // position of j in row i
                const double lt = (d >= -11 && d <= 11) ? ltrans_band[i * 24 + d + 11] : lfar;
                const double v = val[cur][i] + lt;
                if (v > best) { best = v; bi = i; }
            }
            pt[(long)t * NB + j] = (unsigned short)bi;
            val[cur ^ 1][j] = (double)lp[(long)t * NB + j] + best;
        }
        cur ^= 1;
        __syncthreads();
    }
    // state[T-1] = argmax_j value[T-1][j] (first maximum), then backtrack
    if (j == 0) {
        double best = -INFINITY;
        int s = 0;
        for (int i = 0; i < NB; ++i) if (val[cur][i] > best) { best = val[cur][i]; s = i; }
        long* bo = bins_out + (long)seq * max_steps;
        bo[T - 1] = s;
        for (int t = T - 2; t >= 0; --t) { s = pt[(long)(t + 1) * NB + s]; bo[t] = s; }
    }
}
}  // namespace aicg

extern "C" int aicg_crepe_viterbi(const float* probs, const int* seq_len, float* logp_scratch, uint16_t* ptr_scratch,
                                  int64_t* bins_out, int n_seq, int n_bins, int max_steps, int bin_lo, int bin_hi, void* stream) {
    if (!probs || !seq_len || !logp_scratch || !ptr_scratch || !bins_out) return aicg::fail(AICG_E_ARG, "aicg_crepe_viterbi: null pointer");
    if (n_bins > 384 || n_bins < 24 || bin_lo < 0 || bin_hi > n_bins || bin_lo >= bin_hi)
        return aicg::fail(AICG_E_SHAPE, "aicg_crepe_viterbi: bad bin range");
    if (n_seq <= 0) return AICG_OK;
    const size_t lds = (size_t)(2 * 384 + 384 * 24) * sizeof(double);
    allow_dynamic_lds((const void*)aicg::crepe_viterbi_kernel, lds);
    hipLaunchKernelGGL(aicg::crepe_viterbi_kernel, dim3((unsigned)n_seq), dim3(384), lds, (hipStream_t)stream, probs, seq_len,
                       logp_scratch, (unsigned short*)ptr_scratch, (long*)bins_out, n_bins, max_steps, bin_lo, bin_hi);
    return aicg::check_launch("crepe_viterbi_kernel");
}
