// Signal-processing steps either side of the networks, moved onto the device:
//   * 3-tap NaN-aware median / mean of a frame sequence  -- torchcrepe.filter.median / .mean as called for f0_method 'crepe'
//     and 'crepe-tiny' (reference src/vc_infer_pipeline.py:160-161)
//   * zero-phase IIR (scipy.signal.filtfilt, float64)     -- the 48 Hz high-pass of VC.pipeline (:513)
//   * polyphase FIR resampling (scipy.signal.resample_poly semantics) + channel mean -- the 44.1 kHz stereo -> 16 kHz mono hand-over
//     between run_mdx and rvc_infer that the reference does through a WAV file and ffmpeg (mdx.py:273,280; my_utils.py:14-16)
//   * exact k = 8 nearest neighbours + inverse-square-distance feature mix -- index.search / big_npy blend (:409-431)
// All HBM-bound except the kNN distances, which run on the conv kernel as a 1x1 GEMM (ops.py).
#include "common.h"

namespace aicg {

// ---- 3-tap filters ----------------------------------------------------------------------------------------------------------------
// Window = the in-range, non-NaN neighbours {i-1, i, i+1}.  mode 0: lower median (torch.median / sorted[(count - 1) / 2]),
// mode 1: mean; no valid sample -> NaN; a mean of exactly 0 -> NaN (torchcrepe.filter.mean marks empty windows that way).
// Neighbours travel by wave shuffles; only the two edge lanes of a wave touch memory for them.
__global__ void __launch_bounds__(256) filter3_kernel(const float* __restrict__ x, float* __restrict__ out, long n, int mode) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const float nan = __int_as_float(0x7fc00000);
    const float v = i < n ? x[i] : nan;
    float l = __shfl_up(v, 1, 64), r = __shfl_down(v, 1, 64);
    if (lane == 0) l = (i >= 1 && i - 1 < n) ? x[i - 1] : nan;
    if (lane == 63) r = (i + 1 < n) ? x[i + 1] : nan;
    if (i >= n) return;
    if (i == 0) l = nan;
    if (i + 1 >= n) r = nan;
    const bool vl = l == l, vc = v == v, vr = r == r;
    const int cnt = (int)vl + (int)vc + (int)vr;
    float res = nan;
    if (mode == 1) {
        if (cnt > 0) {
            res = ((vl ? l : 0.f) + (vc ? v : 0.f) + (vr ? r : 0.f)) / (float)cnt;
            if (res == 0.f) res = nan;
        }
    } else if (cnt == 3) {
        res = fmaxf(fminf(l, v), fminf(fmaxf(l, v), r));
    } else if (cnt == 2) {
        const float a = vl ? l : v, b = vr ? r : v;   // the two valid ones
        res = fminf(a, b);
    } else if (cnt == 1) {
        res = vl ? l : (vc ? v : r);
    }
    out[i] = res;
}

// ---- zero-phase IIR ------------------------------------------------------------------------------------------------------------------
// scipy.signal.filtfilt(b, a, x) with its defaults (padtype 'odd', padlen = 3 * max(len(a), len(b)), method 'pad'):
//   ext = [2 x[0] - x[padlen..1], x, 2 x[-1] - x[-2..-padlen-1]];  y1 = lfilter(b, a, ext, zi = zi * ext[0]);
//   y2 = lfilter(b, a, reverse(y1), zi = zi * y1[-1]);  result = reverse(y2)[padlen : -padlen].
// lfilter is the transposed direct form II recurrence, strictly sequential along the signal.  The filter is stable, so the
// influence of the state `warm` samples back has decayed below the float64 rounding of the outputs (the host picks
// warm >= log(1e-18) / log(max |pole|)): one thread per block of `block` outputs re-runs the recurrence over the `warm` samples in
// front of its block from a zero state and keeps only its own outputs; block 0 starts from the true initial state.
static constexpr int IIR_MAX = 8;  // filter order limit (the reference's Butterworth is order 5)

struct IirArgs {
    double b[IIR_MAX + 1], a[IIR_MAX + 1], zi[IIR_MAX];
    int order;
};

__global__ void __launch_bounds__(64) odd_ext_kernel(const double* __restrict__ x, double* __restrict__ ext, long n, int padlen) {
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    if (i >= n + 2 * padlen) return;
    double v;
    if (i < padlen) v = 2.0 * x[0] - x[padlen - i];
    else if (i < padlen + n) v = x[i - padlen];
    else v = 2.0 * x[n - 1] - x[n - 2 - (i - padlen - n)];
    ext[i] = v;
}

// in / out: length m; reverse != 0: the signal is traversed back to front (sample j of the pass = in[m - 1 - j], written to
// out[m - 1 - j]); out_off / out_len: only pass outputs j in [out_off, out_off + out_len) are stored, at out[...] - out_off.
__global__ void __launch_bounds__(64) iir_blocks_kernel(const double* __restrict__ in, double* __restrict__ out, long m, int block, int warm,
                                                        int reverse, long out_off, long out_len, IirArgs f) {
    // The direct-form Butterworth is ill-conditioned (five poles clustered at z = 1: ~1e8 amplification of rounding): fused
    // multiply-adds would move the result 1e-8 away from scipy's lfilter, which rounds every product and sum.  No contraction
    // here, same operation order as scipy's _linear_filter: with one block the two are bit-identical.
#pragma clang fp contract(off)
    const long k = (long)blockIdx.x * 64 + threadIdx.x;
    const long j0 = k * block;
    if (j0 >= m) return;
    const long j1 = lmin(j0 + block, m);
    const long js = k == 0 ? 0 : lmax(0, j0 - warm);
    auto at = [&](long j) { return reverse ? in[m - 1 - j] : in[j]; };
    double z[IIR_MAX];
    const double x0 = at(0);
#pragma unroll
    for (int q = 0; q < IIR_MAX; ++q) z[q] = (js == 0 && q < f.order) ? f.zi[q] * x0 : 0.0;
    auto step = [&](long j, double xv) {
        const double y = z[0] + f.b[0] * xv;
        // coefficients beyond the filter order are zero, so the states beyond it stay zero
#pragma unroll
        for (int q = 0; q < IIR_MAX - 1; ++q) z[q] = z[q + 1] + xv * f.b[q + 1] - y * f.a[q + 1];
        z[IIR_MAX - 1] = f.b[IIR_MAX] * xv - f.a[IIR_MAX] * y;
        if (j >= j0 && j >= out_off && j < out_off + out_len) {
            const long o = j - out_off;
            if (reverse) out[out_len - 1 - o] = y; else out[o] = y;
        }
    };
    // the recurrence is a dependent chain, the loads are not: 8 samples are fetched ahead of the 8 steps that consume them
    // (one load per step left every step waiting a full memory round trip: 3.2 ms per pass on a 4-minute track)
    long j = js;
    for (; j + 8 <= j1; j += 8) {
        double xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = at(j + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) step(j + u, xv[u]);
    }
    for (; j < j1; ++j) step(j, at(j));
}

// ---- polyphase resampling -------------------------------------------------------------------------------------------------------------
// y[i] = sum_m h[(i + pre) * down - m * up] * xm[m],  xm = mean over `nch` channels of x (channel stride x_sc), h given as a
// polyphase table hp[phase][tap] = h[phase + tap * up] (taps = ceil(hlen / up), zero padded): the upfirdn form of
// scipy.signal.resample_poly(x, up, down) whose trimming the host expresses through `pre`.  Accumulation in float64.
__global__ void __launch_bounds__(256) resample_poly_kernel(const float* __restrict__ x, float* __restrict__ y, long n_in, long n_out, int nch,
                                                            long x_sc, int up, int down, const float* __restrict__ hp, int taps, long pre) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const long t = (i + pre) * down;
    const long m_hi = t / up;                 // largest m with t - m * up >= 0
    const int phase = (int)(t - m_hi * up);   // h index of the newest sample
    const float* hrow = hp + (long)phase * taps;
    const float inv = 1.f / (float)nch;
    double acc = 0.0;
    for (int j = 0; j < taps; ++j) {
        const long m = m_hi - j;
        if (m < 0) break;
        if (m >= n_in) continue;
        float s = x[m];
        for (int c = 1; c < nch; ++c) s += x[(long)c * x_sc + m];
        acc += (double)hrow[j] * (double)(nch > 1 ? s * inv : s);
    }
    y[i] = (float)acc;
}

// ---- k = 8 nearest neighbours + mix ---------------------------------------------------------------------------------------------------
// dots: [rows][cols] inner products q_r . x_c (row stride lds); xnorm[c] = |x_c|^2; qnorm[r] = |q_r|^2.  Squared L2 distance
// d = qnorm - 2 dot + xnorm.  One wave per row: every lane keeps the 8 smallest of its strided columns (sorted insertion), then 8
// rounds of wave arg-min (shuffle butterflies) pop the global 8 smallest, ties broken by the lower column.  best_* hold the
// running result over column chunks (col_off = first column of this chunk; merge != 0: previous bests take part).
__global__ void __launch_bounds__(64) knn8_kernel(const float* __restrict__ dots, long lds, const float* __restrict__ xnorm,
                                                  const float* __restrict__ qnorm, int rows, int cols, long col_off, float* __restrict__ best_d,
                                                  long* __restrict__ best_i, int merge) {
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= rows) return;
    const float inf = __int_as_float(0x7f800000);
    float d[8];
    long id[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { d[k] = inf; id[k] = 0x7fffffffffffffffL; }
    auto push = [&](float v, long c) {
        if (!(v < d[7] || (v == d[7] && c < id[7]))) return;
        d[7] = v; id[7] = c;
#pragma unroll
        for (int k = 7; k > 0; --k) {
            if (d[k] < d[k - 1] || (d[k] == d[k - 1] && id[k] < id[k - 1])) {
                const float tv = d[k]; d[k] = d[k - 1]; d[k - 1] = tv;
                const long ti = id[k]; id[k] = id[k - 1]; id[k - 1] = ti;
            }
        }
    };
    const float qn = qnorm[r];
    const float* row = dots + (long)r * lds;
    for (int c = lane; c < cols; c += 64) push(fmaxf(qn - 2.f * row[c] + xnorm[c], 0.f), col_off + c);
    if (merge && lane < 8) push(best_d[(long)r * 8 + lane], best_i[(long)r * 8 + lane]);
    for (int k = 0; k < 8; ++k) {
        float v = d[0];
        long c = id[0];
        int who = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const long oc = __shfl_xor(c, o, 64);
            const int ow = __shfl_xor(who, o, 64);
            if (ov < v || (ov == v && oc < c)) { v = ov; c = oc; who = ow; }
        }
        if (lane == 0) { best_d[(long)r * 8 + k] = v; best_i[(long)r * 8 + k] = c; }
        if (lane == who) {  // pop the winner's head
#pragma unroll
            for (int q = 0; q < 7; ++q) { d[q] = d[q + 1]; id[q] = id[q + 1]; }
            d[7] = inf; id[7] = 0x7fffffffffffffffL;
        }
    }
}

// IVF-Flat scan (faiss IndexIVFFlat::search semantics, reference :421 `index.search(npy, k=8)` on an index whose file says
// nprobe): query r visits the `nprobe` inverted lists probe[r][0..nprobe) (its nearest centroids, nearest first) and keeps the 8
// smallest squared L2 distances, each computed DIRECTLY as sum_c (q_c - x_c)^2 like faiss' fvec_L2sqr (no |q|^2 - 2 q.x + |x|^2
// cancellation).  Vectors are stored list by list (vecs[list_off[l] .. list_off[l + 1])), the order of faiss' inverted lists; the
// result is the POSITION in that storage (the caller maps positions to faiss labels).  A candidate replaces the current 8th only
// if it is strictly closer, so among equal distances the earlier-scanned one stays -- faiss' max-heap rule.  Lists with fewer than
// 8 vectors leave distance = +inf, position = -1 (faiss: FLT_MAX / -1; both give weight 0 in the mix).
// One workgroup of 4 waves per query; wave w takes every 4th vector, a lane 4 consecutive components per 256 (float4 loads).
__global__ void __launch_bounds__(256) ivf_scan8_kernel(const float* __restrict__ q, const float* __restrict__ vecs,
                                                        const long* __restrict__ list_off, const long* __restrict__ probe, int probe_ld,
                                                        int nprobe, int rows, int dim, float* __restrict__ best_d, long* __restrict__ best_pos) {
    const int r = blockIdx.x;
    if (r >= rows) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inf = __int_as_float(0x7f800000);
    __shared__ float sd[4][8];
    __shared__ long sk[4][8], sp[4][8];
    float d[8];
    long key[8], pos[8];   // key = position in the query's scan order (the tie-break), pos = position in `vecs`
#pragma unroll
    for (int k = 0; k < 8; ++k) { d[k] = inf; key[k] = 0x7fffffffffffffffL; pos[k] = -1; }
    const float* qr = q + (long)r * dim;
    long scanned = 0;   // scan position of the first vector of the current list
    for (int p = 0; p < nprobe; ++p) {
        const long l = probe[(long)r * probe_ld + p];
        if (l < 0) continue;
        const long b = list_off[l], e = list_off[l + 1];
        for (long v = b + wave; v < e; v += 4) {
            const float* x = vecs + v * dim;
            float acc = 0.f;
            for (int c = lane * 4; c < dim; c += 256) {
                const float4 a = *reinterpret_cast<const float4*>(qr + c);
                const float4 y = *reinterpret_cast<const float4*>(x + c);
                const float t0 = a.x - y.x, t1 = a.y - y.y, t2 = a.z - y.z, t3 = a.w - y.w;
                acc += (t0 * t0 + t1 * t1) + (t2 * t2 + t3 * t3);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            // every lane of the wave now holds the same sum and the same sorted list: uniform insertion, no divergence
            const long kv = scanned + (v - b);
            if (acc < d[7] || (acc == d[7] && kv < key[7])) {
                d[7] = acc; key[7] = kv; pos[7] = v;
#pragma unroll
                for (int k = 7; k > 0; --k) {
                    if (d[k] < d[k - 1] || (d[k] == d[k - 1] && key[k] < key[k - 1])) {
                        const float tv = d[k]; d[k] = d[k - 1]; d[k - 1] = tv;
                        const long tk = key[k]; key[k] = key[k - 1]; key[k - 1] = tk;
                        const long tp = pos[k]; pos[k] = pos[k - 1]; pos[k - 1] = tp;
                    }
                }
            }
        }
        scanned += e - b;
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { sd[wave][k] = d[k]; sk[wave][k] = key[k]; sp[wave][k] = pos[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // 4-way merge of the waves' sorted lists by (distance, scan position)
        int h[4] = {0, 0, 0, 0};
        for (int k = 0; k < 8; ++k) {
            int bw = -1;
            for (int w = 0; w < 4; ++w) {
                if (h[w] >= 8) continue;
                if (bw < 0 || sd[w][h[w]] < sd[bw][h[bw]] || (sd[w][h[w]] == sd[bw][h[bw]] && sk[w][h[w]] < sk[bw][h[bw]])) bw = w;
            }
            best_d[(long)r * 8 + k] = sd[bw][h[bw]];
            best_pos[(long)r * 8 + k] = sp[bw][h[bw]];
            ++h[bw];
        }
    }
}

// feats[r][c] = rate * sum_k w_k big[idx_k][c] + (1 - rate) * feats[r][c],  w_k = (1 / d_k)^2 / sum_j (1 / d_j)^2
// (reference :417-431: weight = np.square(1 / score); weight /= weight.sum(axis=1, keepdims=True)).
// recompute != 0: the distances are re-evaluated directly, d_k = sum_c (feats[r][c] - big[idx_k][c])^2, before they are used
// (and written back to best_d): the exhaustive search ranks by |q|^2 - 2 q.x + |x|^2 from a GEMM, whose fp32 cancellation turns
// near-duplicates into exact zeros; the weights want the direct form faiss computes.
// idx_k < 0 (fewer than 8 neighbours found): weight 0.  An EXACT zero distance makes the reference's weights inf / inf = NaN; here
// the zero-distance neighbours share the whole weight instead (the limit of the formula), so one duplicated training frame cannot
// blank a chunk.
__global__ void __launch_bounds__(256) index_mix_kernel(float* __restrict__ feats, const float* __restrict__ big, float* __restrict__ best_d,
                                                        const long* __restrict__ best_i, int rows, int dim, float rate, int recompute) {
    const int r = blockIdx.x;
    if (r >= rows) return;
    __shared__ float sdist[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (recompute) {
        for (int k = wave; k < 8; k += 4) {
            const long id = best_i[(long)r * 8 + k];
            float acc = 0.f;
            if (id >= 0)
                for (int c = lane; c < dim; c += 64) { const float t = feats[(long)r * dim + c] - big[id * dim + c]; acc += t * t; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (lane == 0) sdist[k] = id >= 0 ? acc : __int_as_float(0x7f800000);
        }
    } else if (threadIdx.x < 8) {
        sdist[threadIdx.x] = best_d[(long)r * 8 + threadIdx.x];
    }
    __syncthreads();
    if (recompute && threadIdx.x < 8) best_d[(long)r * 8 + threadIdx.x] = sdist[threadIdx.x];
    float w[8];
    long id[8];
    float sum = 0.f;
    int zeros = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        id[k] = best_i[(long)r * 8 + k];
        const float dk = sdist[k];
        zeros += (id[k] >= 0 && dk == 0.f) ? 1 : 0;
        const float inv = id[k] >= 0 ? 1.f / dk : 0.f;
        w[k] = inv * inv;
        sum += w[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (zeros) w[k] = (id[k] >= 0 && sdist[k] == 0.f) ? 1.f / (float)zeros : 0.f;
        else w[k] = w[k] / sum;
        if (id[k] < 0) id[k] = 0;   // weight 0: any valid row
    }
    // No neighbour at all (every label -1: the probed inverted lists are empty, which k-means can produce at nprobe = 1): the
    // reference divides 0 by 0 here and the frame becomes NaN.  Like the zero-distance rule above this keeps one degenerate frame from
    // blanking a chunk: the frame keeps its own features.
    if (!zeros && !(sum > 0.f)) return;
    for (int c = threadIdx.x; c < dim; c += 256) {
        const float f = feats[(long)r * dim + c];
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += big[id[k] * dim + c] * w[k];
        feats[(long)r * dim + c] = acc * rate + (1.f - rate) * f;
    }
}

// |v_r|^2 of the rows of a [rows][dim] matrix (one wave per row)
__global__ void __launch_bounds__(64) row_sqnorm_kernel(const float* __restrict__ v, float* __restrict__ out, long rows, int dim) {
    const long r = blockIdx.x;
    if (r >= rows) return;
    float s = 0.f;
    for (int c = threadIdx.x; c < dim; c += 64) { const float t = v[r * dim + c]; s += t * t; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[r] = s;
}

}  // namespace aicg

using namespace aicg;

extern "C" int aicg_filter3(const float* x, float* out, int64_t n, int mode, void* stream) {
    if (!x || !out) return fail(AICG_E_ARG, "aicg_filter3: null pointer");
    if (mode != 0 && mode != 1) return fail(AICG_E_ARG, "aicg_filter3: mode must be 0 (median) or 1 (mean)");
    if (n <= 0) return AICG_OK;
    hipLaunchKernelGGL(filter3_kernel, dim3((unsigned)ldiv_up(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, (long)n, mode);
    return check_launch("filter3_kernel");
}

extern "C" int aicg_filtfilt_f64(const double* x, double* y, int64_t n, const double* b, const double* a, const double* zi, int order,
                                 int padlen, int block, int warm, double* ext, double* mid, void* stream) {
    if (!x || !y || !b || !a || !zi || !ext || !mid) return fail(AICG_E_ARG, "aicg_filtfilt_f64: null pointer");
    if (order < 1 || order > IIR_MAX) return fail(AICG_E_SHAPE, "aicg_filtfilt_f64: filter order must be 1..%d", IIR_MAX);
    if (n <= padlen) return fail(AICG_E_SHAPE, "aicg_filtfilt_f64: the signal must be longer than padlen (%d)", padlen);
    if (block < 1 || warm < 0) return fail(AICG_E_ARG, "aicg_filtfilt_f64: bad block / warm-up length");
    IirArgs f;
    for (int q = 0; q <= IIR_MAX; ++q) { f.b[q] = q <= order ? b[q] / a[0] : 0.0; f.a[q] = q <= order ? a[q] / a[0] : 0.0; }
    for (int q = 0; q < IIR_MAX; ++q) f.zi[q] = q < order ? zi[q] : 0.0;
    f.order = order;
    const long m = n + 2L * padlen;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(odd_ext_kernel, dim3((unsigned)ldiv_up(m, 64)), dim3(64), 0, st, x, ext, (long)n, padlen);
    const unsigned nb = (unsigned)ldiv_up(ldiv_up(m, block), 64);
    hipLaunchKernelGGL(iir_blocks_kernel, dim3(nb), dim3(64), 0, st, (const double*)ext, mid, m, block, warm, 0, 0L, m, f);
    // backward pass over mid; pass sample j is ext position m - 1 - j; keep ext positions [padlen, padlen + n)
    hipLaunchKernelGGL(iir_blocks_kernel, dim3(nb), dim3(64), 0, st, (const double*)mid, y, m, block, warm, 1, (long)padlen, (long)n, f);
    return check_launch("iir_blocks_kernel");
}

extern "C" int aicg_resample_poly(const float* x, float* y, int64_t n_in, int64_t n_out, int n_channels, int64_t x_sc, int up, int down,
                                  const float* hp, int taps, int64_t pre, void* stream) {
    if (!x || !y || !hp) return fail(AICG_E_ARG, "aicg_resample_poly: null pointer");
    if (up < 1 || down < 1 || taps < 1 || n_channels < 1 || pre < 0) return fail(AICG_E_ARG, "aicg_resample_poly: bad rate / filter geometry");
    if (n_out <= 0 || n_in <= 0) return AICG_OK;
    hipLaunchKernelGGL(resample_poly_kernel, dim3((unsigned)ldiv_up(n_out, 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n_in,
                       (long)n_out, n_channels, (long)x_sc, up, down, hp, taps, (long)pre);
    return check_launch("resample_poly_kernel");
}

extern "C" int aicg_row_sqnorm(const float* v, float* out, int64_t rows, int dim, void* stream) {
    if (!v || !out) return fail(AICG_E_ARG, "aicg_row_sqnorm: null pointer");
    if (rows <= 0) return AICG_OK;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, v, out, (long)rows, dim);
    return check_launch("row_sqnorm_kernel");
}

extern "C" int aicg_knn8(const float* dots, int64_t ld, const float* xnorm, const float* qnorm, int rows, int cols, int64_t col_off,
                         float* best_d, int64_t* best_i, int merge, void* stream) {
    if (!dots || !xnorm || !qnorm || !best_d || !best_i) return fail(AICG_E_ARG, "aicg_knn8: null pointer");
    if (rows <= 0) return AICG_OK;
    if (cols < 0) return fail(AICG_E_SHAPE, "aicg_knn8: negative column count");
    hipLaunchKernelGGL(knn8_kernel, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, dots, (long)ld, xnorm, qnorm, rows, cols,
                       (long)col_off, best_d, (long*)best_i, merge);
    return check_launch("knn8_kernel");
}

extern "C" int aicg_index_mix(float* feats, const float* big, float* best_d, const int64_t* best_i, int rows, int dim, float rate,
                              int recompute, void* stream) {
    if (!feats || !big || !best_d || !best_i) return fail(AICG_E_ARG, "aicg_index_mix: null pointer");
    if (rows <= 0) return AICG_OK;
    hipLaunchKernelGGL(index_mix_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, feats, big, best_d, (const long*)best_i,
                       rows, dim, rate, recompute);
    return check_launch("index_mix_kernel");
}

extern "C" int aicg_ivf_scan8(const float* q, const float* vecs, const int64_t* list_off, const int64_t* probe, int probe_ld, int nprobe,
                              int rows, int dim, float* best_d, int64_t* best_pos, void* stream) {
    if (!q || !vecs || !list_off || !probe || !best_d || !best_pos) return fail(AICG_E_ARG, "aicg_ivf_scan8: null pointer");
    if (dim < 4 || dim % 4) return fail(AICG_E_SHAPE, "aicg_ivf_scan8: dim must be a positive multiple of 4 (got %d)", dim);
    if (nprobe < 1 || nprobe > probe_ld) return fail(AICG_E_ARG, "aicg_ivf_scan8: nprobe must be 1..probe_ld");
    if (rows <= 0) return AICG_OK;
    hipLaunchKernelGGL(ivf_scan8_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, q, vecs, (const long*)list_off,
                       (const long*)probe, probe_ld, nprobe, rows, dim, best_d, (long*)best_pos);
    return check_launch("ivf_scan8_kernel");
}
