// GroupNorm (+ optional SiLU) forward / backward on the padded-NHWC layout.
//
// HBM-bound.  A block owns a run of pixels of ONE sample and ALL channels: every lane moves
// 16 B (8 consecutive bf16 channels of one pixel), so a pixel row (C*2 bytes) is read by C/8
// adjacent lanes -- fully coalesced -- and per-group statistics are formed by folding the 8
// per-lane channel accumulators into LDS by group id.  Statistics take two launches (partial
// slab, then fold in the consumer's prologue): no atomics on the forward path, deterministic.
//
//   fwd : stats(x) -> partial[n][chunk][g] = (sum, sumsq)
//         apply    -> y = act(gamma * (x - mean) * rstd + beta)        act = SiLU or identity
//   bwd : stats(dy, x) -> partial[n2][chunk][g] = (S1, S2);  dgamma/dbeta per gradient set
//         apply    -> dx = rstd * (dz*gamma - S1/cnt - xhat * S2/cnt) (+ accum) ; optional
//                     per-(sample, channel) column sums of dx (time-embedding / bias gradient)
// Backward takes n2 = sets * B cotangent samples against B saved samples (x index = n2 % Bx):
// the dual-cotangent backward of the SISS step.
// Output / cotangent rows may be "compact" ([N][H*W][C], no halo) for the attention block.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxG = 32;
constexpr int kMaxC = 1024;

struct GNShape {
    int H, W, C, G, cpg, lpp, ppi;     // lanes per pixel (C/8), pixels per iteration
    int chunk_px, nchunks;             // interior pixels per block, blocks per sample
};

__device__ __forceinline__ long padded_row(int n, int pi, int H, int W) {
    const int y = pi / W, x = pi - y * W;
    return ((long)n * (H + 2) + (y + 1)) * (W + 2) + (x + 1);
}
__device__ __forceinline__ long compact_row(int n, int pi, int H, int W) { return (long)n * H * W + pi; }

__device__ __forceinline__ void unpack8(u32x4_t r, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = __builtin_bit_cast(float, r[j] << 16);
        v[2 * j + 1] = __builtin_bit_cast(float, r[j] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4_t pack8(const float (&v)[8]) {
    return u32x4_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
}

// ---------------------------------------------------------------- forward: partial statistics
__global__ __launch_bounds__(kThreads) void gn_stats_kernel(const bf16_t* __restrict__ x, GNShape s,
                                                            float* __restrict__ partial) {
    __shared__ float sh[2 * kMaxG];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    if (tid < 2 * kMaxG) sh[tid] = 0.f;
    __syncthreads();
    const int slot = tid / s.lpp, cc = tid - slot * s.lpp;
    float a[8] = {}, b[8] = {};
    if (slot < s.ppi) {
        const int p0 = chunk * s.chunk_px;
        int p1 = p0 + s.chunk_px; p1 = p1 < s.H * s.W ? p1 : s.H * s.W;
        for (int pi = p0 + slot; pi < p1; pi += s.ppi) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(x + padded_row(n, pi, s.H, s.W) * s.C + cc * 8), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] += v[e]; b[e] += v[e] * v[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (cc * 8 + e) / s.cpg;
            atomicAdd(&sh[2 * g], a[e]);
            atomicAdd(&sh[2 * g + 1], b[e]);
        }
    }
    __syncthreads();
    if (tid < 2 * s.G) partial[((long)n * s.nchunks + chunk) * 2 * s.G + tid] = sh[tid];
}

// fold the partial slab for sample n: mean / rstd per group into LDS (and out to global once)
__device__ __forceinline__ void fold_stats(const float* __restrict__ partial, const GNShape& s, int n,
                                           float eps, float* sh_mean, float* sh_rstd,
                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                           bool write) {
    const int tid = threadIdx.x;
    if (tid < s.G) {
        double a = 0, b = 0;
        for (int c = 0; c < s.nchunks; ++c) {
            const float* q = partial + ((long)n * s.nchunks + c) * 2 * s.G + 2 * tid;
            a += q[0]; b += q[1];
        }
        const double cnt = (double)s.H * s.W * s.cpg;
        const double m = a / cnt;
        double var = b / cnt - m * m;
        var = var > 0 ? var : 0;
        const float r = (float)(1.0 / sqrt(var + (double)eps));
        sh_mean[tid] = (float)m; sh_rstd[tid] = r;
        if (write) { mean_out[(long)n * s.G + tid] = (float)m; rstd_out[(long)n * s.G + tid] = r; }
    }
    __syncthreads();
}

template <bool SILU>
__global__ __launch_bounds__(kThreads) void gn_apply_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ partial, GNShape s, float eps, int out_compact, bf16_t* __restrict__ y,
    float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ float sh_mean[kMaxG], sh_rstd[kMaxG];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    fold_stats(partial, s, n, eps, sh_mean, sh_rstd, mean_out, rstd_out, chunk == 0);
    const int slot = tid / s.lpp, cc = tid - slot * s.lpp;
    if (slot >= s.ppi) return;
    float sc[8], sf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cc * 8 + e, g = c / s.cpg;
        sc[e] = sh_rstd[g] * gamma[c];
        sf[e] = beta[c] - sh_mean[g] * sc[e];
    }
    const int p0 = chunk * s.chunk_px;
    int p1 = p0 + s.chunk_px; p1 = p1 < s.H * s.W ? p1 : s.H * s.W;
    for (int pi = p0 + slot; pi < p1; pi += s.ppi) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(x + padded_row(n, pi, s.H, s.W) * s.C + cc * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float z = v[e] * sc[e] + sf[e];
            v[e] = SILU ? silu_f(z) : z;
        }
        const long orow = out_compact ? compact_row(n, pi, s.H, s.W) : padded_row(n, pi, s.H, s.W);
        *reinterpret_cast<u32x4_t*>(y + orow * s.C + cc * 8) = pack8(v);
    }
}

// ---------------------------------------------------------------- backward: partial sums
template <bool SILU>
__global__ __launch_bounds__(kThreads) void gn_bwd_stats_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
    GNShape s, int nx, int dy_compact, int set_images, long set_stride, float* __restrict__ partial,
    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float sh[2 * kMaxG];
    __shared__ float shg[kMaxC], shb[kMaxC];
    const int n2 = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int n = n2 % nx;
    if (tid < 2 * kMaxG) sh[tid] = 0.f;
    for (int i = tid; i < s.C; i += kThreads) { shg[i] = 0.f; shb[i] = 0.f; }
    __syncthreads();
    const int slot = tid / s.lpp, cc = tid - slot * s.lpp;
    if (slot < s.ppi) {
        float mu[8], rs[8], ga[8], be[8], a1[8] = {}, a2[8] = {};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cc * 8 + e, g = c / s.cpg;
            mu[e] = mean[(long)n * s.G + g]; rs[e] = rstd[(long)n * s.G + g];
            ga[e] = gamma[c]; be[e] = beta[c];
        }
        const int p0 = chunk * s.chunk_px;
        int p1 = p0 + s.chunk_px; p1 = p1 < s.H * s.W ? p1 : s.H * s.W;
        for (int pi = p0 + slot; pi < p1; pi += s.ppi) {
            float v[8], d[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(x + padded_row(n, pi, s.H, s.W) * s.C + cc * 8), v);
            const long drow = dy_compact ? compact_row(n2, pi, s.H, s.W) : padded_row(n2, pi, s.H, s.W);
            unpack8(*reinterpret_cast<const u32x4_t*>(dy + drow * s.C + cc * 8), d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (v[e] - mu[e]) * rs[e];
                const float dz = SILU ? d[e] * dsilu_f(xh * ga[e] + be[e]) : d[e];
                a1[e] += dz; a2[e] += dz * xh;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cc * 8 + e, g = c / s.cpg;
            atomicAdd(&sh[2 * g], a1[e] * ga[e]);
            atomicAdd(&sh[2 * g + 1], a2[e] * ga[e]);
            atomicAdd(&shb[c], a1[e]);
            atomicAdd(&shg[c], a2[e]);
        }
    }
    __syncthreads();
    if (tid < 2 * s.G) partial[((long)n2 * s.nchunks + chunk) * 2 * s.G + tid] = sh[tid];
    const long so = (long)(n2 / set_images) * set_stride;
    for (int i = tid; i < s.C; i += kThreads) {
        atomicAdd(dgamma + so + i, shg[i]);
        atomicAdd(dbeta + so + i, shb[i]);
    }
}

template <bool SILU>
__global__ __launch_bounds__(kThreads) void gn_bwd_apply_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ partial, GNShape s, int nx, int dy_compact, const bf16_t* __restrict__ accum,
    bf16_t* __restrict__ dx, float* __restrict__ colsum) {
    __shared__ float sh_s1[kMaxG], sh_s2[kMaxG];
    __shared__ float shc[kMaxC];
    const int n2 = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int n = n2 % nx;
    if (tid < s.G) {
        double a = 0, b = 0;
        for (int c = 0; c < s.nchunks; ++c) {
            const float* q = partial + ((long)n2 * s.nchunks + c) * 2 * s.G + 2 * tid;
            a += q[0]; b += q[1];
        }
        const double cnt = (double)s.H * s.W * s.cpg;
        sh_s1[tid] = (float)(a / cnt); sh_s2[tid] = (float)(b / cnt);
    }
    if (colsum) for (int i = tid; i < s.C; i += kThreads) shc[i] = 0.f;
    __syncthreads();
    const int slot = tid / s.lpp, cc = tid - slot * s.lpp;
    if (slot < s.ppi) {
        float mu[8], rs[8], ga[8], be[8], m1[8], m2[8], cs[8] = {};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cc * 8 + e, g = c / s.cpg;
            mu[e] = mean[(long)n * s.G + g]; rs[e] = rstd[(long)n * s.G + g];
            ga[e] = gamma[c]; be[e] = beta[c]; m1[e] = sh_s1[g]; m2[e] = sh_s2[g];
        }
        const int p0 = chunk * s.chunk_px;
        int p1 = p0 + s.chunk_px; p1 = p1 < s.H * s.W ? p1 : s.H * s.W;
        for (int pi = p0 + slot; pi < p1; pi += s.ppi) {
            float v[8], d[8], o[8];
            const long xrow = padded_row(n, pi, s.H, s.W);
            const long orow = padded_row(n2, pi, s.H, s.W);
            unpack8(*reinterpret_cast<const u32x4_t*>(x + xrow * s.C + cc * 8), v);
            const long drow = dy_compact ? compact_row(n2, pi, s.H, s.W) : orow;
            unpack8(*reinterpret_cast<const u32x4_t*>(dy + drow * s.C + cc * 8), d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (v[e] - mu[e]) * rs[e];
                const float dz = SILU ? d[e] * dsilu_f(xh * ga[e] + be[e]) : d[e];
                o[e] = rs[e] * (dz * ga[e] - m1[e] - xh * m2[e]);
                cs[e] += o[e];
            }
            if (accum) {
                float r[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(accum + orow * s.C + cc * 8), r);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += r[e];
            }
            *reinterpret_cast<u32x4_t*>(dx + orow * s.C + cc * 8) = pack8(o);
        }
        if (colsum) {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(&shc[cc * 8 + e], cs[e]);
        }
    }
    if (colsum) {
        __syncthreads();
        for (int i = tid; i < s.C; i += kThreads) atomicAdd(colsum + (long)n2 * s.C + i, shc[i]);
    }
}

bool make_shape(int H, int W, int C, int G, GNShape& s) {
    if (H <= 0 || W <= 0 || C <= 0 || G <= 0 || G > kMaxG || C % G || C % 8 || C > kMaxC) return false;
    s.H = H; s.W = W; s.C = C; s.G = G; s.cpg = C / G;
    s.lpp = C / 8;
    if (s.lpp > kThreads) return false;
    s.ppi = kThreads / s.lpp;
    const int px = H * W;
    int nch = (px + 1023) / 1024;            // >= 1024 pixels per block unless the image is small
    if (nch > 64) nch = 64;
    if (nch < 1) nch = 1;
    s.chunk_px = (px + nch - 1) / nch;
    s.nchunks = (px + s.chunk_px - 1) / s.chunk_px;
    return true;
}

}  // namespace

extern "C" {

// floats needed in `partial` for n samples
long siss_gn_partial_words(int n, int H, int W, int C, int G) {
    GNShape s;
    if (!make_shape(H, W, C, G, s)) return -1;
    return (long)n * s.nchunks * 2 * G;
}

// y = act(GroupNorm(x)); x padded NHWC; y padded or compact ([N][H*W][C]).  Writes mean/rstd [N][G].
int siss_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                       float* rstd, float* partial, int N, int H, int W, int C, int G, float eps,
                       int silu, int out_compact, void* stream) {
    GNShape s;
    SISS_CHECK_ARG(x && gamma && beta && y && mean && rstd && partial && N > 0);
    SISS_CHECK_ARG(make_shape(H, W, C, G, s));
    SISS_CHECK_ARG(((uintptr_t)x | (uintptr_t)y) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(s.nchunks, N);
    gn_stats_kernel<<<grid, kThreads, 0, st>>>((const bf16_t*)x, s, partial);
    if (silu)
        gn_apply_kernel<true><<<grid, kThreads, 0, st>>>((const bf16_t*)x, gamma, beta, partial, s, eps, out_compact, (bf16_t*)y, mean, rstd);
    else
        gn_apply_kernel<false><<<grid, kThreads, 0, st>>>((const bf16_t*)x, gamma, beta, partial, s, eps, out_compact, (bf16_t*)y, mean, rstd);
    SISS_LAUNCH_RET();
}

// dx (padded, n2 samples) from dy (n2 samples, padded or compact) and the saved x (nx samples,
// x index = n2 % nx).  dgamma/dbeta: [sets][...] accumulated atomically at set = n2 / set_images
// with `set_stride` floats between sets.  accum (optional, padded like dx) is added to dx;
// colsum (optional, [n2][C] f32, pre-zeroed) receives the per-sample channel sums of dx.
int siss_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta,
                       const float* mean, const float* rstd, void* dx, const void* accum, float* dgamma,
                       float* dbeta, float* colsum, float* partial, int n2, int nx, int set_images,
                       long set_stride, int H, int W, int C, int G, int silu, int dy_compact, void* stream) {
    GNShape s;
    SISS_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && dgamma && dbeta && partial);
    SISS_CHECK_ARG(n2 > 0 && nx > 0 && set_images > 0 && n2 % set_images == 0);
    SISS_CHECK_ARG(make_shape(H, W, C, G, s));
    SISS_CHECK_ARG(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)accum) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(s.nchunks, n2);
    if (silu) {
        gn_bwd_stats_kernel<true><<<grid, kThreads, 0, st>>>((const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, s, nx, dy_compact, set_images, set_stride, partial, dgamma, dbeta);
        gn_bwd_apply_kernel<true><<<grid, kThreads, 0, st>>>((const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, partial, s, nx, dy_compact, (const bf16_t*)accum, (bf16_t*)dx, colsum);
    } else {
        gn_bwd_stats_kernel<false><<<grid, kThreads, 0, st>>>((const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, s, nx, dy_compact, set_images, set_stride, partial, dgamma, dbeta);
        gn_bwd_apply_kernel<false><<<grid, kThreads, 0, st>>>((const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, partial, s, nx, dy_compact, (const bf16_t*)accum, (bf16_t*)dx, colsum);
    }
    SISS_LAUNCH_RET();
}

}  // extern "C"
