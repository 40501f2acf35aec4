// Token-space pieces of the SD UNet's Transformer2DModel (SURVEY.md §8 a-U config 5, Appendix A7):
// LayerNorm, GEGLU, head split / merge for the batched attention GEMMs, and a row softmax of any length.
// All tensors are compact row-major bf16 [rows][C]; statistics and gradients are f32.
// HBM-bound helpers -- the matrix products around them run on gemm_nt / gemm_tn.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxChunks = 4;            // 8-channel chunks per lane: C <= 64 * 8 * 4 = 2048

__device__ __forceinline__ void unpack8f(const u32x4_t r, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = __builtin_bit_cast(float, r[e] << 16);
        v[2 * e + 1] = __builtin_bit_cast(float, r[e] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4_t pack8f(const float (&v)[8]) {
    return u32x4_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
}

// One wave per row; lane l owns 8-channel chunks l, l+64, ... (NCH of them: the SD widths 320 / 640 / 1280 need 1 / 2 /
// 3; a fixed 4 kept 160+ VGPRs live and ran the backward at two waves per SIMD)
template <int NCH>
__global__ __launch_bounds__(kThreads) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                                 float* __restrict__ mean, float* __restrict__ rstd,
                                                                 long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int nch = C >> 3;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            unpack8f(*reinterpret_cast<const u32x4_t*>(x + r * C + c * 8), v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e];
        }
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (lane + i * 64 < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
        }
    const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu) * rs * gamma[c * 8 + e] + beta[c * 8 + e];
            *reinterpret_cast<u32x4_t*>(y + r * C + c * 8) = pack8f(o);
        }
    }
}

// dx[r] = (accum[r] +) rstd * (dy*gamma - mean_c(dy*gamma) - xhat * mean_c(dy*gamma*xhat)); saved row = r % rows_x.
// dgamma / dbeta: per cotangent set (set = r / set_rows), per-lane partial column sums over the block's rows,
// folded across the block's waves in LDS; then either one f32 atomic per column and block (part == null) or -- round 6 -- one plain
// store per column into the block's slot of `part` ([block][2][C] f32), summed per set by ln_dgamma_reduce_kernel.  (The atomics of
// thousands of blocks land on the same 2 C addresses at the end of the launch: same-address float atomics run at ~1 / 14 of the
// atomic rate -- 4096 blocks x 640 columns were most of a 64-us launch whose HBM time is 30-67 us.)
template <int NCH>
__global__ __launch_bounds__(kThreads) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const bf16_t* __restrict__ accum,
                                                                 bf16_t* __restrict__ dx, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, long rows2, long rows_x, long set_rows,
                                                                 long set_stride, int C, int rows_per_block, float* __restrict__ part) {
    __shared__ float red[2][kThreads / 64][64 * 8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nch = C >> 3;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block; r1 = r1 < rows2 ? r1 : rows2;
    // a block never straddles two sets (rows_per_block divides set_rows)
    const long set = r0 / set_rows;
    float dg[NCH][8], db[NCH][8], ga[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dg[i][e] = 0.f; db[i][e] = 0.f;
            const int c = lane + i * 64;
            ga[i][e] = c < nch ? gamma[c * 8 + e] : 0.f;
        }
    // TWO rows per wave and trip: both rows' loads (dy, x, accum) are issued before either row's reductions -- the loop is a chain
    // load -> two wave reductions -> store per row, and with one row in flight per wave it ran at a quarter of the HBM rate
    constexpr int NW = kThreads / 64;
    for (long rr = r0 + w; rr < r1; rr += 2 * NW) {
        const bool two = rr + NW < r1;
        const long rrow[2] = {rr, two ? rr + NW : rr};
        u32x4_t vd[2][NCH], vx[2][NCH], va[2][NCH];
        float mu[2], rs[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const long r = rrow[q], rx = r % rows_x;
            mu[q] = mean[rx]; rs[q] = rstd[rx];
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = lane + i * 64;
                if (c < nch) {
                    vd[q][i] = *reinterpret_cast<const u32x4_t*>(dy + r * C + c * 8);
                    vx[q][i] = *reinterpret_cast<const u32x4_t*>(x + rx * C + c * 8);
                    if (accum) va[q][i] = *reinterpret_cast<const u32x4_t*>(accum + r * C + c * 8);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (q == 1 && !two) break;
            const long r = rrow[q];
            float d[NCH][8], xh[NCH][8];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = lane + i * 64;
                if (c < nch) {
                    float xv[8];
                    unpack8f(vd[q][i], d[i]);
                    unpack8f(vx[q][i], xv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        xh[i][e] = (xv[e] - mu[q]) * rs[q];
                        dg[i][e] += d[i][e] * xh[i][e];
                        db[i][e] += d[i][e];
                        d[i][e] *= ga[i][e];
                        s1 += d[i][e];
                        s2 += d[i][e] * xh[i][e];
                    }
                }
            }
            s1 = wave_sum(s1) / (float)C;
            s2 = wave_sum(s2) / (float)C;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = lane + i * 64;
                if (c < nch) {
                    float o[8], av[8];
                    if (accum) unpack8f(va[q][i], av);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = rs[q] * (d[i][e] - s1 - xh[i][e] * s2) + (accum ? av[e] : 0.f);
                    *reinterpret_cast<u32x4_t*>(dx + r * C + c * 8) = pack8f(o);
                }
            }
        }
    }
    if (!dgamma) return;
    for (int i = 0; i < NCH; ++i) {
        if (i * 64 >= nch) break;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[0][w][lane * 8 + e] = dg[i][e]; red[1][w][lane * 8 + e] = db[i][e]; }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * 512; t += kThreads) {
            const int which = t >> 9, col = t & 511;
            const int c = i * 512 + col;
            if (c < C) {
                float sum = 0.f;
#pragma unroll
                for (int ww = 0; ww < kThreads / 64; ++ww) sum += red[which][ww][col];
                if (part) part[((long)blockIdx.x * 2 + which) * C + c] = sum;
                else atomicAdd((which ? dbeta : dgamma) + set * set_stride + c, sum);
            }
        }
    }
}

// dgamma / dbeta[set * set_stride + c] += sum over the set's blocks of part[block][which][c].  grid (ceil(2 C / 256), nsets, slices):
// slice z sums every gridDim.z-th block (a few dozen loads per thread instead of a thousand in a row) and adds its share atomically
// -- gridDim.z adds per address instead of one per block of the main kernel.
__global__ __launch_bounds__(kThreads) void ln_dgamma_reduce_kernel(const float* __restrict__ part, int blocks_per_set, int C,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, long set_stride) {
    const int t = blockIdx.x * kThreads + threadIdx.x;
    if (t >= 2 * C) return;
    const long set = blockIdx.y;
    const float* src = part + (set * blocks_per_set) * 2 * C + t;
    const int nz = gridDim.z;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = blockIdx.z;
    for (; b + 3 * nz < blocks_per_set; b += 4 * nz) {
        s0 += src[(long)b * 2 * C]; s1 += src[(long)(b + nz) * 2 * C]; s2 += src[(long)(b + 2 * nz) * 2 * C]; s3 += src[(long)(b + 3 * nz) * 2 * C];
    }
    for (; b < blocks_per_set; b += nz) s0 += src[(long)b * 2 * C];
    const int which = t >= C, c = t - which * C;
    atomicAdd((which ? dbeta : dgamma) + set * set_stride + c, (s0 + s1) + (s2 + s3));
}

__global__ __launch_bounds__(kThreads) void quick_gelu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long nchunks) {
    const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= nchunks) return;
    float v[8];
    unpack8f(*reinterpret_cast<const u32x4_t*>(x + idx * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * sigmoid_f(1.702f * v[e]);
    *reinterpret_cast<u32x4_t*>(y + idx * 8) = pack8f(v);
}

// out[r] = sum_d a[r][d] * b[r % rows_b][d]   (delta = rowsum(dO o O) of the attention backward).  EIGHT lanes per row, 16 B each:
// a wave reads 8 whole 128-B rows per instruction (one thread per row had every lane on a line of its own: 2 TB/s).
__global__ __launch_bounds__(kThreads) void rowdot_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                          float* __restrict__ out, long rows, long rows_b, int D) {
    const long t = (long)blockIdx.x * kThreads + threadIdx.x;
    const long r = t >> 3;
    const int sub = (int)(t & 7);
    float s = 0.f;
    if (r < rows) {
        const bf16_t* pa = a + r * D;
        const bf16_t* pb = b + (r % rows_b) * D;
        for (int c = sub * 8; c < D; c += 64) {
            float x[8], y[8];
            unpack8f(*reinterpret_cast<const u32x4_t*>(pa + c), x);
            unpack8f(*reinterpret_cast<const u32x4_t*>(pb + c), y);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x[e] * y[e];
        }
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (r < rows && sub == 0) out[r] = s;
}

// out[r][f] = h[r][f] * gelu(h[r][F + f])
__global__ __launch_bounds__(kThreads) void geglu_fwd_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ out, long rows, int F) {
    const long idx = (long)blockIdx.x * kThreads + threadIdx.x;       // one 8-channel chunk per thread
    const int nch = F >> 3;
    if (idx >= rows * nch) return;
    const long r = idx / nch; const int c = (int)(idx - r * nch);
    float a[8], g[8], o[8];
    unpack8f(*reinterpret_cast<const u32x4_t*>(h + r * 2 * F + c * 8), a);
    unpack8f(*reinterpret_cast<const u32x4_t*>(h + r * 2 * F + F + c * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = a[e] * gelu_f(g[e]);
    *reinterpret_cast<u32x4_t*>(out + r * F + c * 8) = pack8f(o);
}

// dh[r][:F] = dout * gelu(g), dh[r][F:] = dout * a * gelu'(g); saved row = r % rows_x
__global__ __launch_bounds__(kThreads) void geglu_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ h,
                                                             bf16_t* __restrict__ dh, long rows2, long rows_x, int F) {
    const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
    const int nch = F >> 3;
    if (idx >= rows2 * nch) return;
    const long r = idx / nch; const int c = (int)(idx - r * nch);
    const long rx = r % rows_x;
    float a[8], g[8], d[8], oa[8], og[8];
    unpack8f(*reinterpret_cast<const u32x4_t*>(h + rx * 2 * F + c * 8), a);
    unpack8f(*reinterpret_cast<const u32x4_t*>(h + rx * 2 * F + F + c * 8), g);
    unpack8f(*reinterpret_cast<const u32x4_t*>(dout + r * F + c * 8), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float P, ex;
        gelu_parts(g[e], P, ex);
        oa[e] = d[e] * g[e] * P;
        og[e] = d[e] * a[e] * fmaf(g[e] * 0.3989422804014327f, ex, P);
    }
    *reinterpret_cast<u32x4_t*>(dh + r * 2 * F + c * 8) = pack8f(oa);
    *reinterpret_cast<u32x4_t*>(dh + r * 2 * F + F + c * 8) = pack8f(og);
}

// dst[(b*H + h)][s][d] = src[b][s][h*D + d] for s < S, d < D; zero for the padding (s in [S, Sp), d in [D, Dp))
__global__ __launch_bounds__(kThreads) void head_split_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                              int B, int S, int H, int D, int Sp, int Dp) {
    const long idx = (long)blockIdx.x * kThreads + threadIdx.x;       // one 8-element chunk of dst per thread
    const int nch = Dp >> 3;
    const long total = (long)B * H * Sp * nch;
    if (idx >= total) return;
    const int c = (int)(idx % nch);
    long t = idx / nch;
    const int s = (int)(t % Sp); t /= Sp;
    const int h = (int)(t % H); const int b = (int)(t / H);
    u32x4_t o = u32x4_t{0u, 0u, 0u, 0u};
    if (s < S && c * 8 < D) o = *reinterpret_cast<const u32x4_t*>(src + ((long)b * S + s) * H * D + h * D + c * 8);
    *reinterpret_cast<u32x4_t*>(dst + idx * 8) = o;
}

// dst[b][s][h*D + d] = src[(b*H + h)][s][d]
__global__ __launch_bounds__(kThreads) void head_merge_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                              int B, int S, int H, int D, int Sp, int Dp) {
    const long idx = (long)blockIdx.x * kThreads + threadIdx.x;       // one 8-element chunk of dst per thread
    const int nch = D >> 3;
    const long total = (long)B * S * H * nch;
    if (idx >= total) return;
    const int c = (int)(idx % nch);
    long t = idx / nch;
    const int h = (int)(t % H); t /= H;
    const int s = (int)(t % S); const int b = (int)(t / S);
    *reinterpret_cast<u32x4_t*>(dst + idx * 8) =
        *reinterpret_cast<const u32x4_t*>(src + (((long)b * H + h) * Sp + s) * Dp + c * 8);
}

// Rows of up to 64 * 8 * kRowChunks = 4096 elements (ld % 8 == 0): the wave keeps the WHOLE row in registers (16 B per
// lane and load), so the scores are read once and the probabilities written once.  causal_period > 0: row r may
// only see keys k <= r % causal_period (CLIP text encoder).
constexpr int kRowChunks = 8;

template <int NCH>
__global__ __launch_bounds__(kThreads) void softmax_rows_fwd_vec_kernel(const bf16_t* __restrict__ s, bf16_t* __restrict__ p,
                                                                        long rows, int valid, int ld, int causal_period) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    if (causal_period > 0) { const int lim = (int)(r % causal_period) + 1; valid = valid < lim ? valid : lim; }
    const int nch = ld >> 3;
    float x[NCH][8];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            unpack8f(*reinterpret_cast<const u32x4_t*>(s + r * ld + c * 8), x[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (c * 8 + e >= valid) x[i][e] = -INFINITY;
                mx = fmaxf(mx, x[i][e]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (lane + i * 64 < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { x[i][e] = __expf(x[i][e] - mx); sum += x[i][e]; }
        }
    const float inv = 1.f / wave_sum(sum);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[i][e] *= inv;
            *reinterpret_cast<u32x4_t*>(p + r * ld + c * 8) = pack8f(x[i]);
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(kThreads) void softmax_rows_bwd_vec_kernel(const bf16_t* __restrict__ p, const bf16_t* __restrict__ dp,
                                                                        bf16_t* __restrict__ ds, long rows, long p_rows, int valid,
                                                                        int ld, float scale) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const bf16_t* pr = p + (r % p_rows) * ld;
    const bf16_t* dr = dp + r * ld;
    const int nch = ld >> 3;
    float a[NCH][8], d[NCH][8];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            unpack8f(*reinterpret_cast<const u32x4_t*>(pr + c * 8), a[i]);
            unpack8f(*reinterpret_cast<const u32x4_t*>(dr + c * 8), d[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (c * 8 + e >= valid) a[i][e] = 0.f;       // p is zero there already; never trust dp's padding
                dot += a[i][e] * (c * 8 + e < valid ? d[i][e] : 0.f);
            }
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = c * 8 + e < valid ? scale * a[i][e] * (d[i][e] - dot) : 0.f;
            *reinterpret_cast<u32x4_t*>(ds + r * ld + c * 8) = pack8f(o);
        }
    }
}

// p[r][k] = softmax over k < valid of s[r][k]; p[r][k] = 0 for valid <= k < ld.  One wave per row, any length.
__global__ __launch_bounds__(kThreads) void softmax_rows_fwd_kernel(const bf16_t* __restrict__ s, bf16_t* __restrict__ p,
                                                                    long rows, int valid, int ld, int causal_period) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    if (causal_period > 0) { const int lim = (int)(r % causal_period) + 1; valid = valid < lim ? valid : lim; }
    const bf16_t* src = s + r * ld;
    float mx = -INFINITY;
    for (int k = lane; k < valid; k += 64) mx = fmaxf(mx, bf2f(src[k]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int k = lane; k < valid; k += 64) sum += __expf(bf2f(src[k]) - mx);
    const float inv = 1.f / wave_sum(sum);
    for (int k = lane; k < ld; k += 64) p[r * ld + k] = k < valid ? f2bf(__expf(bf2f(src[k]) - mx) * inv) : (bf16_t)0;
}

// ds[r][k] = scale * p[rp][k] * (dp[r][k] - sum_k p*dp), rp = r % p_rows; zero in the padding
__global__ __launch_bounds__(kThreads) void softmax_rows_bwd_kernel(const bf16_t* __restrict__ p, const bf16_t* __restrict__ dp,
                                                                    bf16_t* __restrict__ ds, long rows, long p_rows, int valid,
                                                                    int ld, float scale) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const bf16_t* pr = p + (r % p_rows) * ld;
    const bf16_t* dr = dp + r * ld;
    float dot = 0.f;
    for (int k = lane; k < valid; k += 64) dot += bf2f(pr[k]) * bf2f(dr[k]);
    dot = wave_sum(dot);
    for (int k = lane; k < ld; k += 64)
        ds[r * ld + k] = k < valid ? f2bf(scale * bf2f(pr[k]) * (bf2f(dr[k]) - dot)) : (bf16_t)0;
}

}  // namespace

extern "C" {

// y = LayerNorm(x) * gamma + beta over the last dim; mean / rstd [rows] are saved for the backward.
int siss_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                       long rows, int C, float eps, void* stream) {
    SISS_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * kMaxChunks);
#define LN_DISPATCH(KERNEL, GRID, ...)                                                        \
    do {                                                                                       \
        const int nchunks = cdiv(C / 8, 64);                                                   \
        if (nchunks <= 1) KERNEL<1><<<GRID, kThreads, 0, (hipStream_t)stream>>>(__VA_ARGS__);  \
        else if (nchunks == 2) KERNEL<2><<<GRID, kThreads, 0, (hipStream_t)stream>>>(__VA_ARGS__); \
        else if (nchunks == 3) KERNEL<3><<<GRID, kThreads, 0, (hipStream_t)stream>>>(__VA_ARGS__); \
        else KERNEL<4><<<GRID, kThreads, 0, (hipStream_t)stream>>>(__VA_ARGS__);              \
    } while (0)
    LN_DISPATCH(layernorm_fwd_kernel, cdiv(rows, kThreads / 64), (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, C, eps);
    SISS_LAUNCH_RET();
}

// dy [rows2][C] (rows2 = nsets * set_rows), saved x / mean / rstd of rows_x rows (index r % rows_x);
// dx = LN backward (+ accum if given, same shape as dx); dgamma/dbeta[set * set_stride + c] += column sums (optional).
int siss_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                       const void* accum, void* dx, float* dgamma, float* dbeta, long rows2, long rows_x, long set_rows,
                       long set_stride, int C, void* stream) {
    SISS_CHECK_ARG(dy && x && gamma && mean && rstd && dx && rows2 > 0 && rows_x > 0 && set_rows > 0);
    SISS_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 64 * 8 * kMaxChunks && rows2 % set_rows == 0 && (!dgamma || dbeta));
    // rows per block: a divisor of set_rows (blocks must not straddle sets), at least 32 (8 rows per wave, two in flight) and large
    // enough that the launch has at most ~2048 blocks: every block ends with a fold of its column sums
    long want = rows2 / 2048;
    if (want < 32) want = 32;
    int rpb = (int)(want < set_rows ? want : set_rows);
    while (rpb > 1 && set_rows % rpb) --rpb;
    const int nblocks = cdiv(rows2, rpb);
    // column sums: per-block partials in the library workspace + a per-set reduction (same stream), or atomics without a workspace
    float* part = nullptr;
    if (dgamma) {
        long bytes = 0;
        float* ws = (float*)siss_workspace(&bytes);
        if (ws && (long)nblocks * 2 * C * (long)sizeof(float) <= bytes - 4096) part = ws;
    }
    LN_DISPATCH(layernorm_bwd_kernel, nblocks, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd,
                (const bf16_t*)accum, (bf16_t*)dx, dgamma, dbeta, rows2, rows_x, set_rows, set_stride, C, rpb, part);
    if (part) {
        const int bps = (int)(set_rows / rpb);
        const int slices = bps >= 256 ? 32 : (bps >= 32 ? 8 : 1);
        ln_dgamma_reduce_kernel<<<dim3(cdiv(2 * C, kThreads), (unsigned)(rows2 / set_rows), slices), kThreads, 0, (hipStream_t)stream>>>(
            part, bps, C, dgamma, dbeta, set_stride);
    }
    SISS_LAUNCH_RET();
}

int siss_geglu_fwd(const void* h, void* out, long rows, int F, void* stream) {
    SISS_CHECK_ARG(h && out && rows > 0 && F > 0 && F % 8 == 0);
    geglu_fwd_kernel<<<cdiv(rows * (F / 8), kThreads), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)h, (bf16_t*)out, rows, F);
    SISS_LAUNCH_RET();
}

int siss_geglu_bwd(const void* dout, const void* h, void* dh, long rows2, long rows_x, int F, void* stream) {
    SISS_CHECK_ARG(dout && h && dh && rows2 > 0 && rows_x > 0 && F > 0 && F % 8 == 0);
    geglu_bwd_kernel<<<cdiv(rows2 * (F / 8), kThreads), kThreads, 0, (hipStream_t)stream>>>(
        (const bf16_t*)dout, (const bf16_t*)h, (bf16_t*)dh, rows2, rows_x, F);
    SISS_LAUNCH_RET();
}

// out[r] = <a[r], b[r % rows_b]> over D contiguous bf16 (D % 8 == 0), f32 result
int siss_rowdot(const void* a, const void* b, float* out, long rows, long rows_b, int D, void* stream) {
    SISS_CHECK_ARG(a && b && out && rows > 0 && rows_b > 0 && D > 0 && D % 8 == 0);
    rowdot_kernel<<<cdiv(rows * 8, kThreads), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)a, (const bf16_t*)b, out, rows, rows_b, D);
    SISS_LAUNCH_RET();
}

// y = x * sigmoid(1.702 x) (CLIP's quick_gelu), in place allowed
int siss_quick_gelu(const void* x, void* y, long n, void* stream) {
    SISS_CHECK_ARG(x && y && n > 0 && n % 8 == 0);
    quick_gelu_kernel<<<cdiv(n / 8, kThreads), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)y, n / 8);
    SISS_LAUNCH_RET();
}

// [B][S][H*D] -> [B*H][Sp][Dp], zero padded (Sp >= S, Dp >= D, all of D, Dp multiples of 8)
int siss_head_split(const void* src, void* dst, int B, int S, int H, int D, int Sp, int Dp, void* stream) {
    SISS_CHECK_ARG(src && dst && B > 0 && S > 0 && H > 0 && D > 0 && D % 8 == 0 && Dp % 8 == 0 && Sp >= S && Dp >= D);
    head_split_kernel<<<cdiv((long)B * H * Sp * (Dp / 8), kThreads), kThreads, 0, (hipStream_t)stream>>>(
        (const bf16_t*)src, (bf16_t*)dst, B, S, H, D, Sp, Dp);
    SISS_LAUNCH_RET();
}

// [B*H][Sp][Dp] -> [B][S][H*D]
int siss_head_merge(const void* src, void* dst, int B, int S, int H, int D, int Sp, int Dp, void* stream) {
    SISS_CHECK_ARG(src && dst && B > 0 && S > 0 && H > 0 && D > 0 && D % 8 == 0 && Dp % 8 == 0 && Sp >= S && Dp >= D);
    head_merge_kernel<<<cdiv((long)B * S * H * (D / 8), kThreads), kThreads, 0, (hipStream_t)stream>>>(
        (const bf16_t*)src, (bf16_t*)dst, B, S, H, D, Sp, Dp);
    SISS_LAUNCH_RET();
}

#define SOFTMAX_DISPATCH(KERNEL, ...)                                                                 \
    do {                                                                                              \
        const int nch = cdiv(ld / 8, 64);                                                             \
        const dim3 grid(cdiv(rows, kThreads / 64));                                                   \
        if (nch <= 1) KERNEL<1><<<grid, kThreads, 0, (hipStream_t)stream>>>(__VA_ARGS__);             \
        else if (nch <= 2) KERNEL<2><<<grid, kThreads, 0, (hipStream_t)stream>>>(__VA_ARGS__);        \
        else if (nch <= 4) KERNEL<4><<<grid, kThreads, 0, (hipStream_t)stream>>>(__VA_ARGS__);        \
        else KERNEL<8><<<grid, kThreads, 0, (hipStream_t)stream>>>(__VA_ARGS__);                      \
    } while (0)

// Row softmax over the first `valid` of `ld` columns (the rest is written as zero); s already carries the scale.
// causal_period > 0: row r additionally sees only columns k <= r % causal_period.
int siss_softmax_rows_fwd(const void* s, void* p, long rows, int valid, int ld, int causal_period, void* stream) {
    SISS_CHECK_ARG(s && p && rows > 0 && valid > 0 && ld >= valid && causal_period >= 0);
    if (ld % 8 == 0 && ld <= 64 * 8 * kRowChunks && ((uintptr_t)s | (uintptr_t)p) % 16 == 0) {
        SOFTMAX_DISPATCH(softmax_rows_fwd_vec_kernel, (const bf16_t*)s, (bf16_t*)p, rows, valid, ld, causal_period);
        SISS_LAUNCH_RET();
    }
    softmax_rows_fwd_kernel<<<cdiv(rows, kThreads / 64), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)s, (bf16_t*)p, rows, valid, ld, causal_period);
    SISS_LAUNCH_RET();
}

int siss_softmax_rows_bwd(const void* p, const void* dp, void* ds, long rows, long p_rows, int valid, int ld, float scale,
                          void* stream) {
    SISS_CHECK_ARG(p && dp && ds && rows > 0 && p_rows > 0 && valid > 0 && ld >= valid);
    if (ld % 8 == 0 && ld <= 64 * 8 * kRowChunks && ((uintptr_t)p | (uintptr_t)dp | (uintptr_t)ds) % 16 == 0) {
        SOFTMAX_DISPATCH(softmax_rows_bwd_vec_kernel, (const bf16_t*)p, (const bf16_t*)dp, (bf16_t*)ds, rows, p_rows, valid, ld, scale);
        SISS_LAUNCH_RET();
    }
    softmax_rows_bwd_kernel<<<cdiv(rows, kThreads / 64), kThreads, 0, (hipStream_t)stream>>>(
        (const bf16_t*)p, (const bf16_t*)dp, (bf16_t*)ds, rows, p_rows, valid, ld, scale);
    SISS_LAUNCH_RET();
}

}  // extern "C"
