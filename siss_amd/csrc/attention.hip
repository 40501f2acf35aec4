// Row softmax forward / backward for the single-head spatial attention block.
// QK^T, PV and the projections run on the MFMA GEMM kernels (gemm_nt / gemm_tn); the score tile
// per sample is small (S x S, S = 256 or 64), so softmax is a separate wave-per-row pass in f32
// (upcast_softmax=True in the reference's attention block).
#include "common.h"

namespace {

constexpr int kThreads = 256;

// p[r][:] = softmax(s[r][:]) ; s already carries the 1/sqrt(d) scale (gemm alpha)
__global__ __launch_bounds__(kThreads) void softmax_fwd_kernel(const bf16_t* __restrict__ s,
                                                               bf16_t* __restrict__ p, long rows, int S) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const bf16_t* src = s + r * S;
    float v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = lane + i * 64;
        v[i] = k < S ? bf2f(src[k]) : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = (lane + i * 64 < S) ? __expf(v[i] - mx) : 0.f; sum += v[i]; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = lane + i * 64;
        if (k < S) p[r * S + k] = f2bf(v[i] * inv);
    }
}

// ds[r][k] = scale * p[rp][k] * (dp[r][k] - sum_k p*dp),  rp = r % p_rows (dual cotangent sets)
__global__ __launch_bounds__(kThreads) void softmax_bwd_kernel(const bf16_t* __restrict__ p,
                                                               const bf16_t* __restrict__ dp,
                                                               bf16_t* __restrict__ ds, long rows, long p_rows,
                                                               int S, float scale) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const bf16_t* pr = p + (r % p_rows) * S;
    const bf16_t* dr = dp + r * S;
    float pv[16], dv[16], dot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = lane + i * 64;
        pv[i] = k < S ? bf2f(pr[k]) : 0.f;
        dv[i] = k < S ? bf2f(dr[k]) : 0.f;
        dot += pv[i] * dv[i];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = lane + i * 64;
        if (k < S) ds[r * S + k] = f2bf(scale * pv[i] * (dv[i] - dot));
    }
}


// ---- multi-head attention with a SMALL head dim (diffusers default attention_head_dim = 8: MNIST UNet) -------
// d = 8 is far below an MFMA tile; one block per (sample, head) keeps the head's K / V (and Q, dO, O in the
// backward) in LDS as f32 and runs an online-softmax row per thread.  Tokens are compact [N][S][C], C = heads*D.
template <int D, typename T>
__global__ __launch_bounds__(kThreads) void mha_small_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                 const T* __restrict__ v, T* __restrict__ o,
                                                                 float* __restrict__ lse, int S, int C, float scale) {
    extern __shared__ float sh[];          // K [S][D], V [S][D]
    float* sk = sh; float* sv = sh + S * D;
    const int h = blockIdx.x, b = blockIdx.y, heads = C / D;
    const long base = (long)b * S * C + h * D;
    for (int i = threadIdx.x; i < S * D; i += kThreads) {
        const int r = i / D, c = i - r * D;
        sk[i] = to_f(k[base + (long)r * C + c]);
        sv[i] = to_f(v[base + (long)r * C + c]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S; i += kThreads) {
        float qi[D], acc[D];
#pragma unroll
        for (int c = 0; c < D; ++c) { qi[c] = to_f(q[base + (long)i * C + c]) * scale; acc[c] = 0.f; }
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j < S; ++j) {
            float sc = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) sc += qi[c] * sk[j * D + c];
            const float mn = fmaxf(m, sc), corr = __expf(m - mn), p = __expf(sc - mn);
            l = l * corr + p;
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] = acc[c] * corr + p * sv[j * D + c];
            m = mn;
        }
        const float inv = 1.f / l;
#pragma unroll
        for (int c = 0; c < D; ++c) o[base + (long)i * C + c] = from_f<T>(acc[c] * inv);
        lse[((long)b * heads + h) * S + i] = m + __logf(l);
    }
}

// n2 cotangent samples against nx saved samples (saved index = n2 % nx)
template <int D, typename T>
__global__ __launch_bounds__(kThreads) void mha_small_bwd_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
    const T* __restrict__ o, const float* __restrict__ lse, const T* __restrict__ dout,
    T* __restrict__ dq, T* __restrict__ dk, T* __restrict__ dv, int nx, int S, int C, float scale) {
    extern __shared__ float sh[];          // Q, K, V, dO [S][D] each, delta [S], lse [S]
    float* sq = sh; float* sk = sq + S * D; float* sv = sk + S * D; float* sd = sv + S * D;
    float* sdel = sd + S * D; float* sl = sdel + S;
    const int h = blockIdx.x, n2 = blockIdx.y, n = n2 % nx, heads = C / D;
    const long bs = (long)n * S * C + h * D, bd = (long)n2 * S * C + h * D;
    for (int i = threadIdx.x; i < S * D; i += kThreads) {
        const int r = i / D, c = i - r * D;
        sq[i] = to_f(q[bs + (long)r * C + c]); sk[i] = to_f(k[bs + (long)r * C + c]);
        sv[i] = to_f(v[bs + (long)r * C + c]); sd[i] = to_f(dout[bd + (long)r * C + c]);
    }
    for (int i = threadIdx.x; i < S; i += kThreads) {
        float t = 0.f;
        for (int c = 0; c < D; ++c) t += to_f(dout[bd + (long)i * C + c]) * to_f(o[bs + (long)i * C + c]);
        sdel[i] = t;
        sl[i] = lse[((long)n * heads + h) * S + i];
    }
    __syncthreads();
    // dQ: thread per query row
    for (int i = threadIdx.x; i < S; i += kThreads) {
        float acc[D];
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] = 0.f;
        for (int j = 0; j < S; ++j) {
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) { sc += sq[i * D + c] * sk[j * D + c]; dp += sd[i * D + c] * sv[j * D + c]; }
            const float p = __expf(sc * scale - sl[i]);
            const float ds = p * (dp - sdel[i]) * scale;
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] += ds * sk[j * D + c];
        }
#pragma unroll
        for (int c = 0; c < D; ++c) dq[bd + (long)i * C + c] = from_f<T>(acc[c]);
    }
    // dK, dV: thread per key row
    for (int j = threadIdx.x; j < S; j += kThreads) {
        float ak[D], av[D];
#pragma unroll
        for (int c = 0; c < D; ++c) { ak[c] = 0.f; av[c] = 0.f; }
        for (int i = 0; i < S; ++i) {
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) { sc += sq[i * D + c] * sk[j * D + c]; dp += sd[i * D + c] * sv[j * D + c]; }
            const float p = __expf(sc * scale - sl[i]);
            const float ds = p * (dp - sdel[i]) * scale;
#pragma unroll
            for (int c = 0; c < D; ++c) { ak[c] += ds * sq[i * D + c]; av[c] += p * sd[i * D + c]; }
        }
#pragma unroll
        for (int c = 0; c < D; ++c) { dk[bd + (long)j * C + c] = from_f<T>(ak[c]); dv[bd + (long)j * C + c] = from_f<T>(av[c]); }
    }
}

}  // namespace

extern "C" {

int siss_softmax_fwd(const void* s, void* p, long rows, int S, void* stream) {
    SISS_CHECK_ARG(s && p && rows > 0 && S > 0 && S <= 1024);
    softmax_fwd_kernel<<<cdiv(rows, kThreads / 64), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)s, (bf16_t*)p, rows, S);
    SISS_LAUNCH_RET();
}
int siss_softmax_bwd(const void* p, const void* dp, void* ds, long rows, long p_rows, int S, float scale, void* stream) {
    SISS_CHECK_ARG(p && dp && ds && rows > 0 && p_rows > 0 && S > 0 && S <= 1024);
    softmax_bwd_kernel<<<cdiv(rows, kThreads / 64), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)p, (const bf16_t*)dp, (bf16_t*)ds, rows, p_rows, S, scale);
    SISS_LAUNCH_RET();
}

#define MHA_DISPATCH(D, CALL)            \
    switch (D) {                         \
        case 8: { constexpr int kD = 8; CALL; } break;   \
        case 16: { constexpr int kD = 16; CALL; } break; \
        case 32: { constexpr int kD = 32; CALL; } break; \
        default: return SISS_ERR_ARG;    \
    }

// o = softmax(q k^T * scale) v per (sample, head); q/k/v/o compact [N][S][C] bf16, C = heads * D; lse [N][heads][S] f32
int siss_mha_small_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int N, int S, int C, int D,
                       float scale, void* stream) {
    SISS_CHECK_ARG(q && k && v && o && lse && N > 0 && S > 0 && C > 0 && D > 0 && C % D == 0);
    SISS_CHECK_ARG((long)S * D * 2 * 4 <= 64 * 1024 && N <= 65535);
    dim3 grid(C / D, N);
    const size_t smem = (size_t)S * D * 2 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    MHA_DISPATCH(D, (mha_small_fwd_kernel<kD, bf16_t><<<grid, kThreads, smem, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, S, C, scale)));
    SISS_LAUNCH_RET();
}

// dq/dk/dv [n2][S][C] from dout [n2][S][C] and the saved q/k/v/o/lse of nx samples (saved index = n2 % nx)
int siss_mha_small_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dout,
                       void* dq, void* dk, void* dv, int n2, int nx, int S, int C, int D, float scale, void* stream) {
    SISS_CHECK_ARG(q && k && v && o && lse && dout && dq && dk && dv && n2 > 0 && nx > 0 && S > 0 && C % D == 0);
    SISS_CHECK_ARG(((long)S * D * 4 + 2 * S) * 4 <= 64 * 1024 && n2 <= 65535);
    dim3 grid(C / D, n2);
    const size_t smem = ((size_t)S * D * 4 + 2 * S) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    MHA_DISPATCH(D, (mha_small_bwd_kernel<kD, bf16_t><<<grid, kThreads, smem, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, lse, (const bf16_t*)dout, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, nx, S, C, scale)));
    SISS_LAUNCH_RET();
}


// The same with f32 tokens (the f32 parity mode: f32_path.hip)
int siss_mha_small_fwd_f32(const void* q, const void* k, const void* v, void* o, float* lse, int N, int S, int C, int D,
                           float scale, void* stream) {
    SISS_CHECK_ARG(q && k && v && o && lse && N > 0 && S > 0 && C > 0 && D > 0 && C % D == 0);
    SISS_CHECK_ARG((long)S * D * 2 * 4 <= 64 * 1024 && N <= 65535);
    dim3 grid(C / D, N);
    const size_t smem = (size_t)S * D * 2 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    MHA_DISPATCH(D, (mha_small_fwd_kernel<kD, float><<<grid, kThreads, smem, st>>>((const float*)q, (const float*)k, (const float*)v, (float*)o, lse, S, C, scale)));
    SISS_LAUNCH_RET();
}
int siss_mha_small_bwd_f32(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dout,
                           void* dq, void* dk, void* dv, int n2, int nx, int S, int C, int D, float scale, void* stream) {
    SISS_CHECK_ARG(q && k && v && o && lse && dout && dq && dk && dv && n2 > 0 && nx > 0 && S > 0 && C % D == 0);
    SISS_CHECK_ARG(((long)S * D * 4 + 2 * S) * 4 <= 64 * 1024 && n2 <= 65535);
    dim3 grid(C / D, n2);
    const size_t smem = ((size_t)S * D * 4 + 2 * S) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    MHA_DISPATCH(D, (mha_small_bwd_kernel<kD, float><<<grid, kThreads, smem, st>>>((const float*)q, (const float*)k, (const float*)v, (const float*)o, lse, (const float*)dout, (float*)dq, (float*)dk, (float*)dv, nx, S, C, scale)));
    SISS_LAUNCH_RET();
}

}  // extern "C"
