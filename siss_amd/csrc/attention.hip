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

}  // extern "C"
