// Time-embedding path: sinusoidal embedding and the small f32 linears
// (M = batch <= 64 rows).  GEMMs with M = 16 are far below an MFMA tile; these are plain
// VALU kernels, W rows read coalesced, x/dy served from cache.
//   fwd : y[m][n] = sum_k act(x[m][k]) * W[n][k] + b[n]                act = SiLU or identity
//   bwd : dy_eff[m][n] = dy[m][n] * dsilu(yact[m % Mx][n])            (when the NEXT op applied SiLU to y)
//         dx[m][k] (+)= sum_n dy_eff[m][n] W[n][k]
//         dW[set][n][k] += sum_{m in set} dy_eff[m][n] act(x[m % Mx][k]) ; db[set][n] += sum dy_eff
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxM = 64;

__global__ void sincos_kernel(const int64_t* __restrict__ t, float* __restrict__ out, int B, int dim, int flip,
                              float freq_shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= B * half) return;
    const int b = i / half, j = i - b * half;
    const float freq = expf(-logf(10000.f) * (float)j / ((float)half - freq_shift));
    const float arg = (float)t[b] * freq;
    const float s = sinf(arg), c = cosf(arg);
    float* o = out + (long)b * dim;
    if (flip) { o[j] = c; o[half + j] = s; }
    else { o[j] = s; o[half + j] = c; }
}

// one wave per output column n; lanes stride over K
__global__ __launch_bounds__(kThreads) void linear_fwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ W,
                                                              const float* __restrict__ b, float* __restrict__ y,
                                                              int M, int N, int K, int act_in) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (n >= N) return;
    for (int m0 = 0; m0 < M; m0 += 8) {
        float acc[8] = {};
        for (int k = lane; k < K; k += 64) {
            const float w = W[(long)n * K + k];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (m0 + i < M) {
                    float xv = x[(long)(m0 + i) * K + k];
                    if (act_in) xv = silu_f(xv);
                    acc[i] += xv * w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float s = wave_sum(acc[i]);
            if (lane == 0 && m0 + i < M) y[(long)(m0 + i) * N + n] = s + b[n];
        }
    }
}

__device__ __forceinline__ float dy_eff(const float* dy, const float* yact, int m, int n, int N, int Mx) {
    float d = dy[(long)m * N + n];
    if (yact) d *= dsilu_f(yact[(long)(m % Mx) * N + n]);
    return d;
}

// one thread per (m, k)
__global__ void linear_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                     const float* __restrict__ W, float* __restrict__ dx, int M2, int Mx, int N,
                                     int K, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M2 * K) return;
    const int m = i / K, k = i - (long)m * K;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += dy_eff(dy, yact, m, n, N, Mx) * W[(long)n * K + k];
    dx[i] = accumulate ? dx[i] + acc : acc;
}

// one thread per (set, n, k); also db when k == 0
__global__ void linear_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                     const float* __restrict__ x, float* __restrict__ dW, float* __restrict__ db,
                                     float* __restrict__ db2, int M2, int Mx, int set_rows, long set_stride_w, long set_stride_b, int N,
                                     int K, int act_in) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nsets = M2 / set_rows;
    if (i >= (long)nsets * N * K) return;
    const int set = i / ((long)N * K);
    const long rem = i - (long)set * N * K;
    const int n = rem / K, k = rem - (long)n * K;
    float acc = 0.f, bsum = 0.f;
    for (int j = 0; j < set_rows; ++j) {
        const int m = set * set_rows + j;
        const float d = dy_eff(dy, yact, m, n, N, Mx);
        float xv = x[(long)(m % Mx) * K + k];
        if (act_in) xv = silu_f(xv);
        acc += d * xv;
        bsum += d;
    }
    dW[(long)set * set_stride_w + (long)n * K + k] += acc;
    if (k == 0) {
        db[(long)set * set_stride_b + n] += bsum;
        if (db2) db2[(long)set * set_stride_b + n] += bsum;
    }
}

}  // namespace

extern "C" {

int siss_timestep_sincos(const int64_t* t, float* out, int B, int dim, int flip_sin_to_cos, float freq_shift, void* stream) {
    SISS_CHECK_ARG(t && out && B > 0 && dim > 0 && dim % 2 == 0);
    sincos_kernel<<<cdiv((long)B * dim / 2, 128), 128, 0, (hipStream_t)stream>>>(t, out, B, dim, flip_sin_to_cos, freq_shift);
    SISS_LAUNCH_RET();
}

int siss_linear_small_fwd(const float* x, const float* W, const float* b, float* y, int M, int N, int K, int act_in_silu, void* stream) {
    SISS_CHECK_ARG(x && W && b && y && M > 0 && M <= kMaxM && N > 0 && K > 0);
    linear_fwd_kernel<<<cdiv(N, kThreads / 64), kThreads, 0, (hipStream_t)stream>>>(x, W, b, y, M, N, K, act_in_silu);
    SISS_LAUNCH_RET();
}

// dx may be NULL (first layer).  dW/db are accumulated in place (+=): zero them at step start.
// db2 (optional) receives the same bias sums (conv1 bias shares its gradient with time_emb_proj bias).
int siss_linear_small_bwd(const float* dy, const float* yact, const float* x, const float* W, float* dx, int accumulate_dx,
                          float* dW, float* db, float* db2, int M2, int Mx, int set_rows, long set_stride_w,
                          long set_stride_b, int N, int K, int act_in_silu, void* stream) {
    SISS_CHECK_ARG(dy && x && W && dW && db && M2 > 0 && Mx > 0 && set_rows > 0 && M2 % set_rows == 0 && N > 0 && K > 0);
    hipStream_t st = (hipStream_t)stream;
    if (dx) linear_bwd_dx_kernel<<<cdiv((long)M2 * K, 256), 256, 0, st>>>(dy, yact, W, dx, M2, Mx, N, K, accumulate_dx);
    const long tot = (long)(M2 / set_rows) * N * K;
    linear_bwd_dw_kernel<<<cdiv(tot, 256), 256, 0, st>>>(dy, yact, x, dW, db, db2, M2, Mx, set_rows, set_stride_w, set_stride_b, N, K, act_in_silu);
    SISS_LAUNCH_RET();
}

}  // extern "C"
