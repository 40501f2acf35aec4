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
constexpr int kMaxM = 4096;     // sanity cap on the batch rows (the kernels walk rows in chunks)

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

// block = (row m, 64 columns k); its four waves split the reduction over n and meet in LDS (one thread per (m, k)
// walking all of n left the chip to 64 blocks of serial L2 round trips: 230 us for 4 MB of traffic)
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                                            const float* __restrict__ W, float* __restrict__ dx, int M2,
                                                            int Mx, int N, int K, int accumulate) {
    __shared__ float part[4][64];
    const int m = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    const int n0 = (int)((long)N * w / 4), n1 = (int)((long)N * (w + 1) / 4);
    float acc = 0.f;
    if (k < K)
        for (int n = n0; n < n1; ++n) acc += dy_eff(dy, yact, m, n, N, Mx) * W[(long)n * K + k];
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && k < K) {
        const float t = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
        const long i = (long)m * K + k;
        dx[i] = accumulate ? dx[i] + t : t;
    }
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


// ---- all time_emb_proj layers of the network as ONE problem --------------------------------------
// Output column n of the concatenated [M][Ntot] matrix belongs to some resnet; woff[n] is the offset
// (floats, into the flat parameter buffer) of its weight ROW, boff[n] of its bias element.
__global__ __launch_bounds__(kThreads) void multi_fwd_kernel(const float* __restrict__ x, const float* __restrict__ P,
                                                             const long* __restrict__ woff, const long* __restrict__ boff,
                                                             float* __restrict__ y, int M, int Ntot, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (n >= Ntot) return;
    const float* W = P + woff[n];
    for (int m0 = 0; m0 < M; m0 += 8) {
        float acc[8] = {};
        for (int k = lane; k < K; k += 64) {
            const float w = W[k];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (m0 + i < M) acc[i] += silu_f(x[(long)(m0 + i) * K + k]) * w;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float s = wave_sum(acc[i]);
            if (lane == 0 && m0 + i < M) y[(long)(m0 + i) * Ntot + n] = s + P[boff[n]];
        }
    }
}

// dW[set][woff[n] + k] += sum_{m in set} dy[m][n] * silu(x[m % Mx][k]) ; db / db2 likewise.
// grid (K / 256, Ntot / 32, nsets); thread = one k, 32 columns per block: silu(x) is evaluated once per (row, k) and
// block -- not once per (row, k, column) -- and the 32 x 32 cotangent tile of each row chunk sits in LDS (broadcast reads).
constexpr int kDwCols = 32, kDwRows = 32;
__global__ __launch_bounds__(256) void multi_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ G,
                                                           const long* __restrict__ woff, const long* __restrict__ boff,
                                                           const long* __restrict__ boff2, int Mx, int set_rows, long set_stride,
                                                           int Ntot, int K) {
    __shared__ float sdy[kDwRows][kDwCols + 1];
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n0 = blockIdx.y * kDwCols, set = blockIdx.z;
    const int ncols = Ntot - n0 < kDwCols ? Ntot - n0 : kDwCols;
    float acc[kDwCols];
#pragma unroll
    for (int c = 0; c < kDwCols; ++c) acc[c] = 0.f;
    float bsum = 0.f;                                       // threads < ncols of the blocks with blockIdx.x == 0
    for (int j0 = 0; j0 < set_rows; j0 += kDwRows) {
        const int nr = set_rows - j0 < kDwRows ? set_rows - j0 : kDwRows;
        __syncthreads();
        for (int i = threadIdx.x; i < kDwRows * kDwCols; i += 256) {
            const int r = i / kDwCols, c = i - r * kDwCols;
            sdy[r][c] = (r < nr && c < ncols) ? dy[(long)(set * set_rows + j0 + r) * Ntot + n0 + c] : 0.f;
        }
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x < ncols)
            for (int r = 0; r < nr; ++r) bsum += sdy[r][threadIdx.x];
        if (k < K) {
            for (int r = 0; r < nr; ++r) {
                const float sx = silu_f(x[(long)((set * set_rows + j0 + r) % Mx) * K + k]);
#pragma unroll
                for (int c = 0; c < kDwCols; ++c) acc[c] += sdy[r][c] * sx;
            }
        }
    }
    float* g = G + (long)set * set_stride;
    if (k < K) {
#pragma unroll
        for (int c = 0; c < kDwCols; ++c)
            if (c < ncols) g[woff[n0 + c] + k] += acc[c];
    }
    if (blockIdx.x == 0 && threadIdx.x < ncols) {
        g[boff[n0 + threadIdx.x]] += bsum;
        g[boff2[n0 + threadIdx.x]] += bsum;
    }
}

// dx[m][k] += sum_n dy[m][n] * W_n[k]     grid: (K / 256, M2, column splits); thread = one (m, k) output over one split
// of kDxSplit columns: the cotangent dy[m][n] and the row offset woff[n] are block-uniform (scalar loads), W_n[k] is one
// coalesced row read, and each thread ends with ONE atomic -- Ntot / kDxSplit-way contention instead of Ntot / 64
// (the previous form, 64 atomics per thread from 160 column blocks into one [M2][K] array, spent 143 us on them).
constexpr int kDxSplit = 512;
__global__ __launch_bounds__(256) void multi_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ P,
                                                           const long* __restrict__ woff, float* __restrict__ dx,
                                                           int M2, int Ntot, int K) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y;
    const int n0 = blockIdx.z * kDxSplit;
    const int n1 = n0 + kDxSplit < Ntot ? n0 + kDxSplit : Ntot;
    if (k >= K) return;
    const float* d = dy + (long)m * Ntot;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int n = n0;
    for (; n + 4 <= n1; n += 4) {
        a0 += d[n] * P[woff[n] + k];
        a1 += d[n + 1] * P[woff[n + 1] + k];
        a2 += d[n + 2] * P[woff[n + 2] + k];
        a3 += d[n + 3] * P[woff[n + 3] + k];
    }
    for (; n < n1; ++n) a0 += d[n] * P[woff[n] + k];
    atomicAdd(dx + (long)m * K + k, (a0 + a1) + (a2 + a3));
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
    if (dx) linear_bwd_dx_kernel<<<dim3(cdiv(K, 64), M2), 256, 0, st>>>(dy, yact, W, dx, M2, Mx, N, K, accumulate_dx);
    const long tot = (long)(M2 / set_rows) * N * K;
    linear_bwd_dw_kernel<<<cdiv(tot, 256), 256, 0, st>>>(dy, yact, x, dW, db, db2, M2, Mx, set_rows, set_stride_w, set_stride_b, N, K, act_in_silu);
    SISS_LAUNCH_RET();
}

// y[m][n] = silu(x[m]) . P[woff[n] .. +K] + P[boff[n]]   for all Ntot concatenated output columns
int siss_linear_multi_fwd(const float* x, const float* params, const long* woff, const long* boff, float* y, int M,
                          int Ntot, int K, void* stream) {
    SISS_CHECK_ARG(x && params && woff && boff && y && M > 0 && M <= kMaxM && Ntot > 0 && K > 0);
    multi_fwd_kernel<<<cdiv(Ntot, kThreads / 64), kThreads, 0, (hipStream_t)stream>>>(x, params, woff, boff, y, M, Ntot, K);
    SISS_LAUNCH_RET();
}

// Backward of the above for M2 cotangent rows (row m uses saved input row m % Mx):
// grads[set][woff[n]+k], grads[set][boff[n]], grads[set][boff2[n]] are accumulated (+=); dx[M2][K] is
// accumulated atomically (zero it first).
int siss_linear_multi_bwd(const float* dy, const float* x, const float* params, float* grads, const long* woff,
                          const long* boff, const long* boff2, float* dx, int M2, int Mx, int set_rows,
                          long set_stride, int Ntot, int K, void* stream) {
    SISS_CHECK_ARG(dy && x && params && grads && woff && boff && boff2 && dx);
    SISS_CHECK_ARG(M2 > 0 && M2 <= kMaxM && Mx > 0 && set_rows > 0 && M2 % set_rows == 0 && Ntot > 0 && K > 0);
    hipStream_t st = (hipStream_t)stream;
    const dim3 gw(cdiv(K, 256), cdiv(Ntot, kDwCols), M2 / set_rows);
    multi_bwd_dw_kernel<<<gw, 256, 0, st>>>(dy, x, grads, woff, boff, boff2, Mx, set_rows, set_stride, Ntot, K);
    dim3 grid(cdiv(K, 256), M2, cdiv(Ntot, kDxSplit));
    multi_bwd_dx_kernel<<<grid, 256, 0, st>>>(dy, params, woff, dx, M2, Ntot, K);
    SISS_LAUNCH_RET();
}

}  // extern "C"
