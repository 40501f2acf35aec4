// conv_out (Cout = in/out image channels, 1..4): 3x3 convolution with a tiny output-channel
// count.  N = 3 would waste 97 % of a 128-wide MFMA tile, and the op is HBM-bound anyway
// (it reads a C=128 activation to produce 3 channels), so these are direct VALU kernels:
//   fprop : padded NHWC bf16 [B][H+2][W+2][C]  -> pred NCHW f32 [B][CO][H][W]
//   dgrad : cotangent NCHW f32 [N2][CO][H][W]  -> dX padded NHWC bf16 [N2][..][C]
//   wgrad : dW[set][tap][co][ci] += sum c * x ;  dbias[set][co] += sum c
// Weight layout: native [9][CO][C] f32 (tap = ky*3+kx).  C/8 lanes cover one pixel row
// (16 B per lane, coalesced); partial dot products are folded with wave shuffles.
#include "common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ void unpack8(u32x4_t r, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = __builtin_bit_cast(float, r[j] << 16);
        v[2 * j + 1] = __builtin_bit_cast(float, r[j] & 0xffff0000u);
    }
}

template <int CO>
__global__ __launch_bounds__(kThreads) void conv_out_fprop_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ pred, int B, int H, int W, int C) {
    extern __shared__ float shw[];   // [9][CO][C]
    for (int i = threadIdx.x; i < 9 * CO * C; i += kThreads) shw[i] = w[i];
    __syncthreads();
    const int lpp = C / 8, ppb = kThreads / lpp;
    const int slot = threadIdx.x / lpp, cc = threadIdx.x - slot * lpp;
    const long npix = (long)B * H * W;
    const int Wp = W + 2;
    for (long pbase = (long)blockIdx.x * ppb; pbase < npix; pbase += (long)gridDim.x * ppb) {
        const long p = pbase + slot;
        const bool ok = p < npix;
        const long pc = ok ? p : npix - 1;
        const int xx = pc % W; long t = pc / W;
        const int yy = t % H; const int n = t / H;
        const long row = ((long)n * (H + 2) + yy + 1) * Wp + xx + 1;
        float acc[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const long r = row + (tap / 3 - 1) * Wp + (tap % 3 - 1);
            float v[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(x + r * C + cc * 8), v);
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                const float* ww = shw + (tap * CO + o) * C + cc * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[o] += v[e] * ww[e];
            }
        }
        for (int off = lpp >> 1; off > 0; off >>= 1)
#pragma unroll
            for (int o = 0; o < CO; ++o) acc[o] += __shfl_xor(acc[o], off, 64);
        if (ok && cc == 0) {
#pragma unroll
            for (int o = 0; o < CO; ++o) pred[(((long)n * CO + o) * H + yy) * W + xx] = acc[o] + bias[o];
        }
    }
}

// Any C % 8 == 0 (the SD UNet's 320-channel head): one WAVE per pixel, lane l owns the 8-channel chunks
// l, l+64, ...; the 64 partial dot products meet in a full-wave butterfly.
template <int CO>
__global__ __launch_bounds__(kThreads) void conv_out_fprop_wave_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ pred, int B, int H, int W, int C) {
    extern __shared__ float shw[];   // [9][CO][C]
    for (int i = threadIdx.x; i < 9 * CO * C; i += kThreads) shw[i] = w[i];
    __syncthreads();
    const int nch = C / 8, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long npix = (long)B * H * W;
    const int Wp = W + 2;
    for (long p = (long)blockIdx.x * (kThreads / 64) + wv; p < npix; p += (long)gridDim.x * (kThreads / 64)) {
        const int xx = p % W; long t = p / W;
        const int yy = t % H; const int n = t / H;
        const long row = ((long)n * (H + 2) + yy + 1) * Wp + xx + 1;
        float acc[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = 0.f;
        for (int cc = lane; cc < nch; cc += 64) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const long r = row + (tap / 3 - 1) * Wp + (tap % 3 - 1);
                float v[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(x + r * C + cc * 8), v);
#pragma unroll
                for (int o = 0; o < CO; ++o) {
                    const float* ww = shw + (tap * CO + o) * C + cc * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[o] += v[e] * ww[e];
                }
            }
        }
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = wave_sum(acc[o]);
        if (lane == 0) {
#pragma unroll
            for (int o = 0; o < CO; ++o) pred[(((long)n * CO + o) * H + yy) * W + xx] = acc[o] + bias[o];
        }
    }
}

// MFMA form for C % 32 == 0 (every UNet here): the op reads a C-channel activation to produce <= 4 channels, so it
// must run at the HBM rate of that one read.  One wave = 16 consecutive pixels of an image row; the weights are the
// MFMA A operand (rows = output channels, zero above CO), the activations the B operand straight from global
// memory (lane: pixel l&15, 8 channels at 32*kc + 8*(l>>4): one 16-B load per lane and (tap, kc)); the nine taps
// re-read neighbouring rows through L1/L2.  Weights are rounded to bf16 like every other convolution of the path
// (the reference's autocast runs conv_out in bf16 as well).
template <int CO>
__global__ __launch_bounds__(kThreads) void conv_out_fprop_mfma_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ pred, int B, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* sw = reinterpret_cast<bf16_t*>(smem_raw);      // [9][CO][C] bf16
    for (int i = threadIdx.x; i < 9 * CO * C; i += kThreads) sw[i] = f2bf(w[i]);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int px = lane & 15, kg = lane >> 4;
    const int segs = (W + 15) >> 4, Wp = W + 2, kchunks = C >> 5;
    const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
    // Work item = a band of kBand image rows x one 16-pixel column segment, walked top to bottom by ONE wave: rows
    // y-1 / y / y+1 of consecutive steps overlap, so each input row is fetched from HBM once and re-read from
    // L1 / L2 (the first version strode whole-image segments over the grid and pulled 6.6x the activation bytes
    // from HBM: profiles/r01j_hbm_traffic.json).  Items are handed out XCD-contiguously (blocks b, b+8, ... share
    // an L2), neighbouring column segments to the four waves of a block.
    constexpr int kBand = 8;
    const int bands = (H + kBand - 1) / kBand;
    const long nitems = (long)B * bands * segs;
    const long nblk = gridDim.x;
    long bid = blockIdx.x;
    { const long q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k; }
    for (long item = bid * (kThreads / 64) + wv; item < nitems; item += nblk * (kThreads / 64)) {
      const int sx = item % segs; long tt = item / segs;
      const int band = tt % bands; const int n = tt / bands;
      const int y0 = band * kBand;
      const int y_end = y0 + kBand < H ? y0 + kBand : H;
      const int xx = sx * 16 + px;
      const bool ok = xx < W;
      const int xc = ok ? xx : W - 1;
      // The wave walks the INPUT rows r = y0-1 .. y_end of its band (padded rows: the halo supplies the zeros).  Input
      // row r feeds output row r+1 through filter row 0, r through row 1 and r-1 through row 2, so its twelve
      // fragments (3 kx x 4 channel chunks) are loaded ONCE and used by 36 MFMAs into three rotating accumulators:
      // a third of the L1 traffic of the per-output-row form (which re-read every input row for each ky and ran at
      // 1/5 of the HBM rate).
      f32x4_t accP = {0.f, 0.f, 0.f, 0.f}, accC = accP, accN = accP;      // output rows r-1, r, r+1
      for (int r = y0 - 1; r <= y_end; ++r) {
        const bf16_t* base = x + (((long)n * (H + 2) + r + 1) * Wp + xc + 1) * C + kg * 8;
        for (int kc0 = 0; kc0 < kchunks; kc0 += 4) {
            bf16x8_t bf[3][4];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bf[kx][j] = kc0 + j < kchunks ? *reinterpret_cast<const bf16x8_t*>(base + (long)(kx - 1) * C + (kc0 + j) * 32) : zero;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {       // innermost: three independent accumulator chains
                        const bf16_t* ap = sw + ((ky * 3 + kx) * CO + (px < CO ? px : 0)) * C + kg * 8;
                        bf16x8_t a = kc0 + j < kchunks ? *reinterpret_cast<const bf16x8_t*>(ap + (kc0 + j) * 32) : zero;
                        if (px >= CO) a = zero;
                        if (ky == 0) accN = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[kx][j], accN, 0, 0, 0);
                        else if (ky == 1) accC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[kx][j], accC, 0, 0, 0);
                        else accP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[kx][j], accP, 0, 0, 0);
                    }
        }
        // output row r-1 has now seen its three input rows; acc[q] = output channel 4*(lane>>4) + q of pixel lane&15
        const int yo = r - 1;
        if (yo >= y0 && yo < y_end && kg == 0 && ok) {
#pragma unroll
            for (int q = 0; q < CO; ++q) pred[(((long)n * CO + q) * H + yo) * W + xx] = accP[q] + bias[q];
        }
        accP = accC; accC = accN; accN = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
}

template <int CO>
__global__ __launch_bounds__(kThreads) void conv_out_dgrad_kernel(
    const float* __restrict__ c, const float* __restrict__ w, bf16_t* __restrict__ dx, int N2, int H, int W,
    int C) {
    extern __shared__ float shw[];   // [9][CO][C]
    for (int i = threadIdx.x; i < 9 * CO * C; i += kThreads) shw[i] = w[i];
    __syncthreads();
    const int lpp = C / 8;
    const long total = (long)N2 * H * W * lpp;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const int cc = i % lpp; long t = i / lpp;
        const int xx = t % W; t /= W;
        const int yy = t % H; const int n = t / H;
        float acc[8] = {};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // x[y,x] feeds out[y - (ky-1), x - (kx-1)] through tap (ky,kx)
            const int oy = yy - (tap / 3 - 1), ox = xx - (tap % 3 - 1);
            if (oy < 0 || oy >= H || ox < 0 || ox >= W) continue;
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                const float g = c[(((long)n * CO + o) * H + oy) * W + ox];
                const float* ww = shw + (tap * CO + o) * C + cc * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += g * ww[e];
            }
        }
        const long row = ((long)n * (H + 2) + yy + 1) * (W + 2) + xx + 1;
        *reinterpret_cast<u32x4_t*>(dx + row * C + cc * 8) =
            u32x4_t{pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3]), pack_bf2(acc[4], acc[5]), pack_bf2(acc[6], acc[7])};
    }
}

// grid: (blocks, 3 (ky), nsets).  Each lane keeps acc[CO][3 kx][8 ch].
template <int CO>
__global__ __launch_bounds__(kThreads) void conv_out_wgrad_kernel(
    const float* __restrict__ c, const bf16_t* __restrict__ x, float* __restrict__ dW, float* __restrict__ dbias,
    int set_images, int nx, long set_stride_w, long set_stride_b, int H, int W, int C) {
    extern __shared__ float sh[];   // [3][CO][C] + CO
    const int ky = blockIdx.y, set = blockIdx.z;
    for (int i = threadIdx.x; i < 3 * CO * C + CO; i += kThreads) sh[i] = 0.f;
    __syncthreads();
    const int lpp = C / 8, ppb = kThreads / lpp;
    const int slot = threadIdx.x / lpp, cc = threadIdx.x - slot * lpp;
    const long npix = (long)set_images * H * W;
    const int Wp = W + 2;
    float acc[CO][3][8];
    float bsum[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        bsum[o] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][k][e] = 0.f;
    }
    if (slot < ppb)
        for (long p = (long)blockIdx.x * ppb + slot; p < npix; p += (long)gridDim.x * ppb) {
            const int xx = p % W; long t = p / W;
            const int yy = t % H; const int nl = t / H;
            const int n2 = set * set_images + nl, n = n2 % nx;
            float g[CO];
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                g[o] = c[(((long)n2 * CO + o) * H + yy) * W + xx];
                if (cc == 0 && ky == 0) bsum[o] += g[o];
            }
            const long row = ((long)n * (H + 2) + yy + 1 + (ky - 1)) * Wp + xx + 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float v[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(x + (row + kx - 1) * C + cc * 8), v);
#pragma unroll
                for (int o = 0; o < CO; ++o)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[o][kx][e] += g[o] * v[e];
            }
        }
    if (slot < ppb) {
#pragma unroll
        for (int o = 0; o < CO; ++o) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int e = 0; e < 8; ++e) atomicAdd(&sh[(kx * CO + o) * C + cc * 8 + e], acc[o][kx][e]);
            if (cc == 0 && ky == 0) atomicAdd(&sh[3 * CO * C + o], bsum[o]);
        }
    }
    __syncthreads();
    float* out = dW + (long)set * set_stride_w + (long)(ky * 3) * CO * C;
    for (int i = threadIdx.x; i < 3 * CO * C; i += kThreads) atomicAdd(out + i, sh[i]);
    if (ky == 0 && threadIdx.x < CO) atomicAdd(dbias + (long)set * set_stride_b + threadIdx.x, sh[3 * CO * C + threadIdx.x]);
}

inline bool lpp_ok(int C) {
    if (C % 8) return false;
    const int l = C / 8;
    return l >= 1 && l <= 64 && (l & (l - 1)) == 0;
}

}  // namespace

#define DISPATCH_CO(CO, CALL) \
    switch (CO) {             \
        case 1: { constexpr int kCO = 1; CALL; } break; \
        case 2: { constexpr int kCO = 2; CALL; } break; \
        case 3: { constexpr int kCO = 3; CALL; } break; \
        case 4: { constexpr int kCO = 4; CALL; } break; \
        default: return SISS_ERR_ARG; \
    }

extern "C" {

int siss_conv_out_fprop(const void* x, const float* w, const float* bias, float* pred, int B, int H, int W, int C,
                        int CO, void* stream) {
    SISS_CHECK_ARG(x && w && bias && pred && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
    SISS_CHECK_ARG(9L * CO * C * sizeof(float) <= 64 * 1024);
    hipStream_t st = (hipStream_t)stream;
    if (C % 32 == 0) {                                     // weights as the MFMA A operand: 860 -> 340 us at CelebA-HQ (LDS-bound before)
        long nb = ((long)B * ((H + 7) / 8) * ((W + 15) / 16) + 3) / 4;        // 8-row x 16-pixel items, 4 per block
        if (nb > 256 * 8) nb = 256 * 8;
        DISPATCH_CO(CO, (conv_out_fprop_mfma_kernel<kCO><<<(int)nb, kThreads, 9 * kCO * C * sizeof(bf16_t), st>>>((const bf16_t*)x, w, bias, pred, B, H, W, C)));
        SISS_LAUNCH_RET();
    }
    if (!lpp_ok(C)) {
        long nb = ((long)B * H * W + 3) / 4;
        if (nb > 4096) nb = 4096;
        DISPATCH_CO(CO, (conv_out_fprop_wave_kernel<kCO><<<(int)nb, kThreads, 9 * kCO * C * sizeof(float), st>>>((const bf16_t*)x, w, bias, pred, B, H, W, C)));
        SISS_LAUNCH_RET();
    }
    const int ppb = kThreads / (C / 8);
    long nb = ((long)B * H * W + ppb - 1) / ppb;
    if (nb > 4096) nb = 4096;
    DISPATCH_CO(CO, (conv_out_fprop_kernel<kCO><<<(int)nb, kThreads, 9 * kCO * C * sizeof(float), st>>>((const bf16_t*)x, w, bias, pred, B, H, W, C)));
    SISS_LAUNCH_RET();
}

int siss_conv_out_dgrad(const float* c, const float* w, void* dx, int N2, int H, int W, int C, int CO, void* stream) {
    SISS_CHECK_ARG(c && w && dx && N2 > 0 && H > 0 && W > 0 && C % 8 == 0);
    long nb = ((long)N2 * H * W * (C / 8) + kThreads - 1) / kThreads;
    if (nb > 8192) nb = 8192;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_CO(CO, (conv_out_dgrad_kernel<kCO><<<(int)nb, kThreads, 9 * kCO * C * sizeof(float), st>>>(c, w, (bf16_t*)dx, N2, H, W, C)));
    SISS_LAUNCH_RET();
}

// c: [nsets*set_images][CO][H][W] f32 cotangent; x: saved conv_out input (nx images, index n2 % nx).
int siss_conv_out_wgrad(const float* c, const void* x, float* dW, float* dbias, int nsets, int set_images, int nx,
                        long set_stride_w, long set_stride_b, int H, int W, int C, int CO, void* stream) {
    SISS_CHECK_ARG(c && x && dW && dbias && nsets > 0 && set_images > 0 && nx > 0 && lpp_ok(C));
    const int ppb = kThreads / (C / 8);
    long nb = ((long)set_images * H * W + (long)ppb * 64 - 1) / ((long)ppb * 64);
    if (nb < 1) nb = 1;
    if (nb > 256) nb = 256;
    dim3 grid((int)nb, 3, nsets);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_CO(CO, (conv_out_wgrad_kernel<kCO><<<grid, kThreads, (3 * kCO * C + kCO) * sizeof(float), st>>>(c, (const bf16_t*)x, dW, dbias, set_images, nx, set_stride_w, set_stride_b, H, W, C)));
    SISS_LAUNCH_RET();
}

}  // extern "C"
