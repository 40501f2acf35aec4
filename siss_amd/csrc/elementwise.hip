// Data-movement kernels on the padded-NHWC layout (SURVEY.md §2b K8 and glue): nearest 2x
// upsample (+ its 2x2-sum backward), channel concat / split, space-to-depth for the stride-2
// downsampler (+ inverse), padded<->compact token copies for the attention block, batched
// transpose, per-set column sums (bias gradients), NCHW image -> im2col rows for conv_in.
// All are HBM-bound: one lane moves 16 B (8 bf16 channels) per access, rows are contiguous.
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
__device__ __forceinline__ u32x4_t pack8(const float (&v)[8]) {
    return u32x4_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
}
__device__ __forceinline__ u32x4_t add8(u32x4_t a, u32x4_t b) {
    float x[8], y[8];
    unpack8(a, x); unpack8(b, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] += y[e];
    return pack8(x);
}
__device__ __forceinline__ long prow(int n, int y, int x, int H, int W) {   // padded row of interior (y,x)
    return ((long)n * (H + 2) + (y + 1)) * (W + 2) + (x + 1);
}

// Generic "gather one 16-B chunk per output chunk" driver: total = N*H*W*(C/8) output chunks.
#define FOR_CHUNKS(total) \
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < (total); i += (long)gridDim.x * kThreads)

// out[n, y, x, :] = in[n, y/2, x/2, :]           (out is 2H x 2W)
__global__ void upsample2x_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int N, int H,
                                  int W, int C) {
    const int cc = C / 8, H2 = 2 * H, W2 = 2 * W;
    const long total = (long)N * H2 * W2 * cc;
    FOR_CHUNKS(total) {
        const int c = i % cc; long r = i / cc;
        const int x = r % W2; r /= W2;
        const int y = r % H2; const int n = r / H2;
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(in + prow(n, y >> 1, x >> 1, H, W) * C + c * 8);
        *reinterpret_cast<u32x4_t*>(out + prow(n, y, x, H2, W2) * C + c * 8) = v;
    }
}
// din[n, y, x, :] = sum of the 2x2 block of dout
__global__ void upsample2x_bwd_kernel(const bf16_t* __restrict__ dout, bf16_t* __restrict__ din, int N,
                                      int H, int W, int C) {
    const int cc = C / 8, H2 = 2 * H, W2 = 2 * W;
    const long total = (long)N * H * W * cc;
    FOR_CHUNKS(total) {
        const int c = i % cc; long r = i / cc;
        const int x = r % W; r /= W;
        const int y = r % H; const int n = r / H;
        float a[8] = {}, t[8];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                unpack8(*reinterpret_cast<const u32x4_t*>(dout + prow(n, 2 * y + dy, 2 * x + dx, H2, W2) * C + c * 8), t);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += t[e];
            }
        *reinterpret_cast<u32x4_t*>(din + prow(n, y, x, H, W) * C + c * 8) = pack8(a);
    }
}

// out[..., :Ca] = a ; out[..., Ca:] = b      (hidden FIRST, then the skip: UNet up blocks)
__global__ void concat_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                              bf16_t* __restrict__ out, int N, int H, int W, int Ca, int Cb) {
    const int C = Ca + Cb, cc = C / 8, ca = Ca / 8;
    const long total = (long)N * H * W * cc;
    FOR_CHUNKS(total) {
        const int c = i % cc; long r = i / cc;
        const int x = r % W; r /= W;
        const int y = r % H; const int n = r / H;
        const long row = prow(n, y, x, H, W);
        const u32x4_t v = c < ca ? *reinterpret_cast<const u32x4_t*>(a + row * Ca + c * 8)
                                 : *reinterpret_cast<const u32x4_t*>(b + row * Cb + (c - ca) * 8);
        *reinterpret_cast<u32x4_t*>(out + row * C + c * 8) = v;
    }
}
// out[..., Ca:] = b only: the head columns were written in place by their producer (conv epilogue with ldc = Ca + Cb)
__global__ void concat_tail_kernel(const bf16_t* __restrict__ b, bf16_t* __restrict__ out, int N, int H, int W,
                                   int Ca, int Cb) {
    const int C = Ca + Cb, cb = Cb / 8;
    const long total = (long)N * H * W * cb;
    FOR_CHUNKS(total) {
        const int c = i % cb; long r = i / cb;
        const int x = r % W; r /= W;
        const int y = r % H; const int n = r / H;
        const long row = prow(n, y, x, H, W);
        *reinterpret_cast<u32x4_t*>(out + row * C + Ca + c * 8) = *reinterpret_cast<const u32x4_t*>(b + row * Cb + c * 8);
    }
}
// da = dcat[..., :Ca] (overwrite) ; db (+)= dcat[..., Ca:]
__global__ void concat_bwd_kernel(const bf16_t* __restrict__ dcat, bf16_t* __restrict__ da,
                                  bf16_t* __restrict__ db, int accumulate_b, int N, int H, int W, int Ca,
                                  int Cb) {
    const int C = Ca + Cb, cc = C / 8, ca = Ca / 8;
    const long total = (long)N * H * W * cc;
    FOR_CHUNKS(total) {
        const int c = i % cc; long r = i / cc;
        const int x = r % W; r /= W;
        const int y = r % H; const int n = r / H;
        const long row = prow(n, y, x, H, W);
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(dcat + row * C + c * 8);
        if (c < ca) {
            *reinterpret_cast<u32x4_t*>(da + row * Ca + c * 8) = v;
        } else {
            u32x4_t* d = reinterpret_cast<u32x4_t*>(db + row * Cb + (c - ca) * 8);
            *d = accumulate_b ? add8(*d, v) : v;
        }
    }
}

// a += b over the interior (halo stays zero)
__global__ void add_inplace_kernel(bf16_t* __restrict__ a, const bf16_t* __restrict__ b, int N, int H, int W,
                                   int C) {
    const int cc = C / 8;
    const long total = (long)N * H * W * cc;
    FOR_CHUNKS(total) {
        const int c = i % cc; long r = i / cc;
        const int x = r % W; r /= W;
        const int y = r % H; const int n = r / H;
        const long o = prow(n, y, x, H, W) * C + c * 8;
        *reinterpret_cast<u32x4_t*>(a + o) = add8(*reinterpret_cast<const u32x4_t*>(a + o),
                                                   *reinterpret_cast<const u32x4_t*>(b + o));
    }
}

// space-to-depth: Z[n, i, j, (py*2+px)*C + c] = in[n, 2i+py, 2j+px, c]     (Z is H/2 x W/2 x 4C)
__global__ void s2d_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ z, int N, int H, int W, int C, int ld_in) {
    const int cc = C / 8, Ho = H / 2, Wo = W / 2;
    const long total = (long)N * H * W * cc;
    FOR_CHUNKS(total) {
        const int c = i % cc; long r = i / cc;
        const int x = r % W; r /= W;
        const int y = r % H; const int n = r / H;
        const int plane = (y & 1) * 2 + (x & 1);
        *reinterpret_cast<u32x4_t*>(z + prow(n, y >> 1, x >> 1, Ho, Wo) * (4 * C) + plane * C + c * 8) =
            *reinterpret_cast<const u32x4_t*>(in + prow(n, y, x, H, W) * ld_in + c * 8);
    }
}
// inverse (depth-to-space) with optional accumulation into din
__global__ void d2s_kernel(const bf16_t* __restrict__ dz, bf16_t* __restrict__ din, int accumulate, int N,
                           int H, int W, int C) {
    const int cc = C / 8, Ho = H / 2, Wo = W / 2;
    const long total = (long)N * H * W * cc;
    FOR_CHUNKS(total) {
        const int c = i % cc; long r = i / cc;
        const int x = r % W; r /= W;
        const int y = r % H; const int n = r / H;
        const int plane = (y & 1) * 2 + (x & 1);
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(dz + prow(n, y >> 1, x >> 1, Ho, Wo) * (4 * C) + plane * C + c * 8);
        u32x4_t* d = reinterpret_cast<u32x4_t*>(din + prow(n, y, x, H, W) * C + c * 8);
        *d = accumulate ? add8(*d, v) : v;
    }
}

// padded -> compact ([N][H*W][C]) copy, and compact + padded residual -> padded
__global__ void pad_to_compact_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int N, int H,
                                      int W, int C) {
    const int cc = C / 8;
    const long total = (long)N * H * W * cc;
    FOR_CHUNKS(total) {
        const int c = i % cc; long r = i / cc;
        const int x = r % W; long r2 = r / W;
        const int y = r2 % H; const int n = r2 / H;
        *reinterpret_cast<u32x4_t*>(out + r * C + c * 8) =
            *reinterpret_cast<const u32x4_t*>(in + prow(n, y, x, H, W) * C + c * 8);
    }
}
__global__ void compact_add_to_pad_kernel(const bf16_t* __restrict__ comp, const bf16_t* __restrict__ res,
                                          bf16_t* __restrict__ out, int N, int H, int W, int C) {
    const int cc = C / 8;
    const long total = (long)N * H * W * cc;
    FOR_CHUNKS(total) {
        const int c = i % cc; long r = i / cc;
        const int x = r % W; long r2 = r / W;
        const int y = r2 % H; const int n = r2 / H;
        const long o = prow(n, y, x, H, W) * C + c * 8;
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(comp + r * C + c * 8);
        if (res) v = add8(v, *reinterpret_cast<const u32x4_t*>(res + o));
        *reinterpret_cast<u32x4_t*>(out + o) = v;
    }
}

// batched transpose: in [B][R][C] -> out [B][C][R]
__global__ void transpose_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R, int C) {
    __shared__ bf16_t tile[32][34];
    const long bo = (long)blockIdx.z * R * C;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y)
        if (r0 + r < R && c0 + threadIdx.x < C) tile[r][threadIdx.x] = in[bo + (long)(r0 + r) * C + c0 + threadIdx.x];
    __syncthreads();
    for (int c = threadIdx.y; c < 32; c += blockDim.y)
        if (c0 + c < C && r0 + threadIdx.x < R) out[bo + (long)(c0 + c) * R + r0 + threadIdx.x] = tile[threadIdx.x][c];
}

// out[set][c] += sum over the set's rows of y[r][c]    (rows are flat; halo rows are zero)
__global__ __launch_bounds__(kThreads) void colsum_kernel(const bf16_t* __restrict__ y, long rows_per_set,
                                                          int C, long out_set_stride, float* __restrict__ out,
                                                          float* __restrict__ out2) {
    extern __shared__ float sh[];   // C floats
    const int set = blockIdx.y, cc = C / 8;
    for (int i = threadIdx.x; i < C; i += kThreads) sh[i] = 0.f;
    __syncthreads();
    const int ppi = kThreads / cc, slot = threadIdx.x / cc, c = threadIdx.x - slot * cc;
    if (slot < ppi) {
        float a[8] = {};
        const bf16_t* base = y + (long)set * rows_per_set * C;
        for (long r = (long)blockIdx.x * ppi + slot; r < rows_per_set; r += (long)gridDim.x * ppi) {
            float t[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(base + r * C + c * 8), t);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += t[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(&sh[c * 8 + e], a[e]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += kThreads) {
        atomicAdd(out + (long)set * out_set_stride + i, sh[i]);
        if (out2) atomicAdd(out2 + (long)set * out_set_stride + i, sh[i]);
    }
}

// conv_in front end: NCHW image (f32 or bf16) -> im2col rows [N][H+2][W+2][K] bf16, K >= 9*Cin,
// k = tap*Cin + ci (zero beyond); halo rows are written as zeros.
// flip = 1 mirrors the tap offsets (k = tap*Cin + ci reads img[y - (ky-1), x - (kx-1)]): the im2col of a
// COTANGENT image, which turns conv_out's dgrad / wgrad into plain one-panel GEMMs.
template <bool BF16, int CIN>                                  // CIN: the channel count as a constant (1, 3, 4), or 0 = run-time Cin
__global__ void im2col3x3_kernel(const void* __restrict__ img, bf16_t* __restrict__ out, int N, int Cin_rt, int H,
                                 int W, int K, int flip) {
    const int Cin = CIN ? CIN : Cin_rt;                         // (a constant divisor: k / Cin is a multiply-shift, not a ~25-instruction divide)
    // one thread per 16-B chunk (8 consecutive k of one row): the K / 8 lanes of a row write its K * 2 bytes contiguously, a wave
    // writes 1 KiB in one piece per store instruction (round 4: one thread per ROW wrote eight 16-B chunks 128 B apart from its
    // neighbours' -- 64 separate segments per store instruction, 1.2 TB/s; 350 us of the CelebA-HQ step for 408 MB)
    // The grid carries (image, padded row): a thread decodes only (pixel, chunk) of its row, in 32 bits -- the flat 64-bit index this
    // kernel used to split cost four 64-bit divisions (~100 instructions each) per 16-B store: 300 us of ALU for 408 MB.
    const unsigned cpr = K >> 3;                                // chunks per row
    const int n = blockIdx.z, yp = blockIdx.y;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < (unsigned)(W + 2) * cpr; idx += gridDim.x * blockDim.x) {
        const int xp = idx / cpr, ch = idx - xp * cpr;
        const long r = ((long)n * (H + 2) + yp) * (W + 2) + xp;
        const bool halo = xp == 0 || yp == 0 || xp == W + 1 || yp == H + 1;
        if (halo || ch * 8 >= 9 * Cin) {                        // halo rows and the zero padding of K beyond 9 Cin: nothing to gather
            *reinterpret_cast<u32x4_t*>(out + r * K + ch * 8) = u32x4_t{0u, 0u, 0u, 0u};
            continue;
        }
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ch * 8 + e, tap = k / Cin, ci = k - tap * Cin;
            float val = 0.f;
            if (!halo && tap < 9) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const int y = yp - 1 + (flip ? -dy : dy), x = xp - 1 + (flip ? -dx : dx);
                if (y >= 0 && y < H && x >= 0 && x < W) {
                    const long o = (((long)n * Cin + ci) * H + y) * W + x;
                    val = BF16 ? bf2f(reinterpret_cast<const bf16_t*>(img)[o]) : reinterpret_cast<const float*>(img)[o];
                }
            }
            v[e] = val;
        }
        *reinterpret_cast<u32x4_t*>(out + r * K + ch * 8) = pack8(v);
    }
}

// Fast form for Cin in {1, 3, 4} with K = 64 (conv_in / conv_out of image-space UNets: 9 Cin <= 36 live columns): one thread gathers
// ALL live columns of its row -- every (tap, channel) is a compile-time constant after unrolling, so an element costs one add of a
// wave-uniform offset, a predicate and a load (the one-chunk-per-thread form above re-derived tap / channel / 64-bit address per
// element with quarter-rate 32-bit multiplies: 270 us of the CelebA-HQ step for 408 MB) -- parks the row in LDS and the block
// writes its 256 rows as one contiguous 32 KiB run of 16-B stores.
template <bool BF16, int CIN>
__global__ __launch_bounds__(kThreads) void im2col3x3_rows_kernel(const void* __restrict__ img, bf16_t* __restrict__ out, int H, int W,
                                                                  int flip) {
    constexpr int K = 64, NK = 9 * CIN, ROWB = 2 * K + 16;                 // LDS row: 128 B + 16 B pad (bank spread)
    __shared__ __attribute__((aligned(16))) char tile[kThreads * ROWB];
    const int n = blockIdx.y, Wp = W + 2, rpi = (H + 2) * Wp;
    const int r0 = blockIdx.x * kThreads, rr = r0 + threadIdx.x;          // padded row within the image
    const long hw = (long)H * W;
    if (rr < rpi) {
        const int yp = rr / Wp, xp = rr - yp * Wp;
        const bool halo = xp == 0 || yp == 0 || xp == W + 1 || yp == H + 1;
        float v[40];
#pragma unroll
        for (int k = 0; k < 40; ++k) v[k] = 0.f;
        if (!halo) {
            const int y0 = yp - 1, x0 = xp - 1;
            const long base = (long)n * CIN * hw + (long)y0 * W + x0;
            bool oky[3], okx[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int dd = flip ? 1 - d : d - 1;
                oky[d] = y0 + dd >= 0 && y0 + dd < H; okx[d] = x0 + dd >= 0 && x0 + dd < W;
            }
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int tap = k / CIN, ci = k - tap * CIN, ty = tap / 3, tx = tap - ty * 3;
                const int dy = flip ? 1 - ty : ty - 1, dx = flip ? 1 - tx : tx - 1;
                if (oky[ty] && okx[tx]) {
                    const long o = base + ci * hw + dy * W + dx;
                    v[k] = BF16 ? bf2f(reinterpret_cast<const bf16_t*>(img)[o]) : reinterpret_cast<const float*>(img)[o];
                }
            }
        }
        char* dst = tile + threadIdx.x * ROWB;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const float (&w8)[8] = *reinterpret_cast<const float (*)[8]>(&v[c * 8]);
            *reinterpret_cast<u32x4_t*>(dst + c * 16) = pack8(w8);
        }
#pragma unroll
        for (int c = 5; c < 8; ++c) *reinterpret_cast<u32x4_t*>(dst + c * 16) = u32x4_t{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    const int nrows = rpi - r0 < kThreads ? rpi - r0 : kThreads;
    bf16_t* gout = out + ((long)n * rpi + r0) * K;
    for (int c = threadIdx.x; c < nrows * 8; c += kThreads)
        *reinterpret_cast<u32x4_t*>(gout + (long)c * 8) = *reinterpret_cast<const u32x4_t*>(tile + (c >> 3) * ROWB + (c & 7) * 16);
}

// out[set][c] += sum over the set's images and pixels of img[n][c][:, :]   (f32 NCHW; conv_out bias gradient)
__global__ __launch_bounds__(kThreads) void nchw_channel_sums_kernel(const float* __restrict__ img, int set_images,
                                                                    int C, long hw, long out_set_stride,
                                                                    float* __restrict__ out) {
    __shared__ float sh[kThreads / 64];
    const int c = blockIdx.y, set = blockIdx.z;
    float a = 0.f;
    for (int n = 0; n < set_images; ++n) {
        const float* src = img + ((long)(set * set_images + n) * C + c) * hw;
        for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < hw; i += (long)gridDim.x * kThreads) a += src[i];
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < kThreads / 64; ++i) t += sh[i];
        atomicAdd(out + (long)set * out_set_stride + c, t);
    }
}

inline int grid_for(long total) {
    long b = (total + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;
    return (int)b;
}

}  // namespace

#define EW_ARGS_OK(N, H, W, C) ((N) > 0 && (H) > 0 && (W) > 0 && (C) > 0 && (C) % 8 == 0)

extern "C" {

int siss_upsample2x(const void* in, void* out, int N, int H, int W, int C, void* stream) {
    SISS_CHECK_ARG(in && out && EW_ARGS_OK(N, H, W, C));
    upsample2x_kernel<<<grid_for((long)N * 4 * H * W * (C / 8)), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)in, (bf16_t*)out, N, H, W, C);
    SISS_LAUNCH_RET();
}
int siss_upsample2x_bwd(const void* dout, void* din, int N, int H, int W, int C, void* stream) {
    SISS_CHECK_ARG(dout && din && EW_ARGS_OK(N, H, W, C));
    upsample2x_bwd_kernel<<<grid_for((long)N * H * W * (C / 8)), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)dout, (bf16_t*)din, N, H, W, C);
    SISS_LAUNCH_RET();
}
int siss_concat(const void* a, const void* b, void* out, int N, int H, int W, int Ca, int Cb, void* stream) {
    SISS_CHECK_ARG(a && b && out && EW_ARGS_OK(N, H, W, Ca) && Cb > 0 && Cb % 8 == 0);
    concat_kernel<<<grid_for((long)N * H * W * ((Ca + Cb) / 8)), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, N, H, W, Ca, Cb);
    SISS_LAUNCH_RET();
}
/* out[..., Ca:] = b (padded NHWC, interior pixels); out's first Ca columns are left as their producer wrote them. */
int siss_concat_tail(const void* b, void* out, int N, int H, int W, int Ca, int Cb, void* stream) {
    SISS_CHECK_ARG(b && out && EW_ARGS_OK(N, H, W, Ca) && Cb > 0 && Cb % 8 == 0);
    concat_tail_kernel<<<grid_for((long)N * H * W * (Cb / 8)), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)b, (bf16_t*)out, N, H, W, Ca, Cb);
    SISS_LAUNCH_RET();
}
int siss_concat_bwd(const void* dcat, void* da, void* db, int accumulate_b, int N, int H, int W, int Ca, int Cb,
                    void* stream) {
    SISS_CHECK_ARG(dcat && da && db && EW_ARGS_OK(N, H, W, Ca) && Cb > 0 && Cb % 8 == 0);
    concat_bwd_kernel<<<grid_for((long)N * H * W * ((Ca + Cb) / 8)), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)dcat, (bf16_t*)da, (bf16_t*)db, accumulate_b, N, H, W, Ca, Cb);
    SISS_LAUNCH_RET();
}
int siss_add_inplace(void* a, const void* b, int N, int H, int W, int C, void* stream) {
    SISS_CHECK_ARG(a && b && EW_ARGS_OK(N, H, W, C));
    add_inplace_kernel<<<grid_for((long)N * H * W * (C / 8)), kThreads, 0, (hipStream_t)stream>>>((bf16_t*)a, (const bf16_t*)b, N, H, W, C);
    SISS_LAUNCH_RET();
}
/* ld_in: row stride of `in` in elements (0 = C; > C when `in` is a column view of a wider concat buffer) */
int siss_space_to_depth_ld(const void* in, void* z, int N, int H, int W, int C, int ld_in, void* stream) {
    SISS_CHECK_ARG(in && z && EW_ARGS_OK(N, H, W, C) && H % 2 == 0 && W % 2 == 0 && (ld_in == 0 || (ld_in >= C && ld_in % 8 == 0)));
    s2d_kernel<<<grid_for((long)N * H * W * (C / 8)), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)in, (bf16_t*)z, N, H, W, C, ld_in ? ld_in : C);
    SISS_LAUNCH_RET();
}
int siss_space_to_depth(const void* in, void* z, int N, int H, int W, int C, void* stream) {
    return siss_space_to_depth_ld(in, z, N, H, W, C, 0, stream);
}
int siss_depth_to_space(const void* dz, void* din, int accumulate, int N, int H, int W, int C, void* stream) {
    SISS_CHECK_ARG(dz && din && EW_ARGS_OK(N, H, W, C) && H % 2 == 0 && W % 2 == 0);
    d2s_kernel<<<grid_for((long)N * H * W * (C / 8)), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)dz, (bf16_t*)din, accumulate, N, H, W, C);
    SISS_LAUNCH_RET();
}
int siss_pad_to_compact(const void* in, void* out, int N, int H, int W, int C, void* stream) {
    SISS_CHECK_ARG(in && out && EW_ARGS_OK(N, H, W, C));
    pad_to_compact_kernel<<<grid_for((long)N * H * W * (C / 8)), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)in, (bf16_t*)out, N, H, W, C);
    SISS_LAUNCH_RET();
}
// out_padded = compact (+ res_padded)
int siss_compact_add_to_pad(const void* comp, const void* res, void* out, int N, int H, int W, int C, void* stream) {
    SISS_CHECK_ARG(comp && out && EW_ARGS_OK(N, H, W, C));
    compact_add_to_pad_kernel<<<grid_for((long)N * H * W * (C / 8)), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)comp, (const bf16_t*)res, (bf16_t*)out, N, H, W, C);
    SISS_LAUNCH_RET();
}
int siss_transpose_bf16(const void* in, void* out, int batch, int R, int C, void* stream) {
    SISS_CHECK_ARG(in && out && batch > 0 && R > 0 && C > 0 && batch <= 65535);
    dim3 grid(cdiv(C, 32), cdiv(R, 32), batch), block(32, 8);
    transpose_kernel<<<grid, block, 0, (hipStream_t)stream>>>((const bf16_t*)in, (bf16_t*)out, R, C);
    SISS_LAUNCH_RET();
}
// out[set][0:C] += column sums of y over each set's rows.  y: [nsets*rows_per_set][C] bf16.
// out2 (optional) receives the same sums (two biases that feed the same pre-activation).
int siss_colsum(const void* y, long rows_per_set, int C, int nsets, long out_set_stride, float* out, float* out2,
                void* stream) {
    SISS_CHECK_ARG(y && out && rows_per_set > 0 && C > 0 && C % 8 == 0 && C / 8 <= kThreads && nsets > 0);
    const int ppi = kThreads / (C / 8);
    long nb = (rows_per_set + (long)ppi * 64 - 1) / ((long)ppi * 64);
    if (nb < 1) nb = 1;
    if (nb > 512) nb = 512;
    dim3 grid((int)nb, nsets);
    colsum_kernel<<<grid, kThreads, C * sizeof(float), (hipStream_t)stream>>>((const bf16_t*)y, rows_per_set, C, out_set_stride, out, out2);
    SISS_LAUNCH_RET();
}
int siss_im2col3x3(const void* img, int img_bf16, void* out, int N, int Cin, int H, int W, int K, int flip,
                   void* stream) {
    SISS_CHECK_ARG(img && out && N > 0 && Cin > 0 && H > 0 && W > 0 && K >= 9 * Cin && K % 8 == 0);
    SISS_CHECK_ARG(N <= 65535 && H + 2 <= 65535);
    if (K == 64 && (Cin == 1 || Cin == 3 || Cin == 4)) {
        const dim3 g2(cdiv((long)(H + 2) * (W + 2), kThreads), N);
#define IM2ROWS(BF, CI) im2col3x3_rows_kernel<BF, CI><<<g2, kThreads, 0, (hipStream_t)stream>>>(img, (bf16_t*)out, H, W, flip)
        if (img_bf16) { if (Cin == 3) IM2ROWS(true, 3); else if (Cin == 4) IM2ROWS(true, 4); else IM2ROWS(true, 1); }
        else { if (Cin == 3) IM2ROWS(false, 3); else if (Cin == 4) IM2ROWS(false, 4); else IM2ROWS(false, 1); }
#undef IM2ROWS
        SISS_LAUNCH_RET();
    }
    const dim3 grid(cdiv((long)(W + 2) * (K / 8), kThreads), H + 2, N);
#define IM2COL(BF, CI) im2col3x3_kernel<BF, CI><<<grid, kThreads, 0, (hipStream_t)stream>>>(img, (bf16_t*)out, N, Cin, H, W, K, flip)
    if (img_bf16) { if (Cin == 3) IM2COL(true, 3); else if (Cin == 4) IM2COL(true, 4); else if (Cin == 1) IM2COL(true, 1); else IM2COL(true, 0); }
    else { if (Cin == 3) IM2COL(false, 3); else if (Cin == 4) IM2COL(false, 4); else if (Cin == 1) IM2COL(false, 1); else IM2COL(false, 0); }
#undef IM2COL
    SISS_LAUNCH_RET();
}

// out[set*out_set_stride + c] += sum_{n in set, y, x} img[n][c][y][x]
int siss_nchw_channel_sums(const float* img, int nsets, int set_images, int C, long hw, long out_set_stride, float* out,
                           void* stream) {
    SISS_CHECK_ARG(img && out && nsets > 0 && set_images > 0 && C > 0 && hw > 0);
    long nb = (hw + kThreads * 2 - 1) / (kThreads * 2);     // (round 4: 96 blocks walked the CelebA-HQ cotangent in 97 us; 25 MB)
    if (nb > 128) nb = 128;
    dim3 grid((int)nb, C, nsets);
    nchw_channel_sums_kernel<<<grid, kThreads, 0, (hipStream_t)stream>>>(img, set_images, C, hw, out_set_stride, out);
    SISS_LAUNCH_RET();
}

}  // extern "C"
