// Shared device helpers for the gfx950 (CDNA4) kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // storage type for bf16 in global memory / C-ABI signatures

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;     // packed-f32 math (v_pk_add / v_pk_fma)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;   // operand of v_dot2c_f32_bf16

#define SISS_OK 0
#define SISS_ERR_ARG 1
#define SISS_ERR_LAUNCH 2

#define SISS_CHECK_ARG(cond) \
    do {                     \
        if (!(cond)) return SISS_ERR_ARG; \
    } while (0)

#define SISS_LAUNCH_RET()                                 \
    do {                                                  \
        return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH; \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) {
    return __builtin_bit_cast(float, (uint32_t)v << 16);
}
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }
// Element access generic over the ACTIVATION type: bf16_t (the product path) or float (the f32 parity mode, f32_path.hip and the
// `_f32` entry points: `mixed_precision: null` of the reference's YAMLs -- every tensor f32, every product on v_mfma_f32_16x16x4_f32).
__device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
__device__ __forceinline__ float to_f(float v) { return v; }
template <typename T> __device__ __forceinline__ T from_f(float f);
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float f) { return f2bf(f); }
template <> __device__ __forceinline__ float from_f<float>(float f) { return f; }
// two floats -> one packed pair: ONE v_cvt_pk_bf16_f32 (a two-element vector conversion).  Written as two scalar casts + shift / or
// it only became that instruction where the SLP vectoriser paired the casts; flash_attn.hip is built without it and paid
// 2 conversions + a shift + an SDWA or per pair (64 instead of 16 vector instructions per tile in the dK / dV loop).
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef float f32x2_pk __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_pk __attribute__((ext_vector_type(2)));
    const bf16x2_pk b = __builtin_convertvector(f32x2_pk{lo, hi}, bf16x2_pk);
    return __builtin_bit_cast(uint32_t, b);
}

// (a.lo + b.lo, a.hi + b.hi) of two packed bf16 pairs: f32 adds, ONE rounding (RNE) by v_cvt_pk_bf16_f32 -- 7 instructions.  The
// conversion is inline asm: written with casts the SLP vectoriser pairs the additions of DIFFERENT words and re-interleaves the
// halves with two more SDWA ors per word.
__device__ __forceinline__ uint32_t add_bf16x2(uint32_t a, uint32_t b) {
    const float lo = __builtin_bit_cast(float, a << 16) + __builtin_bit_cast(float, b << 16);
    const float hi = __builtin_bit_cast(float, a & 0xffff0000u) + __builtin_bit_cast(float, b & 0xffff0000u);
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over the eight lanes that share (lane & 7), result in all of them, on the VALU only (no LDS round trips as
// __shfl_xor's ds_bpermute): lane ^ 8 by a DPP row rotate, lane ^ 16 / lane ^ 32 by gfx950's permlane swaps
// (v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of the second,
// v_permlane32_swap the upper half of the first with the lower half of the second: swapping a value with itself and
// adding the two results adds each lane's partner).
__device__ __forceinline__ float sum_lanes_mod8(float t) {
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    // (operands and results go through empty asm statements: hipcc 7.2 otherwise folds `r[0] + r[1]` of a swap of a value
    // with itself into `r[0] + r[0]` -- checked in the ISA and by tests/test_hip_gn_qstats.py)
    unsigned u = __builtin_bit_cast(unsigned, t), u2 = u;
    asm volatile("" : "+v"(u2));
    const auto a = __builtin_amdgcn_permlane16_swap(u, u2, false, false);
    unsigned a0 = a[0], a1 = a[1];
    asm volatile("" : "+v"(a0), "+v"(a1));
    t = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
    unsigned v = __builtin_bit_cast(unsigned, t), v2 = v;
    asm volatile("" : "+v"(v2));
    const auto b = __builtin_amdgcn_permlane32_swap(v, v2, false, false);
    unsigned b0 = b[0], b1 = b[1];
    asm volatile("" : "+v"(b0), "+v"(b1));
    return __builtin_bit_cast(float, b0) + __builtin_bit_cast(float, b1);
}

// sigmoid with the hardware reciprocal (v_rcp_f32, 1 ulp): an IEEE divide costs ~10 VALU ops and these
// sit in HBM-bound kernels that are otherwise close to VALU-bound
__device__ __forceinline__ float sigmoid_f(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
__device__ __forceinline__ float silu_f(float z) { return z * sigmoid_f(z); }
// d/dz [z * sigmoid(z)]
__device__ __forceinline__ float dsilu_f(float z) {
    const float s = sigmoid_f(z);
    return s * (1.f + z * (1.f - s));
}

// Exact-GELU pieces from ONE exponential: Phi(g) = 0.5 (1 + erf(g / sqrt 2)) with erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the bf16 the results are rounded to): erf(x) = 1 - (a1 t + ... + a5 t^5) exp(-x^2), t = 1 / (1 + p x),
// x = |g| / sqrt 2 -- and exp(-x^2) = exp(-g^2 / 2) is the density term of gelu'(g) = Phi(g) + g phi(g) as well.  libm's erff cost
// ~30 instructions per call (two calls + an expf per element in the backward): these kernels were VALU-bound at 2.5-3.3 TB/s.
__device__ __forceinline__ void gelu_parts(float g, float& Phi, float& e) {
    const float x = fabsf(g) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.f));
    e = __builtin_amdgcn_exp2f(g * g * -0.72134752044448170f);               // exp(-g^2 / 2)
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erf_abs = fmaf(-poly, e, 1.f);
    Phi = 0.5f * (1.f + copysignf(erf_abs, g));
}
__device__ __forceinline__ float gelu_f(float g) { float P, e; gelu_parts(g, P, e); return g * P; }
__device__ __forceinline__ float dgelu_f(float g) { float P, e; gelu_parts(g, P, e); return fmaf(g * 0.3989422804014327f, e, P); }


// LDS-DMA (global_load_lds_dwordx4) as inline asm: 16 B per lane, lane-linear LDS destination starting at the
// wave-uniform byte address `lds_dst`.  Unlike __builtin_amdgcn_global_load_lds the compiler does not know this
// statement writes LDS, so it does NOT put a conservative s_waitcnt vmcnt(0) in front of the next LDS read --
// completion is counted by hand (s_waitcnt vmcnt(N) + barrier before the staged bytes are read).
__device__ __forceinline__ void glds16_asm(const void* g, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
// The same with the address as wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset (the `saddr` form): no vector
// arithmetic per piece when only the base moves between pieces, and m0 is simply set -- it is a reserved register that
// compiler-generated code sets itself before each of its own uses, so there is nothing to preserve.
__device__ __forceinline__ void glds16_saddr(unsigned off, const void* base, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" :: "v"(off), "s"(lds_dst), "s"(base) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- host-side launcher state, PER DEVICE (a process may drive several GPUs; nothing here is shared between them) ----
constexpr int kMaxDevices = 64;
// Geometry of the GroupNorm statistics a 3x3 convolution can leave for its consumer (NTParams::qstats, siss_groupnorm_fwd_qs):
// entry (2 * t + h, slot) covers rows [t * kQsTileRows + h * kQsHalfRows, ...) of the flat padded row space, h = 0 / 1, of the
// image floor(t * kQsTileRows / rows_per_image) + slot.
constexpr int kQsTileRows = 254, kQsHalfRows = 128;
static inline int siss_current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices ? dev : -1;
}
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel, device).
// `done` is the caller's own per-kernel flag array (one byte per device; racing threads at worst set it twice).
static inline int siss_ensure_smem(const void* kernel, int bytes, unsigned char (&done)[kMaxDevices]) {
    const int dev = siss_current_device();
    if (dev < 0) return SISS_ERR_LAUNCH;
    if (!__atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return SISS_ERR_LAUNCH;
        __atomic_store_n(&done[dev], (unsigned char)1, __ATOMIC_RELEASE);
    }
    return SISS_OK;
}

// ---- dispatch counters (diagnostics): which DEVICE KERNEL a launcher call landed on.  Tests read them through
//      siss_dispatch_count() to prove that a parity case really exercised e.g. gemm_nt_c3p_kernel. ----
enum SissKernelId { SISS_K_NT = 0, SISS_K_NT_C3P, SISS_K_FLASH_FWD, SISS_K_FLASH_BWD, SISS_K_NT_SPLITK, SISS_K_TN1, SISS_K_TN3,
                    SISS_K_GN_SLAB, SISS_K_GN_QSTATS, SISS_K_FLASH_QSPLIT, SISS_K_ATTN1H_FWD, SISS_K_ATTN1H_BWD, SISS_K_TN_PAIR, SISS_K_FLASH32, SISS_K_FLASH32_FWD, SISS_K_NT_WIDE, SISS_K_COUNT };
void siss_count_dispatch(int kernel_id);   // gemm_nt.hip
void* siss_workspace(long* bytes);         // gemm_nt.hip: the current device's workspace (siss_gemm_nt_set_workspace), or null
