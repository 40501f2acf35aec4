// How many CUs does it take to stream at the HBM ceiling?  GroupNorm-like mixes (2 reads; 2 reads + 1 write) from ONE 1024-thread
// block per CU on `ncu` CUs (the dispatcher deals a grid of <= 256 blocks to distinct CUs), U independent 16-B loads per lane and
// stream in flight: in-flight bytes per CU = 1024 x U x NR x 16.  If 128 CUs hold the rate that 256 hold, an HBM-bound kernel can
// leave half the chip to an MFMA-bound one beside it.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_half.hip -o tools/probes/hbm_half && tools/probes/hbm_half
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, int NR, int NW>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ a, u32x4* __restrict__ out, long nvec, u32x4* sink) {
    const long stride = (long)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride * U) {
        u32x4 v[U][NR];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const long j = i + u * stride;
                v[u][r] = a[r * nvec + (j < nvec ? j : i)];
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u32x4 s = v[u][0];
#pragma unroll
            for (int r = 1; r < NR; ++r) s ^= v[u][r];
            acc ^= s;
            const long j = i + u * stride;
            if (NW > 0 && j < nvec) out[j] = s;
        }
    }
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) *sink = acc;
}

template <int U, int NR, int NW>
double run(const u32x4* a, u32x4* out, long nvec, u32x4* sink, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    stream_kernel<U, NR, NW><<<blocks, 1024>>>(a, out, nvec, sink);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) stream_kernel<U, NR, NW><<<blocks, 1024>>>(a, out, nvec, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)(NR + NW) * nvec * 16 / (ms / reps * 1e-3) / 1e12;
}

int main() {
    const long bytes = 512L << 20;
    const long nvec = bytes / 16;
    u32x4 *a, *out, *sink;
    hipMalloc(&a, 2 * bytes); hipMalloc(&out, bytes); hipMalloc(&sink, 64);
    hipMemset(a, 1, 2 * bytes); hipMemset(out, 0, bytes);
    printf("streams of %ld MB; TB/s of (reads + writes); one 1024-thread block per CU\n", bytes >> 20);
    for (int ncu : {64, 96, 128, 160, 192, 256, 512}) {
        printf("blocks %3d | R2   U1 %.2f U2 %.2f U4 %.2f | R2W1 U1 %.2f U2 %.2f U4 %.2f\n", ncu,
               run<1, 2, 0>(a, out, nvec, sink, ncu), run<2, 2, 0>(a, out, nvec, sink, ncu), run<4, 2, 0>(a, out, nvec, sink, ncu),
               run<1, 2, 1>(a, out, nvec, sink, ncu), run<2, 2, 1>(a, out, nvec, sink, ncu), run<4, 2, 1>(a, out, nvec, sink, ncu));
        fflush(stdout);
    }
    return 0;
}
