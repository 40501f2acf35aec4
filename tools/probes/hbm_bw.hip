// HBM streaming ceilings of this chip from hand-written kernels (torch's reduction kernel, which profiles/r02_mall_bandwidth_probe.txt
// used for the "read" column, is not a bandwidth kernel): read-only, copy, and the GroupNorm-backward mixes (3 reads; 4 reads + 1 write),
// 16 B per lane, U independent loads in flight per lane, persistent grid of `bpc` blocks per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_bw.hip -o tools/probes/hbm_bw && tools/probes/hbm_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, int NR, int NW, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ a, u32x4* __restrict__ out, long nvec, u32x4* sink) {
    // NR read streams (a + k * nvec), NW write streams (out + k * nvec); nvec 16-B vectors per stream
    const long stride = (long)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride * U) {
        u32x4 v[U][NR];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const long j = i + u * stride;
                const u32x4* p = a + r * nvec + (j < nvec ? j : i);
                v[u][r] = NT ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u32x4 s = v[u][0];
#pragma unroll
            for (int r = 1; r < NR; ++r) s ^= v[u][r];
            acc ^= s;
            const long j = i + u * stride;
            if (NW > 0 && j < nvec) {
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    if (NT) __builtin_nontemporal_store(s, out + w * nvec + j); else out[w * nvec + j] = s;
                }
            }
        }
    }
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) *sink = acc;      // never true: keeps the loads alive
}

template <int U, int NR, int NW, bool NT>
double run(const u32x4* a, u32x4* out, long nvec, u32x4* sink, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) stream_kernel<U, NR, NW, NT><<<blocks, 256>>>(a, out, nvec, sink);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) stream_kernel<U, NR, NW, NT><<<blocks, 256>>>(a, out, nvec, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)(NR + NW) * nvec * 16 / (ms / reps * 1e-3) / 1e12;
}

int main() {
    const long bytes = 512L << 20;                  // per stream: beyond the 256 MB Infinity Cache
    const long nvec = bytes / 16;
    u32x4 *a, *out, *sink;
    hipMalloc(&a, 4 * bytes); hipMalloc(&out, 2 * bytes); hipMalloc(&sink, 64);
    hipMemset(a, 1, 4 * bytes); hipMemset(out, 0, 2 * bytes);
    printf("streams of %ld MB, TB/s of (reads + writes)\n", bytes >> 20);
    for (int bpc : {2, 4, 8, 16}) {
        const int blocks = 256 * bpc;
        printf("blocks/CU %2d | R1  U1 %.2f U2 %.2f U4 %.2f U8 %.2f | R1 nt U4 %.2f | R1W1 U2 %.2f U4 %.2f nt %.2f | R3 U2 %.2f U4 %.2f | R4W1 U1 %.2f U2 %.2f nt %.2f | R4W2 U2 %.2f\n", bpc,
               run<1, 1, 0, false>(a, out, nvec, sink, blocks), run<2, 1, 0, false>(a, out, nvec, sink, blocks),
               run<4, 1, 0, false>(a, out, nvec, sink, blocks), run<8, 1, 0, false>(a, out, nvec, sink, blocks),
               run<4, 1, 0, true>(a, out, nvec, sink, blocks),
               run<2, 1, 1, false>(a, out, nvec, sink, blocks), run<4, 1, 1, false>(a, out, nvec, sink, blocks), run<4, 1, 1, true>(a, out, nvec, sink, blocks),
               run<2, 3, 0, false>(a, out, nvec, sink, blocks), run<4, 3, 0, false>(a, out, nvec, sink, blocks),
               run<1, 4, 1, false>(a, out, nvec, sink, blocks), run<2, 4, 1, false>(a, out, nvec, sink, blocks), run<2, 4, 1, true>(a, out, nvec, sink, blocks),
               run<2, 4, 2, false>(a, out, nvec, sink, blocks));
        fflush(stdout);
    }
    return 0;
}
