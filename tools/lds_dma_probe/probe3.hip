/* Do ordinary vector loads and LDS-DMA loads of one wave return in issue order?  The compiler's s_waitcnt vmcnt(N) for its
 * own loads assumes they do (it does not see the inline-asm DMA instructions; with in-order return a wait that leaves N
 * younger operations outstanding still covers every older one).
 * Per stage: ordinary load A (a line nobody touched before: HBM), 8 DMA instructions on lines that are L2 hits, ordinary
 * load B; A is consumed first (the compiler waits with vmcnt(1): B may stay outstanding).  If the DMA completions could
 * overtake A, the counter would reach 1 with A still in flight and a stale register would be summed.
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off probe3.hip -o probe3 && ./probe3 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
template <int IMM>
__device__ static inline void dma16(const void *sbase, unsigned voff, double *ldsp)
{
    const unsigned lds = (unsigned) (uintptr_t) (__attribute__((address_space(3))) double *) ldsp;
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds), "n"(IMM) : "memory");
}
constexpr int CH = 1024;
__device__ static inline void dma_stage(const double *src, int lane, double *buf)
{
    dma16<0>(src, lane * 16, buf); dma16<1024>(src, lane * 16, buf); dma16<2048>(src, lane * 16, buf); dma16<3072>(src, lane * 16, buf);
    dma16<0>(src + 512, lane * 16, buf + 512); dma16<1024>(src + 512, lane * 16, buf + 512);
    dma16<2048>(src + 512, lane * 16, buf + 512); dma16<3072>(src + 512, lane * 16, buf + 512);
}
template <int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
k(const double *hot, const double *cold, double *out, int stages, size_t cstride)
{
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const double *hb = hot + (size_t) (blockIdx.x & 63) * CH; /* 64 hot regions shared by all blocks: L2 hits */
    const double *cb = cold + (size_t) blockIdx.x * 64 + lane;
    double sa = 0.0, sb = 0.0, sl = 0.0;
    for (int kk = 0; kk < stages; kk++)
    {
        const double a = cb[(size_t) kk * cstride];                 /* A: cold line */
        dma_stage(hb, lane, smem + (kk & 1) * CH);                  /* 8 DMA instructions, hot lines */
        const double b = cb[(size_t) kk * cstride + cstride / 2];   /* B: another cold line */
        sa += a * (double) (kk + 1);                                /* consumes A: vmcnt(1) */
        asm volatile("" : "+v"(sa));                                /* ... in front of the full wait below */
        sb += b * (double) (kk + 2);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        sl += smem[(kk & 1) * CH + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[(size_t) blockIdx.x * 64 + lane] = sa;
    out[(size_t) (gridDim.x + blockIdx.x) * 64 + lane] = sb;
    out[(size_t) (2 * gridDim.x + blockIdx.x) * 64 + lane] = sl;
}
int main()
{
    const int NB = 4096, stages = 16;
    const size_t cstride = (size_t) NB * 64 * 2;
    std::vector<double> hot((size_t) 64 * CH), cold(cstride * stages);
    for (size_t i = 0; i < hot.size(); i++) hot[i] = (double) (i % 977) + 0.5;
    for (size_t i = 0; i < cold.size(); i++) cold[i] = (double) ((i * 2654435761u) % 1000003) * 0.25;
    double *dh, *dc, *o;
    hipMalloc(&dh, sizeof(double) * hot.size()); hipMalloc(&dc, sizeof(double) * cold.size()); hipMalloc(&o, sizeof(double) * 3 * 64 * NB);
    hipMemcpy(dh, hot.data(), sizeof(double) * hot.size(), hipMemcpyHostToDevice);
    hipMemcpy(dc, cold.data(), sizeof(double) * cold.size(), hipMemcpyHostToDevice);
    int total = 0;
    for (int rep = 0; rep < 6; rep++)
    {
        if (rep % 3 == 0) hipLaunchKernelGGL((k<1>), dim3(NB), dim3(64), 2 * CH * 8, 0, dh, dc, o, stages, cstride);
        else if (rep % 3 == 1) hipLaunchKernelGGL((k<2>), dim3(NB), dim3(64), 2 * CH * 8, 0, dh, dc, o, stages, cstride);
        else hipLaunchKernelGGL((k<4>), dim3(NB), dim3(64), 2 * CH * 8, 0, dh, dc, o, stages, cstride);
        std::vector<double> r((size_t) 3 * 64 * NB);
        hipMemcpy(r.data(), o, sizeof(double) * r.size(), hipMemcpyDeviceToHost);
        int ba = 0, bb = 0, bl = 0;
        for (int b = 0; b < NB; b++)
            for (int l = 0; l < 64; l++)
            {
                double sa = 0.0, sb = 0.0, sl = 0.0;
                for (int kk = 0; kk < stages; kk++)
                {
                    sa += cold[(size_t) kk * cstride + (size_t) b * 64 + l] * (double) (kk + 1);
                    sb += cold[(size_t) kk * cstride + cstride / 2 + (size_t) b * 64 + l] * (double) (kk + 2);
                    sl += hot[(size_t) (b & 63) * CH + l];
                }
                ba += r[(size_t) b * 64 + l] != sa; bb += r[(size_t) (NB + b) * 64 + l] != sb; bl += r[(size_t) (2 * NB + b) * 64 + l] != sl;
            }
        printf("compiled for %d wave(s) per SIMD: wrong sums A %d  B %d  LDS %d  (of %d lanes)\n", rep % 3 == 0 ? 1 : rep % 3 == 1 ? 2 : 4, ba, bb, bl, NB * 64);
        total += ba + bb + bl;
    }
    return total != 0;
}
