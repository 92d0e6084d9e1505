/* LDS-DMA with several single-wave workgroups per SIMD: a double-buffered stage pipeline of the shape the two-rows-per-lane
 * sweeps use (wait for the DMA of stage k, issue the DMA of stage k + 1 into the other buffer, read stage k from LDS).
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off probe2.hip -o probe2 && ./probe2
 * Each variant prints the number of wrong lanes.  WPE = waves per SIMD the kernel is compiled for, PAD = extra LDS per block
 * (limits the blocks per CU at run time). */
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
constexpr int CH = 1024; /* doubles per stage and block: 8 DMA instructions of 1 KB */
__device__ static inline void dma_stage(const double *src, int lane, double *buf)
{
    dma16<0>(src, lane * 16, buf); dma16<1024>(src, lane * 16, buf); dma16<2048>(src, lane * 16, buf); dma16<3072>(src, lane * 16, buf);
    dma16<0>(src + 512, lane * 16, buf + 512); dma16<1024>(src + 512, lane * 16, buf + 512);
    dma16<2048>(src + 512, lane * 16, buf + 512); dma16<3072>(src + 512, lane * 16, buf + 512);
}
template <int WPE, int SPIN>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k(const double *g, double *out, int stages)
{
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const double *base = g + (size_t) blockIdx.x * stages * CH;
    dma_stage(base, lane, smem);
    double s = 0.0;
    for (int kk = 0; kk < stages; kk++)
    {
        double *cur = smem + (kk & 1) * CH, *nxt = smem + ((kk + 1) & 1) * CH;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kk + 1 < stages) dma_stage(base + (size_t) (kk + 1) * CH, lane, nxt);
        double a = 0.0;
        for (int i = 0; i < CH / 64; i++) a += cur[lane + 64 * i] * (double) (i + 1);
        for (int q = 0; q < SPIN; q++) a = a * 1.0000000001 + 1e-30; /* arithmetic between the stages */
        s += a * (double) (kk + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[(size_t) blockIdx.x * 64 + lane] = s;
}
template <int WPE, int SPIN>
static int run(const char *name, int NB, int stages, int pad, const double *g, const std::vector<double> &h)
{
    double *o; hipMalloc(&o, sizeof(double) * 64 * NB);
    hipLaunchKernelGGL((k<WPE, SPIN>), dim3(NB), dim3(64), 2 * CH * 8 + pad, 0, g, o, stages);
    std::vector<double> r((size_t) 64 * NB);
    hipMemcpy(r.data(), o, sizeof(double) * 64 * NB, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < NB; b++)
        for (int l = 0; l < 64; l++)
        {
            double s = 0.0;
            for (int kk = 0; kk < stages; kk++)
            {
                double a = 0.0;
                for (int i = 0; i < CH / 64; i++) a += h[((size_t) b * stages + kk) * CH + l + 64 * i] * (double) (i + 1);
                for (int q = 0; q < SPIN; q++) a = a * 1.0000000001 + 1e-30;
                s += a * (double) (kk + 1);
            }
            if (r[(size_t) b * 64 + l] != s) bad++;
        }
    printf("%-44s blocks %d stages %d pad %d KB: wrong lanes %d of %d\n", name, NB, stages, pad / 1024, bad, NB * 64);
    hipFree(o);
    return bad;
}
int main()
{
    const int NB = 8192, stages = 12;
    std::vector<double> h((size_t) NB * stages * CH);
    for (size_t i = 0; i < h.size(); i++) h[i] = (double) ((i * 2654435761u) % 1000003) * 0.25;
    double *g; hipMalloc(&g, sizeof(double) * h.size());
    hipMemcpy(g, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    int bad = 0;
    bad += run<1, 0>("one wave per SIMD (compiled), no pad", NB, stages, 0, g, h);
    bad += run<1, 0>("one wave per SIMD, 24 KB pad (4 blocks / CU)", NB, stages, 24 * 1024, g, h);
    bad += run<2, 0>("two waves per SIMD", NB, stages, 0, g, h);
    bad += run<2, 400>("two waves per SIMD, arithmetic between", NB, stages, 0, g, h);
    bad += run<4, 0>("four waves per SIMD", NB, stages, 0, g, h);
    bad += run<4, 400>("four waves per SIMD, arithmetic between", NB, stages, 0, g, h);
    return bad != 0;
}
