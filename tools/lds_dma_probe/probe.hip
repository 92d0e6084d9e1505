#include <hip/hip_runtime.h>
#include <stdint.h>
template <int IMM>
__device__ static inline void dma16(const void *sbase, unsigned voff, double *ldsp)
{
    const unsigned lds = (unsigned) (uintptr_t) (__attribute__((address_space(3))) double *) ldsp;
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds), "n"(IMM) : "memory");
}
__global__ void k(const double *g, double *out, int E)
{
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    for (int q = 0; q < 4; q++)
    {
        const double *src = g + (size_t) (blockIdx.x * 4 + q) * E;
        dma16<0>(src, lane * 16, smem + q * 512);
        dma16<1024>(src, lane * 16, smem + q * 512);
        if (lane < 20) dma16<2048>(src, lane * 16, smem + q * 512);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double s = 0;
    for (int i = 0; i < 8; i++) s += smem[lane + 64 * i] * (i + 1);
    out[blockIdx.x * 64 + lane] = s;
    out[gridDim.x * 64 + blockIdx.x * 64 + lane] = smem[512 + 256 + lane];
}
int main(int argc, char **argv)
{
    const int E = argc > 1 ? atoi(argv[1]) : 300, NB = 3;
    double *g, *o; hipMalloc(&g, sizeof(double) * E * 4 * NB); hipMalloc(&o, sizeof(double) * 128 * NB);
    double *h = new double[E * 4 * NB]; for (int i = 0; i < E * 4 * NB; i++) h[i] = i * 0.5;
    hipMemcpy(g, h, sizeof(double) * E * 4 * NB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 2048 * 8, 0, g, o, E);
    double r[128 * 3]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < NB; b++)
        for (int l = 0; l < 64; l++)
        {
            double s = 0;
            for (int i = 0; i < 8; i++)
            {
                const int e = l + 64 * i, q = e / 512, w = e % 512;   /* smem[e]: instance q, element w (w < 296 valid: 128+128+40) */
                s += (w < 296 ? h[(size_t) (b * 4 + q) * E + w] : 0.0) * (i + 1);
            }
            /* elements beyond 296 in each region are uninitialised LDS: only compare where all terms valid */
            bool ok = true; for (int i = 0; i < 8; i++) if ((l + 64 * i) % 512 >= 296) ok = false;
            if (ok && r[b * 64 + l] != s) bad++;
            const double want = (256 + l < 296) ? h[(size_t) (b * 4 + 1) * E + 256 + l] : -1;
            if (want >= 0 && r[NB * 64 + b * 64 + l] != want) bad++;
        }
    printf("bad %d  sample %g %g\n", bad, r[0], r[NB * 64 + 3]);
    return bad != 0;
}
