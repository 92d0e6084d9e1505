// v_fmac_f64_dpp with row_newbcast (DP-ALU DPP, gfx90a+): value check, and cycles of the two ways to feed two multiply-adds
// with one row broadcast -- a v_mov_b64_dpp followed by two plain v_fmac_f64 (what the compiler emits for
// __builtin_amdgcn_update_dpp + fma) against two v_fmac_f64_dpp reading the broadcast lane directly.
// Build: hipcc --offload-arch=gfx950 -O3 probe.hip -o probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int J>
__device__ static inline void fmac_bc(double &acc, double src, double other)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(other), "n"(J));
}
template <int J>
__device__ static inline double bc(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xF, 0xF, true); }

__global__ void check(const double *a, const double *b, double *o)
{
    const int t = threadIdx.x;
    double acc = 1.0, x = a[t], y = b[t];
    fmac_bc<3>(acc, x, y);
    fmac_bc<10>(acc, x, y);
    o[t] = acc;
}
template <bool FUSED>
__global__ void bench(const double *a, double *o, long long *cyc, int iters)
{
    const int t = threadIdx.x;
    double w0 = a[t], w1 = a[t + 64], m[16];
    for (int c = 0; c < 16; c++) m[c] = a[t] * c;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int c = 0; c < 8; c++)
        {
            if (FUSED)
            {
                switch (c) {
#define C(J) case J: fmac_bc<J>(m[2 * J], w0, w0); fmac_bc<J>(m[2 * J + 1], w0, w1); break;
                    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7)
#undef C
                }
            }
            else
            {
                double v;
                switch (c) {
#define C(J) case J: v = bc<J>(w0); break;
                    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7)
#undef C
                    default: v = 0;
                }
                m[2 * c] += v * w0; m[2 * c + 1] += v * w1;
            }
        }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < 16; c++) s += m[c];
    o[t + blockIdx.x * 64] = s;
    if (t == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    double ha[128], hb[64], ho[64], *a, *b, *o; long long *cyc, hc;
    for (int i = 0; i < 128; i++) ha[i] = (i % 64) * 1e-3 + 0.5;
    for (int i = 0; i < 64; i++) hb[i] = 2.0 + i * 0.25;
    hipMalloc(&a, 1024); hipMalloc(&b, 512); hipMalloc(&o, 512 * 2048); hipMalloc(&cyc, 8);
    hipMemcpy(a, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, a, b, o);
    hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++)
    {
        const int r = i & ~15;
        const double want = 1.0 + ha[r + 3] * hb[i] + ha[r + 10] * hb[i];
        if (ho[i] != want) { bad++; if (bad <= 6) printf("  lane %d: got %.17g want %.17g diff %.3g\n", i, ho[i], want, ho[i] - want); }
    }
    printf("v_fmac_f64_dpp row_newbcast value check: %d wrong lanes of 64\n", bad);
    const int iters = 2000;
    for (int blocks : {1, 1024})
        for (int fused = 0; fused < 2; fused++)
        {
            if (fused) hipLaunchKernelGGL(bench<true>, dim3(blocks), dim3(64), 0, 0, a, o, cyc, iters);
            else hipLaunchKernelGGL(bench<false>, dim3(blocks), dim3(64), 0, 0, a, o, cyc, iters);
            hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
            printf("%4d wave(s), %s: %.2f cycles per (broadcast + 2 multiply-adds)\n", blocks, fused ? "2 x v_fmac_f64_dpp          " : "v_mov_b64_dpp + 2 x v_fmac_f64",
                   (double) hc / iters / 8);
        }
    return bad != 0;
}
