// FP64 MFMA microbenchmark ("confirm with a v_mfma_f64_16x16x4_f64 microbenchmark before quoting a
// fraction"): (1) operand / result layout check with A = I and an ASYMMETRIC B, (2) issue rate with 4 independent
// accumulators per wave at 1, 2, 4 waves per SIMD, (3) latency of a dependent accumulator chain, (4) the same flops as
// v_fma_f64 for comparison.  Build + run on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe/probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double *A, const double *B, double *D)
{
    // D (16x16) = A (16x4) * B (4x16); lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]
    const int l = threadIdx.x;
    d4 c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r]; // row = (lane >> 4) + 4 reg, col = lane & 15
}

template <int NACC>
__global__ void __launch_bounds__(256) k_rate(double *out, int iters, double a0, double b0)
{
    d4 c[NACC];
    for (int q = 0; q < NACC; q++) c[q] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int q = 0; q < NACC; q++) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[q], 0, 0, 0);
    }
    double s = 0.0;
    for (int q = 0; q < NACC; q++) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
    if (s == 1.2345e-300) out[0] = s;
}

__global__ void __launch_bounds__(256) k_fma(double *out, int iters, double a0, double b0)
{
    double c[16];
    for (int q = 0; q < 16; q++) c[q] = q;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int q = 0; q < 16; q++) c[q] = __builtin_fma(a, b, c[q]);
    }
    double s = 0.0;
    for (int q = 0; q < 16; q++) s += c[q];
    if (s == 1.2345e-300) out[0] = s;
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class F>
static double time_ms(F f)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    f();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    f();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("device %s, %d CUs, %.2f GHz\n", p.name, cus, ghz);
    // (1) layout
    std::vector<double> A(64, 0.0), B(64), D(256);
    for (int i = 0; i < 4; i++) A[i * 4 + i] = 1.0;                 // A = [I4; 0]
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 100.0 * k + j + 0.5 * (k == 2); // asymmetric
    double *dA, *dB, *dD;
    CHK(hipMalloc(&dA, 64 * 8)); CHK(hipMalloc(&dB, 64 * 8)); CHK(hipMalloc(&dD, 256 * 8));
    CHK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CHK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) bad += D[i * 16 + j] != (i < 4 ? B[i * 16 + j] : 0.0);
    printf("layout check (A = [I;0], asymmetric B; A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4r col=l&15): %s\n", bad ? "MISMATCH" : "ok");
    // (2) issue rate
    const int iters = 20000;
    double *out;
    CHK(hipMalloc(&out, 8));
    for (int wps = 1; wps <= 4; wps *= 2)
    {
        const int blocks = cus * wps; // 256 threads = 4 waves per block = one wave per SIMD and block
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
        const double nmfma = (double) blocks * 4 * iters * 4;
        const double tf = nmfma * 2048.0 / (ms * 1e-3) / 1e12;
        const double cyc = ms * 1e-3 * ghz * 1e9 / ((double) iters * 4 * wps);
        printf("v_mfma_f64_16x16x4_f64, 4 independent accumulators, %d wave(s)/SIMD: %.1f TFLOP/s, %.1f cycles per MFMA per SIMD\n", wps, tf, cyc);
    }
    {
        const int blocks = cus;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
        printf("dependent chain (1 accumulator, 1 wave/SIMD): %.1f cycles per MFMA\n", ms * 1e-3 * ghz * 1e9 / iters);
    }
    for (int wps = 1; wps <= 4; wps *= 2)
    {
        const int blocks = cus * wps;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
        const double tf = (double) blocks * 256 * iters * 16 * 2 / (ms * 1e-3) / 1e12;
        printf("v_fma_f64, 16 independent accumulators, %d wave(s)/SIMD: %.1f TFLOP/s, %.2f cycles per wave instruction\n", wps, tf,
               ms * 1e-3 * ghz * 1e9 / ((double) iters * 16 * wps));
    }
    return bad;
}
