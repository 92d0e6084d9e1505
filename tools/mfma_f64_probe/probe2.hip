// Round 4 probe: what could feed the O(n^3) phases of the two-rows family faster than `v_mov_b64_dpp row_newbcast` + two
// v_fma_f64 (8 + 4 + 4 cycles per broadcast group at one wave per SIMD)?
//   (1) v_mfma_f64_4x4x4_4b_f64: FOUR independent 4x4x4 products per instruction, one per 16-lane row -- the very mapping
//       of that family (one instance per row).  Layout found by trying the eight index conventions; issue rate at 1, 2, 4
//       waves per SIMD with 8 independent accumulators; dependent-chain latency.
//   (2) the broadcast operand from LDS instead of DPP: every lane of a row reads the SAME address (ds_read_b64 /
//       ds_read_b128 = two operands), four rows read four addresses; all four SIMDs of every CU busy, so the one LDS pipe
//       of the CU is shared the way it would be in the kernels.  Reported: cycles per broadcast group (one operand, two
//       multiply-adds) for DPP only, LDS b64, LDS b128, and a 1:2 mix.
// Build + run on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe/probe2.hip -o /tmp/mfma_probe2 && /tmp/mfma_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_layout(const double *A, const double *B, double *D)
{
    const int l = threadIdx.x;
    double c = 0.0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], c, 0, 0, 0);
    D[l] = c;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_rate(double *out, int iters, double a0, double b0)
{
    double c[NACC];
    for (int q = 0; q < NACC; q++) c[q] = 0.0;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int q = 0; q < NACC; q++) c[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[q], 0, 0, 0);
    }
    double s = 0.0;
    for (int q = 0; q < NACC; q++) s += c[q];
    if (s == 1.2345e-300) out[0] = s;
}

template <int J>
__device__ static inline double bc(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xF, 0xF, true); }

// MODE 0: 16 DPP broadcasts per pass; 1: 16 operands by ds_read_b64; 2: by 8 ds_read_b128; 3: 5 DPP + 11 LDS (b128 pairs + one b64)
typedef __attribute__((address_space(3))) double LDSD;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) d2 LDSD2;
template <int MODE>
__global__ void __launch_bounds__(256) k_feed(const double *src, double *out, long long *cyc, int iters)
{
    __shared__ double sh[4 * 4 * 32]; // [wave][row][32]
    const int t = threadIdx.x, w = t >> 6, rq = (t >> 4) & 3;
    double *mine = sh + (w * 4 + rq) * 32;
    for (int i = t & 15; i < 32; i += 16) mine[i] = src[i] + 1e-3 * rq;
    __syncthreads();
    double w0 = src[t & 63], w1 = src[(t & 63) + 64], x = src[t & 15];
    double m0[16], m1[16];
    for (int c = 0; c < 16; c++) { m0[c] = 0.0; m1[c] = 0.0; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++)
    {
        // (address space 3 pointer: a generic pointer would compile to flat loads)
        const volatile LDSD *p = (const volatile LDSD *) (sh + (w * 4 + rq) * 32 + (it & 1) * 16);
        double v[16];
        if (MODE == 0)
        {
#define C(J) v[J] = bc<J>(x);
            C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
        }
        else if (MODE == 1)
        {
#pragma unroll
            for (int c = 0; c < 16; c++) v[c] = p[c];
        }
        else if (MODE == 2)
        {
#pragma unroll
            for (int c = 0; c < 8; c++)
            {
                const d2 q = ((const volatile LDSD2 *) p)[c];
                v[2 * c] = q[0]; v[2 * c + 1] = q[1];
            }
        }
        else
        {
#define C(J) v[J] = bc<J>(x);
            C(0) C(1) C(2) C(3) C(4)
#undef C
            v[5] = p[5];
#pragma unroll
            for (int c = 3; c < 8; c++)
            {
                const d2 q = ((const volatile LDSD2 *) p)[c];
                v[2 * c] = q[0]; v[2 * c + 1] = q[1];
            }
        }
#pragma unroll
        for (int c = 0; c < 16; c++) { m0[c] = __builtin_fma(v[c], w0, m0[c]); m1[c] = __builtin_fma(v[c], w1, m1[c]); }
        x += 1e-12;
    }
    const long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < 16; c++) s += m0[c] + m1[c];
    out[t + blockIdx.x * 256] = s;
    if (t == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

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
    // (1) layout of v_mfma_f64_4x4x4_4b_f64
    std::vector<double> A(64), B(64), D(64);
    srand(7);
    for (int i = 0; i < 64; i++) { A[i] = (rand() % 17) - 8; B[i] = (rand() % 13) - 6; }
    double *dA, *dB, *dD;
    CHK(hipMalloc(&dA, 512)); CHK(hipMalloc(&dB, 512)); CHK(hipMalloc(&dD, 512));
    CHK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CHK(hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost));
    // conventions: block = lane >> 4; inside the block lane = lo + 4 * hi; a bit per operand says which of (lo, hi) is the
    // FIRST index: A[i][k], B[k][j], D[i][j]
    int found = 0;
    for (int conv = 0; conv < 8; conv++)
    {
        const int ca = conv & 1, cb = (conv >> 1) & 1, cd = (conv >> 2) & 1;
        int bad = 0;
        for (int blk = 0; blk < 4; blk++)
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++)
                {
                    double s = 0.0;
                    for (int k = 0; k < 4; k++)
                    {
                        const int la = blk * 16 + (ca ? k + 4 * i : i + 4 * k), lb = blk * 16 + (cb ? j + 4 * k : k + 4 * j);
                        s += A[la] * B[lb];
                    }
                    const int ld = blk * 16 + (cd ? j + 4 * i : i + 4 * j);
                    bad += D[ld] != s;
                }
        if (!bad)
        {
            found++;
            printf("v_mfma_f64_4x4x4_4b layout: block = lane >> 4; A[i][k] at lane %s; B[k][j] at lane %s; D[i][j] at lane %s\n",
                   ca ? "k + 4 i" : "i + 4 k", cb ? "j + 4 k" : "k + 4 j", cd ? "j + 4 i" : "i + 4 j");
        }
    }
    if (!found) printf("v_mfma_f64_4x4x4_4b layout: NONE of the eight conventions matches (blocks are not 16-lane rows?)\n");
    // (2) issue rate
    const int iters = 20000;
    double *out;
    CHK(hipMalloc(&out, 8 * 256 * 4096));
    for (int wps = 1; wps <= 4; wps *= 2)
    {
        const int blocks = cus * wps;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_rate<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
        const double nmfma = (double) blocks * 4 * iters * 8;
        const double tf = nmfma * 512.0 / (ms * 1e-3) / 1e12;
        const double cyc = ms * 1e-3 * ghz * 1e9 / ((double) iters * 8 * wps);
        printf("v_mfma_f64_4x4x4_4b_f64, 8 independent accumulators, %d wave(s)/SIMD: %.1f TFLOP/s, %.1f cycles per MFMA per SIMD (512 flops each)\n", wps, tf, cyc);
    }
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_rate<1>, dim3(cus), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
        printf("dependent chain (1 accumulator, 1 wave/SIMD): %.1f cycles per MFMA\n", ms * 1e-3 * ghz * 1e9 / iters);
    }
    // (3) operand feed
    std::vector<double> S(128);
    for (int i = 0; i < 128; i++) S[i] = 0.5 + 1e-3 * i;
    double *dS; long long *dc, hc;
    CHK(hipMalloc(&dS, 1024)); CHK(hipMalloc(&dc, 8));
    CHK(hipMemcpy(dS, S.data(), 1024, hipMemcpyHostToDevice));
    const int fit = 4000;
    const char *names[4] = {"16 DPP row broadcasts", "16 operands by ds_read_b64 (row-uniform address)", "16 operands by 8 ds_read_b128", "5 DPP + 11 from LDS (b128 pairs)"};
    for (int mode = 0; mode < 4; mode++)
        for (int blocks : {1, cus})
        {
            auto go = [&] {
                if (mode == 0) hipLaunchKernelGGL(k_feed<0>, dim3(blocks), dim3(256), 0, 0, dS, out, dc, fit);
                else if (mode == 1) hipLaunchKernelGGL(k_feed<1>, dim3(blocks), dim3(256), 0, 0, dS, out, dc, fit);
                else if (mode == 2) hipLaunchKernelGGL(k_feed<2>, dim3(blocks), dim3(256), 0, 0, dS, out, dc, fit);
                else hipLaunchKernelGGL(k_feed<3>, dim3(blocks), dim3(256), 0, 0, dS, out, dc, fit);
            };
            const double ms = time_ms(go);
            CHK(hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost));
            printf("feed: %-50s %4d block(s) of 4 waves: %.1f cycles per group (1 operand + 2 v_fma_f64), wave 0 clock; %.1f from the launch time\n",
                   names[mode], blocks, (double) hc / ((double) fit * 16), ms * 1e-3 * ghz * 1e9 / ((double) fit * 16));
        }
    return 0;
}
