/* Does an LDS-DMA request (global_load_lds_dwordx4) count and return in vmcnt like any other vector load?  The two-rows
 * kernels (ipm_kernels_w16r.hpp) issue LDS-DMA from inline asm -- invisible to hipcc's wait-count insertion -- next to
 * compiler-scheduled VGPR loads, and the compiler puts PARTIAL waits `s_waitcnt vmcnt(n > 0)` in front of the consumers of
 * its own loads (tools/isa_lint.py lists them: 9-65 per kernel).  With in-order returns a request the compiler does not know
 * about can only make such a wait MORE conservative: the load X it waits for has n compiler-known loads behind it AND d
 * DMA requests, vmcnt(n) forces everything but the last n issued to complete, X is not among the last n.  That argument
 * breaks only if a DMA request is counted differently (not at all, twice, or released early).  This probe measures it:
 *
 *     D D  X  D D  Y Y   |  s_waitcnt vmcnt(W)  |  copy of X taken  |  s_waitcnt vmcnt(0)  |  checks
 *
 * X, Y: cold per-lane global_load_dwordx2 into registers preset to a sentinel; D: cold 1 KB LDS-DMA requests.
 *   W = 2   what a compiler that sees only X, Y, Y would emit (two known loads behind X)          must be exact
 *   W = 4   the true number of requests behind X                                                   must be exact if in order
 *   W = 5   one too many: the CONTROL -- X may still be in flight, stale copies must show up, else the probe is blind
 * for kernels built for 1, 2 and 4 waves per SIMD, alone and with a foreign streaming kernel on a second stream.
 *   hipcc --offload-arch=gfx950 -O3 probe5.hip -o probe5 && ./probe5 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

template <int W, int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
probe(const double *dma_src, const double *cold, int steps, size_t lane_stride, unsigned *bad_x, unsigned *bad_y, unsigned *bad_lds)
{
    __shared__ double lds[4 * 128]; /* four requests of 64 lanes x 16 bytes */
    const int lane = threadIdx.x;
    const unsigned voff = lane * 16u;
    const unsigned ldsb = (unsigned) (uintptr_t) (__attribute__((address_space(3))) const double *) lds;
    unsigned bx = 0, by = 0, bl = 0;
    for (int s = 0; s < steps; s++)
    {
        const size_t it = (size_t) blockIdx.x * steps + s;
        const double *sb = dma_src + it * 512;                    /* wave-uniform: 4 KB of cold data per step */
        const double *ax = cold + (it * 3 + 0) * lane_stride + lane, *ay0 = cold + (it * 3 + 1) * lane_stride + lane,
                     *ay1 = cold + (it * 3 + 2) * lane_stride + lane;
        double x, y0, y1, cx;
        asm volatile("v_mov_b64 %[x], -1.0\n\tv_mov_b64 %[y0], -1.0\n\tv_mov_b64 %[y1], -1.0\n\t"
                     "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[voff], %[sb] offset:0\n\t"
                     "global_load_lds_dwordx4 %[voff], %[sb] offset:1024\n\t"
                     "global_load_dwordx2 %[x], %[ax], off\n\t"
                     "global_load_lds_dwordx4 %[voff], %[sb] offset:2048\n\t"
                     "global_load_lds_dwordx4 %[voff], %[sb] offset:3072\n\t"
                     "global_load_dwordx2 %[y0], %[ay0], off\n\t"
                     "global_load_dwordx2 %[y1], %[ay1], off\n\t"
                     "s_waitcnt vmcnt(%[w])\n\t"
                     "v_mov_b64 %[cx], %[x]\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "s_barrier"
                     : [x] "=&v"(x), [y0] "=&v"(y0), [y1] "=&v"(y1), [cx] "=&v"(cx)
                     : [voff] "v"(voff), [sb] "s"(sb), [lds] "s"(ldsb), [ax] "v"(ax), [ay0] "v"(ay0), [ay1] "v"(ay1), [w] "n"(W)
                     : "memory", "m0");
        /* expected values: cold[i] = i & 0xfffff as a double (>= 0: the sentinel -1.0 never appears) */
        const double wx = (double) (((it * 3 + 0) * lane_stride + lane) & 0xfffff), wy0 = (double) (((it * 3 + 1) * lane_stride + lane) & 0xfffff),
                     wy1 = (double) (((it * 3 + 2) * lane_stride + lane) & 0xfffff);
        bx += cx != wx;
        by += (x != wx) + (y0 != wy0) + (y1 != wy1);
        for (int r = 0; r < 4; r++)
        {
            const double l0 = lds[r * 128 + lane * 2], l1 = lds[r * 128 + lane * 2 + 1];
            bl += (l0 != (double) ((it * 512 + r * 128 + lane * 2) & 0xfffff)) + (l1 != (double) ((it * 512 + r * 128 + lane * 2 + 1) & 0xfffff));
        }
        __syncthreads();
    }
    if (bx) atomicAdd(bad_x, bx);
    if (by) atomicAdd(bad_y, by);
    if (bl) atomicAdd(bad_lds, bl);
}

__global__ void stream_copy(const double *a, double *b, size_t n, int reps)
{
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) b[i] = a[i] + (double) r;
}

template <int W, int WPE>
static int run(const double *dsrc, const double *cold, int NB, int steps, size_t ls, unsigned *cnt, hipStream_t st, const char *what)
{
    CHK(hipMemsetAsync(cnt, 0, 12, st));
    hipLaunchKernelGGL((probe<W, WPE>), dim3(NB), dim3(64), 0, st, dsrc, cold, steps, ls, cnt, cnt + 1, cnt + 2);
    unsigned h[3];
    CHK(hipMemcpyAsync(h, cnt, 12, hipMemcpyDeviceToHost, st));
    CHK(hipStreamSynchronize(st));
    printf("%-28s wait vmcnt(%d), built for %d wave(s) per SIMD: stale copies of X %8u   wrong after vmcnt(0) %u   wrong LDS-DMA data %u   (of %d lane-steps)\n",
           what, W, WPE, h[0], h[1], h[2], NB * 64 * steps);
    return (W <= 4 && (h[0] || h[1] || h[2])) ? 1 : 0;
}

int main()
{
    const int NB = 4096, steps = 24;
    const size_t ls = 64;                                        /* doubles between the lines of X, Y0, Y1 of one step (never touched before: cold) */
    const size_t n_dma = (size_t) NB * steps * 512, n_cold = ((size_t) NB * steps * 3 + 1) * ls;
    std::vector<double> h(n_dma > n_cold ? n_dma : n_cold);
    for (size_t i = 0; i < h.size(); i++) h[i] = (double) (i & 0xfffff);
    double *dsrc, *cold, *fa, *fb;
    unsigned *cnt;
    CHK(hipMalloc(&dsrc, sizeof(double) * n_dma)); CHK(hipMalloc(&cold, sizeof(double) * n_cold)); CHK(hipMalloc(&cnt, 12));
    CHK(hipMemcpy(dsrc, h.data(), sizeof(double) * n_dma, hipMemcpyHostToDevice));
    CHK(hipMemcpy(cold, h.data(), sizeof(double) * n_cold, hipMemcpyHostToDevice));
    const size_t nf = (size_t) 1 << 27;
    CHK(hipMalloc(&fa, sizeof(double) * nf)); CHK(hipMalloc(&fb, sizeof(double) * nf));
    CHK(hipMemset(fa, 0, sizeof(double) * nf));
    hipStream_t s1, s2;
    CHK(hipStreamCreate(&s1)); CHK(hipStreamCreate(&s2));
    int fail = 0, control_seen = 0;
    for (int foreign = 0; foreign < 2; foreign++)
    {
        const char *what = foreign ? "with a foreign stream kernel" : "alone";
        for (int rep = 0; rep < 2; rep++)
        {
            if (foreign) hipLaunchKernelGGL(stream_copy, dim3(2048), dim3(256), 0, s2, fa, fb, nf, 6);
            fail += run<2, 1>(dsrc, cold, NB, steps, ls, cnt, s1, what); fail += run<2, 2>(dsrc, cold, NB, steps, ls, cnt, s1, what);
            fail += run<2, 4>(dsrc, cold, NB, steps, ls, cnt, s1, what);
            fail += run<4, 1>(dsrc, cold, NB, steps, ls, cnt, s1, what); fail += run<4, 2>(dsrc, cold, NB, steps, ls, cnt, s1, what);
            fail += run<4, 4>(dsrc, cold, NB, steps, ls, cnt, s1, what);
            run<5, 1>(dsrc, cold, NB, steps, ls, cnt, s1, what); run<5, 2>(dsrc, cold, NB, steps, ls, cnt, s1, what);
            run<5, 4>(dsrc, cold, NB, steps, ls, cnt, s1, what);
            if (foreign) CHK(hipStreamSynchronize(s2));
        }
    }
    printf(fail ? "RESULT: a wait that should have been sufficient was not (see above)\n"
                : "RESULT: vmcnt(2) [compiler's view] and vmcnt(4) [true count] exact in every configuration; the vmcnt(5) lines are the control\n");
    return fail != 0;
}
