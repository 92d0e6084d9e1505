/* Is a PARTIAL s_waitcnt vmcnt(N) enough in front of the consumer of a UNIFORM-address table load that is followed by
 * per-lane loads?  This is the pattern the two-rows forward sweep (ipm_kernels_w16r.hpp) had while the stage table was read
 * through a plain pointer: hipcc fetched the descriptor with vector loads in SADDR form (global_load_dword v, v_zero, s[..]),
 * issued the stage's per-lane loads behind them and put `s_waitcnt vmcnt(4)` / `vmcnt(3)` in front of the v_readfirstlane /
 * v_mov that consume the descriptor.  Built for two waves per SIMD that kernel was wrong on 30-160 of 65,536 instances per
 * solve; with `s_waitcnt vmcnt(0)` behind the descriptor loads, or with the table in the constant address space (s_load),
 * it is exact (profiles/NOTES.md, round 3).  Here the pattern alone: every wave walks a table shared by all waves (hot in
 * L2 / TCP), loads two cold per-lane lines per step behind each entry and consumes the entry first.
 *   hipcc --offload-arch=gfx950 -O3 probe4.hip -o probe4 && ./probe4 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
struct Desc { int nb, ng, ns, o_ct, o_s, o_g, has_dyn, pad; uint64_t bmask, emask; int8_t srev[64]; }; /* 112 bytes, as GqpStage */
struct Dev { const Desc *st; const double *cold; double *out; uint64_t *outm; int *flag; }; /* pointers inside a by-value struct, as GqpDev */
template <int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
k(Dev D, int stages, size_t cstride)
{
    const Desc *tab = D.st; const double *cold = D.cold; double *out = D.out; uint64_t *outm = D.outm;
    const int lane = threadIdx.x;
    const double *cb = cold + (size_t) blockIdx.x * 64 + lane;
    /* stores of the table's field types first: from here on the table may not be read with scalar loads */
    out[(size_t) blockIdx.x * 64 + lane] = 0.0;
    outm[(size_t) blockIdx.x * 64 + lane] = 0;
    D.flag[(size_t) blockIdx.x * 64 + lane] = 1;
    double sa = 0.0;
    uint64_t sm = 0;
    for (int kk = 0; kk < stages; kk++)
    {
        /* the entry (uniform address: SADDR-form vector loads), then two per-lane cold lines whose addresses do not depend
         * on it, then the entry's consumers: the compiler waits with vmcnt(4) ... vmcnt(2) in front of them */
        const int nbv = tab[kk].nb, oct = tab[kk].o_ct;
        const uint64_t bm = tab[kk].bmask, em = tab[kk].emask;
        const double a = cb[(size_t) kk * cstride];
        const double b = cb[(size_t) kk * cstride + cstride / 2];
        const int nb = __builtin_amdgcn_readfirstlane(nbv);
        sm += (bm >> (lane & 31)) + (em ^ (uint64_t) nb);
        asm volatile("" : "+v"(sm));
        sa += a * (double) (nb + 1) + b * (double) (oct + 2);
    }
    out[(size_t) blockIdx.x * 64 + lane] = sa;
    outm[(size_t) blockIdx.x * 64 + lane] = sm;
}
int main()
{
    const int NB = 8192, stages = 51;
    const size_t cstride = (size_t) NB * 64 * 2;
    std::vector<Desc> tab(stages);
    for (int k = 0; k < stages; k++)
    {
        tab[k].nb = 3 + (k * 7) % 11; tab[k].o_ct = k * 6 + 1; tab[k].bmask = 0x7ull + ((uint64_t) k << 20); tab[k].emask = (uint64_t) (k % 5) << 3;
    }
    std::vector<double> cold(cstride * stages);
    for (size_t i = 0; i < cold.size(); i++) cold[i] = (double) ((i * 2654435761u) % 1000003) * 0.25;
    Desc *dt; double *dc, *o; uint64_t *om; int *fl;
    hipMalloc(&dt, sizeof(Desc) * stages); hipMalloc(&dc, sizeof(double) * cold.size()); hipMalloc(&o, sizeof(double) * 64 * NB); hipMalloc(&om, 8 * 64 * NB); hipMalloc(&fl, 4 * 64 * NB);
    hipMemcpy(dt, tab.data(), sizeof(Desc) * stages, hipMemcpyHostToDevice);
    hipMemcpy(dc, cold.data(), sizeof(double) * cold.size(), hipMemcpyHostToDevice);
    std::vector<double> want((size_t) 64 * NB); std::vector<uint64_t> wantm((size_t) 64 * NB);
    for (int b = 0; b < NB; b++)
        for (int l = 0; l < 64; l++)
        {
            double sa = 0.0; uint64_t sm = 0;
            for (int kk = 0; kk < stages; kk++)
            {
                const int nb = tab[kk].nb;
                const double a = cold[(size_t) kk * cstride + (size_t) b * 64 + l];
                const double bb = cold[(size_t) kk * cstride + cstride / 2 + (size_t) b * 64 + l];
                sa += a * (double) (nb + 1) + bb * (double) (tab[kk].o_ct + 2);
                sm += (tab[kk].bmask >> (l & 31)) + (tab[kk].emask ^ (uint64_t) nb);
            }
            want[(size_t) b * 64 + l] = sa; wantm[(size_t) b * 64 + l] = sm;
        }
    const Dev dv = {dt, dc, o, om, fl};
    int total = 0;
    for (int rep = 0; rep < 9; rep++)
    {
        const int w = rep % 3 == 0 ? 1 : rep % 3 == 1 ? 2 : 4;
        if (w == 1) hipLaunchKernelGGL((k<1>), dim3(NB), dim3(64), 0, 0, dv, stages, cstride);
        else if (w == 2) hipLaunchKernelGGL((k<2>), dim3(NB), dim3(64), 0, 0, dv, stages, cstride);
        else hipLaunchKernelGGL((k<4>), dim3(NB), dim3(64), 0, 0, dv, stages, cstride);
        std::vector<double> r((size_t) 64 * NB); std::vector<uint64_t> rm((size_t) 64 * NB);
        hipMemcpy(r.data(), o, sizeof(double) * r.size(), hipMemcpyDeviceToHost);
        hipMemcpy(rm.data(), om, 8 * rm.size(), hipMemcpyDeviceToHost);
        int ba = 0, bm = 0;
        for (size_t i = 0; i < r.size(); i++) { ba += r[i] != want[i]; bm += rm[i] != wantm[i]; }
        printf("compiled for %d wave(s) per SIMD: wrong sums (values) %d  (masks) %d  of %d lanes\n", w, ba, bm, NB * 64);
        total += ba + bm;
    }
    return total != 0;
}
