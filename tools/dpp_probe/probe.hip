// probe: does v_mov_b32_dpp row_newbcast:J deliver lane J of every 16-lane row on this GPU?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int J>
__device__ inline double row_bcast(double v)
{
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xF, 0xF, true); /* v_mov_b64_dpp row_newbcast:J bound_ctrl:1 */
}
__global__ void k(double *p)
{
    const double v = 1000.0 + threadIdx.x;
    p[threadIdx.x] = row_bcast<5>(v);
    p[64 + threadIdx.x] = row_bcast<15>(v);
    if ((threadIdx.x & 15) < 8) p[128 + threadIdx.x] = row_bcast<12>(v); /* partially active rows: source lane inactive */
}
int main()
{
    double *d, h[192];
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) { if (h[i] != 1000.0 + (i & ~15) + 5) bad++; if (h[64 + i] != 1000.0 + (i & ~15) + 15) bad++; }
    printf("row_newbcast: %d mismatches; partially active row, lane 0 got %g (source lane 12 inactive)\n", bad, h[128]);
    return bad != 0;
}
