// Layout of v_mfma_f64_4x4x4_4b_f64 found by experiment: A = delta(lane la), B = delta(lane lb) for all 64 x 64 pairs;
// the lanes of D that come out non-zero give every (la, lb, ld) triple of the instruction (256 of them: 4 blocks x 4 x 4 x 4).
//     hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe/probe3.hip -o tools/mfma_f64_probe/build/probe3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void k_pairs(unsigned long long *mask)
{
    const int l = threadIdx.x, la = blockIdx.x, lb = blockIdx.y;
    const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    const unsigned long long m = __ballot(d != 0.0);
    if (l == 0) mask[la * 64 + lb] = m;
}
int main()
{
    unsigned long long *dm;
    std::vector<unsigned long long> m(4096);
    hipMalloc(&dm, 4096 * 8);
    hipLaunchKernelGGL(k_pairs, dim3(64, 64), dim3(64), 0, 0, dm);
    hipMemcpy(m.data(), dm, 4096 * 8, hipMemcpyDeviceToHost);
    int n = 0;
    for (int la = 0; la < 64; la++)
    {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; lb++)
            if (m[la * 64 + lb])
            {
                printf("  B %2d -> D", lb);
                for (int ld = 0; ld < 64; ld++) if ((m[la * 64 + lb] >> ld) & 1) { printf(" %2d", ld); n++; }
            }
        printf("\n");
    }
    printf("%d (la, lb, ld) triples\n", n);
    // try: lane = x + 4 y + 16 z with (x, y, z) any assignment of (row/col index, block, k or second index)
    const char *nm[3] = {"lane & 3", "(lane >> 2) & 3", "lane >> 4"};
    for (int pa = 0; pa < 6; pa++) for (int pb = 0; pb < 6; pb++) for (int pd = 0; pd < 6; pd++)
    {
        static const int perm[6][3] = {{0,1,2},{0,2,1},{1,0,2},{1,2,0},{2,0,1},{2,1,0}};
        // operand fields: A: (i, k, blk) at bit-pair positions perm[pa]; B: (k, j, blk) at perm[pb]; D: (i, j, blk) at perm[pd]
        int bad = 0;
        for (int blk = 0; blk < 4 && !bad; blk++) for (int i = 0; i < 4 && !bad; i++) for (int j = 0; j < 4 && !bad; j++) for (int k = 0; k < 4; k++)
        {
            const int la = (i << (2 * perm[pa][0])) | (k << (2 * perm[pa][1])) | (blk << (2 * perm[pa][2]));
            const int lb = (k << (2 * perm[pb][0])) | (j << (2 * perm[pb][1])) | (blk << (2 * perm[pb][2]));
            const int ld = (i << (2 * perm[pd][0])) | (j << (2 * perm[pd][1])) | (blk << (2 * perm[pd][2]));
            if (m[la * 64 + lb] != (1ull << ld)) { bad = 1; break; }
        }
        if (!bad)
            printf("MATCH: A[i][k] of block b: i at %s, k at %s, b at %s;  B[k][j]: k at %s, j at %s, b at %s;  D[i][j]: i at %s, j at %s, b at %s\n",
                   nm[perm[pa][0]], nm[perm[pa][1]], nm[perm[pa][2]], nm[perm[pb][0]], nm[perm[pb][1]], nm[perm[pb][2]], nm[perm[pd][0]], nm[perm[pd][1]], nm[perm[pd][2]]);
    }
    return 0;
}
