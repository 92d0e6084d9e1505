/*
 * mfma4.hpp -- v_mfma_f64_4x4x4_4b_f64 as the products of a wavefront that carries FOUR INSTANCES, and the small cross-lane
 * helpers of that mapping (pcond_kernels_mfma.hpp, ipm_kernels_w16t.hpp).
 *
 * Layout of the instruction (tools/mfma_f64_probe/probe3.hip, profiles/r04_mfma4x4x4_layout.txt): block b = (lane >> 2) & 3;
 * with x = lane & 3, y = lane >> 4:  A[i][k] in lane (x = i, y = k),  B[k][j] in lane (x = j, y = k),  D[i][j] in lane
 * (x = j, y = i).  A block -- one instance -- is a QUAD of lanes in each of the four 16-lane rows.  Tiles are kept in the D
 * layout, "lane (x, y) holds element [y][x]"; such a tile P passed as the A operand is read as P', as the B operand as itself:
 *     gqp_mfma4(P, Q, C) = C + P' Q.
 * Rate (profiles/r04_mfma4x4x4_probe.txt): 17-19 cycles per instruction and SIMD = 67-73 TFLOP/s, above v_fma_f64 (69) and
 * far above v_mfma_f64_16x16x4 (47.6); dependent accumulator chain 48 cycles.
 * Cross-lane traffic inside a block: x is the position in a quad (DPP quad_perm), y the 16-lane row (no DPP reaches across
 * rows: an MFMA with a 0/1 selector tile does, see the row broadcast of the diagonal blocks in ipm_kernels_w16t.hpp).
 */
#ifndef GQP_MFMA4_HPP_
#define GQP_MFMA4_HPP_

namespace gqp
{

#if defined(__HIP_DEVICE_COMPILE__)
__device__ static inline double gqp_mfma4(double p, double q, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(p, q, c, 0, 0, 0); }
/* bit (x + 4 y) of the result = predicate of lane (x, y) of this lane's MFMA block */
__device__ static inline unsigned mfma4_blockbits(bool pr)
{
    const unsigned long long bal = __ballot(pr) >> (threadIdx.x & 12); /* (bits 2-3 of the lane index: the block) */
    return (unsigned) ((bal & 0xF) | ((bal >> 12) & 0xF0) | ((bal >> 24) & 0xF00) | ((bal >> 36) & 0xF000));
}
/* value of lane x = J of this lane's quad (same y) */
template <int J>
__device__ static inline double mfma4_qbc(double v) { return __builtin_amdgcn_update_dpp(v, v, J * 0x55, 0xF, 0xF, true); }
/* sum over the four lanes of the quad, result in all of them */
__device__ static inline double mfma4_qsum(double v)
{
    v += __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, true); /* quad_perm:[1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, true); /* quad_perm:[2,3,0,1] */
    return v;
}
#else
/* host pass of hipcc (never executed) and the host simulation of the CPU test tier: the lanes of a workgroup are coroutines
 * of one host thread, operands are exchanged through storage they share */
__device__ static inline double gqp_mfma4(double p, double q, double c)
{
    __shared__ double m4_a[256], m4_b[256];
    const int w0 = threadIdx.x & ~63, l = threadIdx.x & 63, blk = (l >> 2) & 3, j = l & 3, i = l >> 4;
    m4_a[w0 + l] = p; m4_b[w0 + l] = q;
    __syncthreads();
    double s = c;
    for (int k = 0; k < 4; k++) s += m4_a[w0 + i + 4 * blk + 16 * k] * m4_b[w0 + j + 4 * blk + 16 * k];
    __syncthreads();
    return s;
}
__device__ static inline unsigned mfma4_blockbits(bool pr)
{
    __shared__ int m4_f[256];
    const int w0 = threadIdx.x & ~63, l = threadIdx.x & 63, blk = (l >> 2) & 3;
    m4_f[w0 + l] = pr ? 1 : 0;
    __syncthreads();
    unsigned m = 0;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) m |= m4_f[w0 + x + 4 * blk + 16 * y] ? 1u << (x + 4 * y) : 0u;
    __syncthreads();
    return m;
}
template <int J>
__device__ static inline double mfma4_qbc(double v)
{
    __shared__ double m4_q[256];
    m4_q[threadIdx.x] = v;
    __syncthreads();
    const double r = m4_q[(threadIdx.x & ~3) + J];
    __syncthreads();
    return r;
}
__device__ static inline double mfma4_qsum(double v)
{
    __shared__ double m4_s[256];
    m4_s[threadIdx.x] = v;
    __syncthreads();
    const int q0 = threadIdx.x & ~3;
    /* the order of the device: (v + partner) + (partner pair) */
    const double a = m4_s[threadIdx.x] + m4_s[threadIdx.x ^ 1], b = m4_s[threadIdx.x ^ 2] + m4_s[threadIdx.x ^ 3];
    (void) q0;
    __syncthreads();
    return a + b;
}
#endif

} // namespace gqp

#endif
