// Development probe (round 6): can a kernel read n separately malloc'ed, hipHostRegister'ed host blocks (an acados capsule's QP memory)
// at PCIe rate?  Prints: registration time for n blocks, kernel read rate from registered blocks (coalesced 8-byte lanes, one
// workgroup per block chunk), the same from one hipHostMalloc block, and a hipMemcpy of the same bytes for reference.
//     hipcc --offload-arch=gfx950 -O2 probe.hip -o probe && ./probe [n_blocks] [kb_per_block]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_read(const double *const *ptrs, int words, double *out)
{
    const double *p = ptrs[blockIdx.x];
    double s = 0.0;
    for (int w = threadIdx.x; w < words; w += blockDim.x) s += p[w];
    out[(size_t) blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 4096, kb = argc > 2 ? atoi(argv[2]) : 150;
    const int words = kb * 1024 / 8;
    std::vector<double *> blocks(n);
    for (int i = 0; i < n; i++) { blocks[i] = (double *) calloc(words, 8); for (int w = 0; w < words; w += 512) blocks[i][w] = 1.0; }
    double t0 = now();
    int fails = 0;
    for (int i = 0; i < n; i++) if (hipHostRegister(blocks[i], (size_t) words * 8, hipHostRegisterDefault) != hipSuccess) fails++;
    double t_reg = now() - t0;
    printf("register %d blocks of %d KB: %.1f ms (%.1f us each), failures %d\n", n, kb, t_reg * 1e3, t_reg * 1e6 / n, fails);
    if (fails) { (void) hipGetLastError(); }
    std::vector<const double *> dptr(n);
    for (int i = 0; i < n; i++) { void *d = nullptr; if (hipHostGetDevicePointer(&d, blocks[i], 0) != hipSuccess) d = blocks[i]; dptr[i] = (const double *) d; }
    printf("device pointer == host pointer: %s\n", dptr[0] == blocks[0] ? "yes" : "no");
    const double **d_ptrs; double *d_out;
    CK(hipMalloc(&d_ptrs, sizeof(double *) * n)); CK(hipMalloc(&d_out, sizeof(double) * (size_t) n * 256));
    CK(hipMemcpy(d_ptrs, dptr.data(), sizeof(double *) * n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++)
    {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_read, dim3(n), dim3(256), 0, 0, (const double *const *) d_ptrs, words, d_out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("kernel read of registered blocks: %.2f ms = %.1f GB/s\n", ms, (double) n * words * 8 / ms / 1e6);
    }
    double *pinned; CK(hipHostMalloc(&pinned, (size_t) n * words * 8));
    for (int i = 0; i < n; i++) dptr[i] = pinned + (size_t) i * words;
    CK(hipMemcpy(d_ptrs, dptr.data(), sizeof(double *) * n, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++)
    {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_read, dim3(n), dim3(256), 0, 0, (const double *const *) d_ptrs, words, d_out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("kernel read of one hipHostMalloc block: %.2f ms = %.1f GB/s\n", ms, (double) n * words * 8 / ms / 1e6);
    }
    double *dev; CK(hipMalloc(&dev, (size_t) n * words * 8));
    CK(hipEventRecord(e0, 0)); CK(hipMemcpyAsync(dev, pinned, (size_t) n * words * 8, hipMemcpyHostToDevice, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipMemcpy H2D of the pinned block: %.2f ms = %.1f GB/s\n", ms, (double) n * words * 8 / ms / 1e6);
    t0 = now();
    for (int i = 0; i < n; i++) (void) hipHostUnregister(blocks[i]);
    printf("unregister: %.1f ms\n", (now() - t0) * 1e3);
    return 0;
}
