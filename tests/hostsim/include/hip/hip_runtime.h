/*
 * TEST INFRASTRUCTURE ONLY -- host simulation of the small HIP subset the solver uses.
 *
 * Lets the *unchanged* kernel sources (the .hip / .hpp files of acados_amd/csrc) be compiled with g++
 * and run one "lane" at a time on the CPU, so kernel logic can be checked against the
 * oracle in the CPU-only test tier (`-m "not gpu"`).  It is built by tests/hostsim/build.py
 * into tests/hostsim/libgqp_hostsim.so and is never loaded by the acados_amd package:
 * the product library is the hipcc build of the same sources and needs a real GPU.
 * One-instance-per-lane kernels run lane after lane; kernels with cross-lane communication through shared memory
 * and barriers run as coroutines (GQP_LAUNCH_COOP below).
 */
#ifndef HOSTSIM_HIP_RUNTIME_H_
#define HOSTSIM_HIP_RUNTIME_H_

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
#include <functional>
#include <ucontext.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local /* shared by the simulated lanes of the calling host thread */
#define __launch_bounds__(...)

struct dim3
{
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim; /* one copy across translation units */

typedef int hipError_t;
#define hipSuccess 0
typedef void *hipStream_t;
struct hostsim_event { std::chrono::steady_clock::time_point t; };
typedef hostsim_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };

static inline const char *hipGetErrorString(hipError_t) { return "hostsim"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n) { *p = malloc(n); return hipSuccess; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
enum { hipHostRegisterDefault = 0 };
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; } /* (host memory IS the simulation's device memory) */
static inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
#define hipStreamDefault 0u
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned int, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 1; *greatest = -1; return hipSuccess; }
/* fault injection for the CPU tier (tests/test_boundary.py::test_device_failure_is_a_status_not_an_exit): with
 * GQP_HOSTSIM_FAIL_SYNC=n in the environment the n-th stream synchronisation from now on reports an error (once) */
static inline hipError_t hipStreamSynchronize(hipStream_t)
{
    const char *e = getenv("GQP_HOSTSIM_FAIL_SYNC");
    if (e && *e)
    {
        static int seen = 0;
        static int armed_for = -1;
        const int n = atoi(e);
        if (n != armed_for) { armed_for = n; seen = 0; }
        if (n > 0 && ++seen == n) { seen = -1000000000; return 719; }
    }
    return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hostsim_event(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
static inline int atomicSub(int *p, int v) { int o = *p; *p = o - v; return o; }
static inline int atomicAdd(int *p, int v) { int o = *p; *p = o + v; return o; }
static inline double atomicAdd(double *p, double v) { double o = *p; *p = o + v; return o; }

/* ---- cooperative kernels (lanes of a block that exchange data through shared memory) ----
 * Kernels that use __syncthreads() and block-shared memory are launched through GQP_LAUNCH_COOP: every lane of a
 * block becomes a coroutine (ucontext) on the calling thread; a barrier hands control back to a round-robin
 * scheduler, which resumes a lane only after every other unfinished lane has run up to ITS next barrier (or to its
 * end).  All lanes of a synchronisation group execute the same barrier sequence, so they advance in lockstep; groups
 * that do not share data (the 16-lane rows of the sixteen-lanes kernels) may run different numbers of barriers.
 * `__shared__` statics and the dynamic shared buffer are shared by the lanes.  Blocks run one after the other.
 * (Coroutines instead of one host thread per lane: a futex barrier among 64 threads on a few cores costs ~100 us, a
 * context switch well under 1 us -- the CPU test tier runs the wave-per-instance kernels thousands of times.) */
struct hostsim_coop
{
    enum { MAXL = 256, STACK = 1 << 20 }; /* (256: the four independent waves of a km_pcond workgroup) */
    struct state
    {
        ucontext_t sched, ctx[MAXL];
        char *stacks = nullptr;
        bool done[MAXL];
        int cur = 0;
        std::function<void()> body;
    };
    static state &st() { static thread_local state s; return s; }
    static std::vector<double> &dyn() { static thread_local std::vector<double> v; return v; }
    static void yield() { state &s = st(); swapcontext(&s.ctx[s.cur], &s.sched); }
    static void tramp() { state &s = st(); s.body(); s.done[s.cur] = true; }
    static void run_block(unsigned nl)
    {
        state &s = st();
        if (!s.stacks) s.stacks = (char *) malloc((size_t) MAXL * STACK);
        for (unsigned l = 0; l < nl; l++)
        {
            getcontext(&s.ctx[l]);
            s.ctx[l].uc_stack.ss_sp = s.stacks + (size_t) l * STACK;
            s.ctx[l].uc_stack.ss_size = STACK;
            s.ctx[l].uc_link = &s.sched;
            makecontext(&s.ctx[l], (void (*)()) tramp, 0);
            s.done[l] = false;
        }
        for (unsigned left = nl; left > 0;)
            for (unsigned l = 0; l < nl; l++)
            {
                if (s.done[l]) continue;
                s.cur = (int) l;
                threadIdx.x = l;
                swapcontext(&s.sched, &s.ctx[l]);
                if (s.done[l]) left--;
            }
    }
};
static inline void __syncthreads() { hostsim_coop::yield(); }
/* lanes of one 16-lane row (kernels with several independent instances per block) */
#define GQP_ROWSYNC() hostsim_coop::yield()
#define GQP_DYN_SHARED(name) double *name = hostsim_coop::dyn().data()

#define GQP_LAUNCH_COOP(kern, grid, block, shmem, stream, ...)                              \
    do {                                                                                    \
        dim3 g_ = (grid); dim3 b_ = (block);                                                \
        hostsim_coop::dyn().assign(((size_t) (shmem) + 7) / 8 + 8, 0.0);                    \
        hostsim_coop::st().body = [&]() { kern(__VA_ARGS__); };                             \
        for (unsigned by_ = 0; by_ < g_.y; by_++)                                           \
            for (unsigned bx_ = 0; bx_ < g_.x; bx_++)                                       \
            {                                                                               \
                blockDim = b_; gridDim = g_; blockIdx.x = bx_; blockIdx.y = by_;            \
                hostsim_coop::run_block(b_.x);                                              \
            }                                                                               \
    } while (0)

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...)                           \
    do {                                                                                    \
        dim3 g_ = (grid); dim3 b_ = (block);                                                         \
        blockDim = b_; gridDim = g_;                                                        \
        for (unsigned by_ = 0; by_ < g_.y; by_++)                                           \
            for (unsigned bx_ = 0; bx_ < g_.x; bx_++)                                       \
                for (unsigned tx_ = 0; tx_ < b_.x; tx_++)                                   \
                {                                                                           \
                    blockIdx.x = bx_; blockIdx.y = by_; threadIdx.x = tx_;                  \
                    kern(__VA_ARGS__);                                                      \
                }                                                                           \
    } while (0)

#endif
