/*
 * gpu_shapes_large.hip -- kernel instantiations for stage blocks that exceed the register file of
 * the one-instance-per-lane mapping (nu+nx > 16).  Compiled with rolled loops (GQP_NO_UNROLL): the
 * per-lane blocks are addressed dynamically and live in scratch.  Serves configurations C4
 * (nx=24, nu=3, ng=4, ns=8) and the nx=24 classes of C5 for parity; the register-resident
 * multi-lane mapping these shapes need for speed is the next design step.
 */
#define GQP_NO_UNROLL 1
#include <hip/hip_runtime.h>

#include "ipm_kernels.hpp"
#include "ipm_kernels_box.hpp"
#include "ipm_kernels_box_small.hpp"
#include "kernel_sets.h"

const KernelSet g_ksets_large[] = {
    GQP_KSET(24, 3, 4, 8),
    GQP_KSET(24, 6, 0, 0),
    GQP_KSET(8, 15, 0, 0), /* C3: N=50 nx=8 nu=3 condensed to N2=10 blocks of 5 */
};
const PcondSet g_pcond_sets[] = {
    GQP_PCOND(8, 3, 5),
    GQP_PCOND(4, 1, 4),
};
const int g_n_pcond_sets = (int) (sizeof(g_pcond_sets) / sizeof(g_pcond_sets[0]));
const int g_n_ksets_large = (int) (sizeof(g_ksets_large) / sizeof(g_ksets_large[0]));
