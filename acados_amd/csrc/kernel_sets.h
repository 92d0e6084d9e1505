/*
 * kernel_sets.h -- table of compiled kernel instantiations ("shape classes").
 * Small shapes are fully unrolled (register-resident stage blocks, gpu_batch.hip); large shapes
 * are compiled in their own translation unit with rolled loops (gpu_shapes_large.hip): their stage
 * blocks do not fit the register file of a one-instance-per-lane mapping and live in scratch --
 * correct and covered by the parity tests, but not the design point for those shapes.
 */
#ifndef KERNEL_SETS_H_
#define KERNEL_SETS_H_

#include "gpu_ipm_internal.h"
#include "pcond_kernels.hpp"

typedef void (*kern_opts_t)(GqpDev, GqpOpts);
typedef void (*kern_redo_t)(GqpDev, GqpOpts, int);
typedef void (*kern_plain_t)(GqpDev);

struct KernelSet
{
    int NX, NU, NG, NS;
    kern_opts_t init;
    kern_redo_t back_fact, back_rhs, fwd_aff, fwd_corr;
    kern_plain_t finalize;
    /* fast path for box-only QPs (ipm_kernels_box.hpp; ipm_kernels_box_small.hpp for nu + nx <= 6); index = XBOX (any box
     * row on a state) */
    kern_redo_t box_fact[2], box_rhs[2], box_fwd_aff[2], box_fwd_corr[2];
    kern_plain_t box_finalize;
    /* the ipm_kernels_box.hpp kernels for every shape (ACADOS_AMD_KB_SMALL=0: cross-check of the small-block kernels) */
    kern_redo_t kb_fact[2], kb_rhs[2], kb_fwd_aff[2], kb_fwd_corr[2];
};

#define GQP_KSET(NX, NU, NG, NS)                                                               \
    {NX, NU, NG, NS, gqp::k_init<NX, NU, NG, NS>, gqp::k_backward<NX, NU, NG, NS, true>,       \
     gqp::k_backward<NX, NU, NG, NS, false>, gqp::k_forward<NX, NU, NG, NS, false>,            \
     gqp::k_forward<NX, NU, NG, NS, true>, gqp::k_finalize<NX, NU, NG, NS>,                    \
     {gqp::kb_factor_for<NX, NU, false>(), gqp::kb_factor_for<NX, NU, true>()},                \
     {gqp::kb_backrhs_for<NX, NU, false>(), gqp::kb_backrhs_for<NX, NU, true>()},              \
     {gqp::kb_forward_for<NX, NU, false, false>(), gqp::kb_forward_for<NX, NU, true, false>()}, \
     {gqp::kb_forward_for<NX, NU, false, true>(), gqp::kb_forward_for<NX, NU, true, true>()},  \
     gqp::kb_finalize<NX, NU>,                                                                 \
     {gqp::kb_factor<NX, NU, false>, gqp::kb_factor<NX, NU, true>},                            \
     {gqp::kb_backrhs<NX, NU, false>, gqp::kb_backrhs<NX, NU, true>},                          \
     {gqp::kb_forward<NX, NU, false, false>, gqp::kb_forward<NX, NU, true, false>},            \
     {gqp::kb_forward<NX, NU, false, true>, gqp::kb_forward<NX, NU, true, true>}}

/* partial condensing: parent shape (NX, NU), blocks of at most BSMAX stages -> child shape
 * (NX, BSMAX*NU); kernels in pcond_kernels.hpp */
typedef void (*kern_pcond_t)(GqpDev, GqpDev, gqp::PcondMap);
struct PcondSet
{
    int NX, NU, BSMAX;
    kern_pcond_t cond, expand;
};
#define GQP_PCOND(NX, NU, BS) {NX, NU, BS, gqp::k_pcond<NX, NU, BS>, gqp::k_pexpand<NX, NU, BS>}
extern const PcondSet g_pcond_sets[];
extern const int g_n_pcond_sets;

extern const KernelSet g_ksets_large[];
extern const int g_n_ksets_large;

#endif
