/*
 * gpu_ipm_internal.h -- device-side data model of one shape-uniform batch of OCP-QPs.
 *
 * HBM layout: every per-instance quantity is a wave-tiled array (GArrT below),
 *     a.p[((instance / 64) * a.E + stage_offset + element) * 64 + instance % 64]
 * with the batch padded to a multiple of 64 (Bp): the 64 lanes of a wavefront (one instance
 * per lane) read 64 consecutive doubles = 512 B per load instruction, and everything one
 * wavefront touches in one array is one contiguous region.
 * Stage dims are padded to the compile-time (NX, NU) of the kernel instantiation:
 * padded variables get unit Hessian diagonal, zero gradient and zero dynamics rows /
 * columns, so they stay exactly zero and contribute exactly zero residual.
 *
 * Variables of a stage are ordered [u(NU); x(NX)] as in acados/HPIPM
 * (RSQrq block layout: acados/utils/print.c:234-325 in the reference).
 */
#ifndef GPU_IPM_INTERNAL_H_
#define GPU_IPM_INTERNAL_H_

#include <stdint.h>

#define GQP_MAX_ROWS 64 /* nb+ng per stage; inequality sides 2(nb+ng)+2ns <= 64 * AW (activity words per stage:
                           1 for the one-instance-per-lane kernels, up to 2 for the wave-per-instance ones) */

/* per-stage structure shared by the whole batch (read through scalar loads) */
struct GqpStage
{
    int nb, ng, ns;        /* box rows, general rows, slacks of this stage */
    int o_ct;              /* element offset of this stage in lam/t/rd/rm/dlam/dt/dvec */
    int o_s;               /* element offset of this stage in the slack arrays (2*ns entries) */
    int o_g;               /* row offset of this stage in DCt */
    int has_dyn;           /* k < N */
    int pad_;
    uint64_t bmask;        /* padded variable j carries a box row */
    uint64_t emask;        /* padded variable j is fixed by an equality-flagged bound */
    int8_t srev[GQP_MAX_ROWS]; /* slack index of constraint row (sorted order), -1 = hard */
};

/* The stage table is written by the host before the first launch and never by a kernel.  Kernels read it through the
 * CONSTANT address space: scalar loads (s_load, lgkmcnt) at every point of a kernel.  Through a plain pointer hipcc may use
 * scalar loads only up to the kernel's first store (the table might alias it); from there on it fetched the table with
 * vector loads, and the first use of a stage's fields -- offsets, masks -- waited for vmcnt to drain past that load: an
 * exposed round trip per stage and a stop for everything requested ahead of it (register prefetch, LDS-DMA).  Measured on
 * C2 (one-instance-per-lane kernels, same box): 58.6 -> 57.1 ms per solve. */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GQP_STAGE_VECTOR_LOADS) /* (development builds: the old fetch, for A/B runs) */
#define GQP_CONST_AS __attribute__((address_space(4)))
#else
#define GQP_CONST_AS
#endif
typedef const GQP_CONST_AS GqpStage *GqpStagePtr;
#define GQP_STAGE_REF const GQP_CONST_AS GqpStage &

struct GqpOpts
{
    double mu0, tol_stat, tol_eq, tol_ineq, tol_comp, alpha_min, tau_min, lam_min, t_min, reg_prim;
    int iter_max, pred_corr, cond_pred_corr, warm_start;
    int t0_init; /* cold start of (t, lam): 0 = (sqrt(mu0), sqrt(mu0)), 1 = (1, mu0), 2 = from the constraint residuals
                    (acados_ocp_options.py:1128-1143); 0 / 1 leave the primal iterate at zero */
    int ext_update; /* set by the host loop for the launch-per-sweep corrector sweeps of the sixteen-lanes families: the sweep leaves
                       its step length in GqpDev::apend and the step is applied by k_step_update, a launch of its own (ipm_kernels.hpp) */
};

/* One per-instance HBM array: `E` elements per instance, stored WAVE-TILED,
 *     p[((i / 64) * E + e) * 64 + (i % 64)]          (instance i, element e)
 * so that the 64 lanes of a wavefront still read 512 contiguous bytes per load instruction
 * AND consecutive elements / stages of one wavefront are adjacent in memory: a wave streams
 * its own contiguous region of every array (DRAM-row and TLB friendly), instead of touching
 * a new 512-KB-distant address per element as in a plain [element][instance] layout. */
template <class T>
struct GArrT
{
    T *p;
    int E;
    int aos; /* 0: wave-tiled (one instance per lane kernels); 1: instance-major p[i * E + e] -- the
                wave-per-instance kernels (ipm_kernels_wpi.hpp) stream one instance's block per wave */
};
typedef GArrT<double> GArr;
typedef GArrT<uint64_t> GArrU64;

/* all device pointers of one batch; passed by value to the kernels */
struct GqpDev
{
    int B, Bp, N, NX, NU, NG, NS;
    int AW;             /* activity words per stage in amask (1 or 2) */
    GqpStagePtr st; /* N+1 entries */
    /* problem data */
    /* Slot conventions that make the stage body uniform (no k<N / k>0 branches in the fast
     * kernels): dynamics arrays have N+1 stage slots, slot N is all zero ("x_{N+1} = 0*x+0*u+0");
     * pi/dpi have N+2 slots, slot s holds the multiplier of the dynamics that PRODUCE x_s
     * (acados pi[k] lives in slot k+1), slots 0 and N+1 are zero; ux/dux have N+2 slots, slot
     * N+1 zero; row arrays carry 16 spare elements so that a clamped dummy row index is
     * always readable. */
    GArr BAt;      /* [N+1][n*NX]  BAt[r*NX+c] = d x+_c / d v_r                     */
    GArr bvec;     /* [N+1][NX]                                                      */
    GArr RSQ;      /* [N+1][n(n+1)/2] packed lower, row-major packed: (r,c)->r(r+1)/2+c */
    GArr rq;       /* [N+1][n]                                                       */
    GArr dvec;     /* [sum nct] natural-sign bounds, order [lb lg ub ug lls lus]     */
    GArrU64 amask; /* [(N+1) * AW] per instance: bit e of stage k (word k*AW + e/64) set <=> inequality side e takes part */
    GArr DCt;      /* [sum ng][n]  row g: d(general row)/d v                         */
    GArr Zz;       /* [sum 2ns][2]: (Z, z) for sl then su                            */
    /* iterate */
    GArr ux;       /* [N+2][n] */
    GArr sv;       /* [sum 2ns] slack values sl then su */
    GArr pi;       /* [N+2][NX], see slot conventions */
    GArr lam, t;   /* [sum nct] */
    /* work */
    GArr rg, rgs, rb, rd, rm;
    GArr dux, dsv, dpi, dlam, dt;
    GArr pcorr;    /* [sum nct] dlam_aff*dt_aff of the affine step (fast path stores only the product) */
    GArr sD, sR;   /* [sum 2ns] per-slack D = Z + sum Gamma and r~ (condensed slack rhs) */
    GArr Lf;       /* [N+1][n(n+1)/2] Cholesky factors */
    GArr lf;       /* [N+1][n] */
    /* per instance scalars */
    double *res;   /* [4] */
    double *mu, *smu, *alpha, *obj;
    double *apend; /* step length of a step (dux, dpi, dlam, dt) that is computed but not applied yet: the corrector sweep of the
                      phase-ordered one-instance-per-lane kernels leaves it to the next factor sweep (ipm_kernels_box.hpp, "folded
                      update"); 0 = nothing pending.  Other families apply their steps themselves and leave it 0 */
    int *iter, *status;
    int *n_active; /* single counter: instances still iterating */
    double *stat;  /* [stat_rows][STAT_COLS][Bp_stat] for the first stat_inst instances */
    int stat_inst, stat_rows;
    /* sixteen-lanes-per-instance sweeps: row slot -> instance.  Null: slot = instance.  Otherwise the n_perm instances that
     * are still iterating, listed densely (rebuilt by k_active_perm as they converge): the grid shrinks with them and every
     * wave carries four live rows instead of paying a whole sweep for one */
    const int *perm;
    int n_perm;
};

#define GQP_STAT_COLS 20
/* Step to the boundary: the ratio test's alpha scaled as HPIPM's update does (its sources are absent from the reference tree: upstream
 * knowledge of UPDATE_VAR_QP) -- alpha * ((1 - alpha) 0.99 + alpha 0.9999999) below 1, the full step at 1.  Rounds 1-5 of this
 * restatement scaled by the constant 0.995 of older HPIPM versions: the same solutions (golden vectors unchanged), 2 - 5 % more
 * iterations (C2 mean 9.35 against 9.02 on 1,024 instances; oracle and device in lockstep, profiles/NOTES.md round 5). */
#define gqp_step_scale(alpha) ((alpha) < 1.0 ? (alpha) * ((1.0 - (alpha)) * 0.99 + (alpha) * 0.9999999) : (alpha))

#define GQP_RUNNING (-2)

#endif
