/*
 * ocp_qp_gpu_batch.h -- C-ABI of the MI355X batched OCP-QP solver (device-resident batch).
 *
 * This is the batch entry point the reference does not have: it replaces the OpenMP loop
 *   for (i < N_batch) ocp_nlp_solve(capsule[i]) -> ... -> d_ocp_qp_ipm_solve(qp_in, qp_out, ...)
 * (interfaces/acados_template/acados_template/c_templates_tera/acados_solver.in.c:3222-3243,
 *  acados/ocp_qp/ocp_qp_hpipm.c:347 in /root/reference) for the QP-solve part: `n_batch`
 * structurally identical QPs (same dims / idxb / idxs_rev / idxe, different numbers) are
 * packed once into an element-major, instance-minor HBM layout and every IPM iteration
 * advances all of them in a handful of kernel launches.
 *
 * Plain pointers and sizes only; no torch types.  Field names and data conventions are
 * those of `ocp_qp_in_set` / `d_ocp_qp_set` and `ocp_qp_out_get`
 * (interfaces/acados_c/ocp_qp_interface.c:405-480; Python driver
 * acados_ocp_qp_solver.py:277-292): column-major matrices, natural-sign bounds,
 * lam/t ordered [lbu lbx lg ubu ubx ug ls us].
 */
#ifndef ACADOS_AMD_OCP_QP_GPU_BATCH_H_
#define ACADOS_AMD_OCP_QP_GPU_BATCH_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ocp_qp_gpu_batch ocp_qp_gpu_batch;

/* acados return codes (acados/utils/types.h:74-87) as produced per instance */
#define ACADOS_AMD_SUCCESS 0
#define ACADOS_AMD_NAN_DETECTED 1
#define ACADOS_AMD_MAXITER 2
#define ACADOS_AMD_MINSTEP 3
#define ACADOS_AMD_INFEASIBLE 9

/* Create a batch of `n_batch` QPs with the per-stage dims of ocp_qp_dims
 * (acados/ocp_qp/ocp_qp_common.h:49; arrays of length N+1).  `device` < 0 keeps the
 * current HIP device.  Returns NULL (and prints the reason) if no kernel covers the shape (limits: nu + nx <= 64 per
 * stage, ng <= 32, ns <= 32, at most 128 inequality sides 2(nb+ng)+2ns per stage) or no GPU is present. */
ocp_qp_gpu_batch *ocp_qp_gpu_batch_create(int N, const int *nx, const int *nu, const int *nbx,
                                          const int *nbu, const int *ng, const int *ns,
                                          int n_batch, int device);
void ocp_qp_gpu_batch_destroy(ocp_qp_gpu_batch *b);

/* Structure shared by all instances (integers): "idxb" (nb entries, indices into [u;x]),
 * "idxbu", "idxbx", "idxs_rev" (nb+ng entries, -1 = hard), "idxe" (`n` entries, positions
 * in the bound list of equality-flagged x bounds; sets nbxe = n).  Must be set before any
 * numeric data of that stage. */
int ocp_qp_gpu_batch_set_int(ocp_qp_gpu_batch *b, const char *field, int stage, const int *value, int n);

/* Numeric data.  `data` holds n_batch consecutive blocks of the field's length:
 * data[i*len + e], e in acados column-major order.  Fields: A B b Q S R q r lbx ubx lbu ubu
 * lg ug C D Zl Zu zl zu lls lus  *_mask (lbx ubx lbu ubu lg ug lls lus)  and, for warm
 * starts, x u sl su pi lam t.  `is_device` != 0: `data` is a device pointer (e.g. a torch
 * tensor's data_ptr()) and no host transfer takes place.  `stage` = -1 applies the same
 * block to every stage that has the field (data is then read once per stage). */
int ocp_qp_gpu_batch_set(ocp_qp_gpu_batch *b, const char *field, int stage, const double *data, int is_device);

/* Bulk pack / unpack: ONE host->device copy and ONE launch move every numeric field of every stage
 * (the acados adapter's evaluate path: all member arrays of n ocp_qp_in in one go).  The per-instance
 * blob is the concatenation, stage by stage, of the fields
 *   input  (output = 0): A B b Q S R q r lbu ubu lbx ubx lg ug C D Zl Zu zl zu lls lus and the 8 *_mask
 *   output (output = 1): u x sl su pi lam t
 * (fields of length 0 at a stage are skipped).  _bulk_len gives the doubles per instance,
 * _bulk_offset the position (and length) of one field; for an equality-flagged lbx the blob has a second
 * segment "lbx#value" right after "lbx" (same numbers: the bound is also the value of the variable).
 * blob[i * len + off + e], e in acados column-major order. */
int ocp_qp_gpu_batch_bulk_len(ocp_qp_gpu_batch *b, int output);
int ocp_qp_gpu_batch_bulk_offset(ocp_qp_gpu_batch *b, int output, const char *field, int stage, int *len);
int ocp_qp_gpu_batch_set_bulk(ocp_qp_gpu_batch *b, const double *blob, int is_device);
/* the VECTOR fields alone (b q r, bounds, zl zu, masks): blob layout of _bulk_len / _bulk_offset with `output` = 2.  The matrices stay
 * what the last _set_bulk brought: the host->device traffic of an RTI feedback step whose preparation step condensed the matrices
 * (replaces the vector part of d_ocp_qp_set_* between ocp_qp_xcond_solver.c:591-620 and :623-669; 12.6 KB instead of 85 KB per C2-shaped QP) */
int ocp_qp_gpu_batch_set_bulk_vec(ocp_qp_gpu_batch *b, const double *blob, int is_device);
int ocp_qp_gpu_batch_get_bulk(ocp_qp_gpu_batch *b, double *blob, int is_device);
/* ZERO-COPY GATHER (the batch entries of the acados-side adapter, replaces the host pass over n ocp_qp_in structs -- the
 * d_ocp_qp_set_* / blasfeo_unpack_* loops of an adapter, ocp_qp_clarabel.c:205-683 -- by device reads over PCIe): the device reads every
 * instance's QP data from the CALLER'S OWN host memory.  _host_register pins a block of it and maps it for the device (hipHostRegister:
 * the device pointer equals the host pointer; the block stays registered until _host_unregister).  _gather_tables: once per class and
 * blob (`which` 0: the input blob, 2: its vector part) the word tables -- word w is element w_off[w] of source array w_slot[w] (0 <= slot < P,
 * e.g. the panel-major storage of one BLASFEO matrix) and goes to position w_pos[w] of the blob (layout of _bulk_len / _bulk_offset), negated
 * where w_neg[w]; a position no word writes is zero; sort the words by (slot, offset) so that consecutive device lanes read consecutive host
 * addresses.  _gather_run: per call
 * the n_batch * P source addresses (ptrs[i * P + slot], each inside a registered block) -- gather into a device-side blob, then exactly what
 * _set_bulk / _set_bulk_vec do with it.  0 on success, -1 on a device failure or a table that does not fit the blob. */
int ocp_qp_gpu_host_register(void *p, size_t bytes);
int ocp_qp_gpu_host_unregister(void *p);
int ocp_qp_gpu_batch_gather_tables(ocp_qp_gpu_batch *b, int which, int P, int n_words, const int *w_slot, const int *w_off, const int *w_pos,
                                   const unsigned char *w_neg);
int ocp_qp_gpu_batch_gather_run(ocp_qp_gpu_batch *b, int which, const void *const *ptrs);
/* _set_bulk in pieces (the batch entries of the acados-side adapter: the host->device copy of instance range j overlaps with the host
 * threads still unpacking range j + 1): `blob_chunk` = instance `first` of the caller's pinned blob, `count` instances, asynchronous;
 * _set_bulk_staged scatters what the chunks brought (all n_batch instances must have been handed over) and waits */
int ocp_qp_gpu_batch_set_bulk_chunk(ocp_qp_gpu_batch *b, const double *blob_chunk, int first, int count);
int ocp_qp_gpu_batch_set_bulk_staged(ocp_qp_gpu_batch *b);
/* the QP DATA of the batch in the INPUT blob layout (inverse of _set_bulk): how a condensing module on acados' types reads the
 * condensed QP of a child batch (ocp_qp_gpu_batch_condense) into the container the reference's orchestration hands to the QP
 * solver next (xcond_qp_in, ocp_qp_partial_condensing.c:523-556; integration/ocp_qp_gpu_pcond.c) */
int ocp_qp_gpu_batch_get_bulk_in(ocp_qp_gpu_batch *b, double *blob, int is_device);
/* an iterate in the OUTPUT blob layout written into the batch (inverse of _get_bulk): the starting point of a hot start */
int ocp_qp_gpu_batch_set_bulk_out(ocp_qp_gpu_batch *b, const double *blob, int is_device);

/* Options by name, as ocp_qp_xcond_solver_opts_set forwards them: iter_max
 * tol_stat tol_eq tol_ineq tol_comp warm_start mu0 alpha_min tau_min reg_prim
 * cond_pred_corr print_level t0_init t0_min lam0_min (lower clips of t / lam at a hot start) update_fact_exit hpipm_mode ric_alg cond_N (partial condensing to N2 blocks) cond_block_size (int[N2+1], after cond_N; ocp_qp_partial_condensing.c:305-313; a non-zero last entry becomes one more block in front of an input-free terminal stage) profile.  int* or double* or char* as in
 * acados.  Unknown field: message + return -1. */
int ocp_qp_gpu_batch_opts_set(ocp_qp_gpu_batch *b, const char *field, const void *value);

/* Solve all instances (one IPM iteration = 4-6 launches over the whole batch) on the
 * batch's stream.  Returns the number of instances whose status is not ACADOS_SUCCESS. */
int ocp_qp_gpu_batch_solve(ocp_qp_gpu_batch *b);

/* RTI split (ocp_qp_xcond_solver.c:591-669): condense the matrix part while waiting for the new
 * initial state, then condense the vector part and solve.  With cond_N == N (or a QP whose condensed stages
 * would exceed the limits of INTEGRATION.md) condense_lhs is a no-op and the second call is a plain solve.
 * The matrix-dependent condensed blocks stay resident in HBM between the two calls. */
int ocp_qp_gpu_batch_condense_lhs(ocp_qp_gpu_batch *b);
int ocp_qp_gpu_batch_condense_rhs_and_solve(ocp_qp_gpu_batch *b);
/* Condensing-only boundary: ocp_qp_condense / ocp_qp_expand of interfaces/acados_c/condensing_interface.h:73-75
 * (the `condensing(qp_in, xcond_qp_in, ...)` / `expansion(...)` slots of ocp_qp_xcond_config,
 * acados/ocp_qp/ocp_qp_common.h:84-107).  _condense runs partial condensing (opts cond_N, cond_block_size) and
 * returns the condensed QP as a batch object OWNED BY `b` (do not destroy it; NULL when the QP is not condensed --
 * cond_N == N or beyond the limits of INTEGRATION.md): its data are readable with ocp_qp_gpu_batch_get / _get_dims /
 * _get_int (what memory_get "xcond_qp_in" / dims_get "xcond_dims" answer in the reference,
 * ocp_qp_partial_condensing.c:138-155, :467-504), it can be solved with ocp_qp_gpu_batch_solve or receive a
 * solution computed elsewhere (set "x" "u" "pi" "lam" "t" "sl" "su").  _expand maps the condensed batch's current
 * solution back to the stages of `b` (x, u, sl, su, pi, lam, t). */
ocp_qp_gpu_batch *ocp_qp_gpu_batch_condense(ocp_qp_gpu_batch *b);
int ocp_qp_gpu_batch_expand(ocp_qp_gpu_batch *b);
/* the other slots of ocp_qp_xcond_config (acados/ocp_qp/ocp_qp_common.h:84-107): _condense_rhs = vector part only on top of
 * a resident matrix part (`condense_rhs`, ocp_qp_partial_condensing.c:602-630; after ocp_qp_gpu_batch_condense_lhs), returns
 * the condensed batch; _condense_sol = the current iterate of `b` restated in the condensed variables and written into the
 * condensed batch (`condense_qp_out`, :559-571): block inputs stacked, first state of each block, pi at the block
 * boundaries, rows and slacks with their multipliers */
ocp_qp_gpu_batch *ocp_qp_gpu_batch_condense_rhs(ocp_qp_gpu_batch *b);
ocp_qp_gpu_batch *ocp_qp_gpu_batch_condensed(ocp_qp_gpu_batch *b); /* the condensed batch owned by `b`, or NULL */
int ocp_qp_gpu_batch_condense_sol(ocp_qp_gpu_batch *b);
/* "N" "n_batch" (one int) or "nx" "nu" "nbx" "nbu" "nb" "ng" "ns" "nbxe" (N+1 ints) */
int ocp_qp_gpu_batch_get_dims(ocp_qp_gpu_batch *b, const char *field, int *out);
/* "idxb" (nb) "idxs_rev" (nb+ng) "idxe": returns the number of entries written */
int ocp_qp_gpu_batch_get_int(ocp_qp_gpu_batch *b, const char *field, int stage, int *out);

/* Results, same blocked convention as _set: x u sl su pi lam t per stage; and the Riccati factor of
 * the last factorisation: ric_L ((nu+nx)^2, column-major lower Cholesky factor [Lr 0; Ls Lx] of the
 * stage matrix) and ric_l (nu+nx), from which P = Lx Lx', p = Lx lx, K = -Lr^-T Ls', k = -Lr^-T lr
 * (the getters of ocp_qp_hpipm.c:417-478) follow. */
int ocp_qp_gpu_batch_get(ocp_qp_gpu_batch *b, const char *field, int stage, double *data, int is_device);
/* Solution sensitivities with the factorisation at the solution: what d_ocp_qp_ipm_sens_frw / _sens_adj do behind
 * ocp_qp_hpipm_eval_forw_sens / _adj_sens (acados/ocp_qp/ocp_qp_hpipm.c:481-506; callers
 * acados/ocp_nlp/ocp_nlp_common.c:4091, 4141).  A seed is the derivative of the problem data w.r.t. a parameter p:
 * "seed_q" "seed_r" (gradient), "seed_b" (dynamics offset), "seed_lbu" "seed_ubu" "seed_lbx" "seed_ubx" "seed_lg"
 * "seed_ug" (bounds, natural sign; for an equality-flagged bound -- x0 -- the seed is the derivative of the variable
 * itself, ocp_nlp_common.c:4057-4064).  `data`: host pointer, n_batch blocks of the field's length.  The first seed
 * after a solve opens a new all-zero seed set; _sens_solve runs one rhs-only backward and one forward sweep and leaves
 * d(solution)/dp in the fields "sens_u" "sens_x" "sens_sl" "sens_su" "sens_pi" "sens_lam" "sens_t" of
 * ocp_qp_gpu_batch_get.  The KKT matrix is symmetric, so the adjoint solve of a seed in (q, r) is the same call.
 * The sweeps exist in the wave-per-instance / sixteen-lanes kernel families; a one-instance-per-lane batch is walked
 * in slices of 16,384 instances through a sub-batch of those families (ACADOS_AMD_SENS_SLICE changes the slice).
 * After a partially condensed solve the sensitivities are computed in the full space at the expanded solution. */
int ocp_qp_gpu_batch_sens_set(ocp_qp_gpu_batch *b, const char *field, int stage, const double *data);
int ocp_qp_gpu_batch_sens_solve(ocp_qp_gpu_batch *b);
/* The same in bulk -- every seed of every instance in ONE host->device copy and one launch, every direction back in one
 * launch and one copy: the batched eval_forw_sens / eval_adj_sens (acados_solver.in.c:3292-3337 loops the single-QP
 * slots of ocp_qp_hpipm.c:481-506 over the capsules).  Seed blob (output = 0), per instance the concatenation, stage by
 * stage, of  seed_r seed_q seed_zl seed_zu | seed_b | seed_lbu seed_lbx seed_lg seed_ubu seed_ubx seed_ug seed_lls seed_lus
 * i.e. d_ocp_qp_seed's seed_g[k], seed_b[k], seed_d[k] one after the other, NATURAL sign (acados stores the upper part of
 * seed_d negated like d, ocp_nlp_common.c:4078-4081: its adapter flips it).  _sens_set_bulk opens a fresh seed set
 * (whatever single seeds were set before are overwritten) ; _sens_solve as above; _sens_get_bulk returns the directions
 * in the OUTPUT blob layout of ocp_qp_gpu_batch_get_bulk (sens_u sens_x sens_sl sens_su sens_pi sens_lam sens_t).
 * _sens_bulk_len / _sens_bulk_offset: doubles per instance and position of one field ("seed_q" ... / "sens_x" ...). */
int ocp_qp_gpu_batch_sens_bulk_len(ocp_qp_gpu_batch *b, int output);
int ocp_qp_gpu_batch_sens_bulk_offset(ocp_qp_gpu_batch *b, int output, const char *field, int stage, int *len);
int ocp_qp_gpu_batch_sens_set_bulk(ocp_qp_gpu_batch *b, const double *blob, int is_device);
int ocp_qp_gpu_batch_sens_get_bulk(ocp_qp_gpu_batch *b, double *blob, int is_device);

/* KKT residuals of an arbitrary (qp_in, qp_out): what ocp_qp_res_compute -> d_ocp_qp_res_compute and
 * ocp_qp_res_compute_nrm_inf do (acados/ocp_qp/ocp_qp_common.c:559-667; wrapper ocp_qp_inf_norm_residuals,
 * interfaces/acados_c/ocp_qp_interface.c:642-650; asserted by test/ocp_qp/test_qpsolvers.cpp:240-251).  _res_compute
 * evaluates the QP data and the iterate (x u sl su pi lam t) that are in HBM right now -- written by a solve, by
 * ocp_qp_gpu_batch_set, or by an expansion -- in ONE launch of a kernel that shares nothing with the IPM sweeps, and
 * leaves the residual vectors readable through ocp_qp_gpu_batch_get: "res_g" (nu+nx, stationarity w.r.t. [u; x]),
 * "res_gs" (2ns, w.r.t. [sl; su]), "res_b" (nx_{k+1}), "res_d" and "res_m" (2(nb+ng+ns), order [lb lg ub ug ls us]).
 * _res_nrm_inf copies res[i*4 + q] = inf-norm of (res_g, res_b, res_d, res_m) of instance i to the host (and runs
 * _res_compute first if it has not run). */
int ocp_qp_gpu_batch_res_compute(ocp_qp_gpu_batch *b);
int ocp_qp_gpu_batch_res_nrm_inf(ocp_qp_gpu_batch *b, double *res);

/* per-instance: "status" "iter" (int), "res_stat" "res_eq" "res_ineq" "res_comp" "mu" "obj" (double) */
int ocp_qp_gpu_batch_get_info(ocp_qp_gpu_batch *b, const char *field, void *data);
/* HPIPM-shaped statistics of instance `inst` (< 64): (iter+1) x 20, row-major */
int ocp_qp_gpu_batch_get_stat(ocp_qp_gpu_batch *b, int inst, double *stat, int max_rows);
/* scalars of the last solve: "time_tot" (s, HIP events around the whole solve),
 * "time_pack" (s, accumulated since the last solve), "iter_max_batch", "launches" */
double ocp_qp_gpu_batch_get_scalar(ocp_qp_gpu_batch *b, const char *field);
/* ---- multi-GPU: instances are independent, one process per GPU, no collective on the data path.  The only
 * exchange is ONE all-gather of the results after the solve: RCCL over xGMI, device buffers, on the batch's stream.
 *   _comm_unique_id   rank 0: fills 128 bytes (ncclUniqueId) that the host program hands to every rank
 *   _comm_create      every rank (collective): communicator of n_ranks processes; `device` < 0 keeps the current device
 *   _batch_gather     every rank (collective), same n_batch on every rank: gathers into DEVICE buffers
 *                       sol_all  [n_ranks][n_batch][bulk_len(output)]  u x sl su pi lam t of every instance (the blob layout
 *                                of ocp_qp_gpu_batch_get_bulk; 12,720 B per instance for the C2 shape)
 *                       info_all [n_ranks][n_batch][2]  int32 (status, iter)
 *                       time_all [n_ranks]              solve time of the last solve of each rank (s)
 * RCCL is bound at run time (the copy already loaded by the process if any).  Returns 0, or -1 with a message. ---- */
typedef struct ocp_qp_gpu_comm ocp_qp_gpu_comm;
int ocp_qp_gpu_comm_unique_id(void *id128);
ocp_qp_gpu_comm *ocp_qp_gpu_comm_create(const void *id128, int n_ranks, int rank, int device);
void ocp_qp_gpu_comm_destroy(ocp_qp_gpu_comm *c);
int ocp_qp_gpu_batch_gather(ocp_qp_gpu_batch *b, ocp_qp_gpu_comm *c, double *sol_all, int *info_all, double *time_all);
/* the same payload to ONE rank (ncclSend / ncclRecv in one group): every other rank moves 1x its payload over its xGMI
 * link to `root` instead of receiving n_ranks x; sol_all / info_all / time_all are read on `root` only (NULL elsewhere) */
int ocp_qp_gpu_batch_gather_root(ocp_qp_gpu_batch *b, ocp_qp_gpu_comm *c, int root, double *sol_all, int *info_all, double *time_all);
/* UNEVEN shards (n_total instances not a multiple of n_ranks; acados_solver.in.c:3222-3243 shards a batch of any size over
 * its threads): counts[r] = instances on rank r (counts[own rank] must be this batch's n_batch; NULL = the same n_batch
 * everywhere).  Receive buffers are in rank order without padding -- rank r's instances start at instance offset
 * counts[0] + ... + counts[r-1]: sol_all [sum counts][bulk_len], info_all [sum counts][2], time_all [n_ranks].  root < 0:
 * every rank receives; root >= 0: that rank only.  Uneven counts travel as exact-size sends / receives in one group. */
int ocp_qp_gpu_batch_gather_v(ocp_qp_gpu_batch *b, ocp_qp_gpu_comm *c, int root, const int *counts, double *sol_all, int *info_all,
                              double *time_all);
/* A communicator over a transport the HOST PROGRAM supplies instead of RCCL (an MPI program; the CPU test tier, which runs
 * the gather's packing / offset / ordering logic over torch.distributed's gloo).  Semantics of the five entries are NCCL's:
 * all_gather: every rank contributes `count` elements, recvbuf holds n_ranks x count in rank order; send / recv: matched
 * point-to-point transfers that may be issued in any order between group_start and group_end and complete by group_end;
 * buffers are device pointers of the batch's device, `stream` is the batch's hipStream_t; return 0 on success.
 * dtype codes are ncclDataType_t's. */
#define GQP_COMM_I32 2
#define GQP_COMM_F64 8
typedef struct ocp_qp_gpu_comm_ops
{
    void *ctx;
    int (*all_gather)(void *ctx, const void *sendbuf, void *recvbuf, size_t count, int dtype, void *stream);
    int (*send)(void *ctx, const void *buf, size_t count, int dtype, int peer, void *stream);
    int (*recv)(void *ctx, void *buf, size_t count, int dtype, int peer, void *stream);
    int (*group_start)(void *ctx);
    int (*group_end)(void *ctx);
} ocp_qp_gpu_comm_ops;
ocp_qp_gpu_comm *ocp_qp_gpu_comm_create_from_ops(const ocp_qp_gpu_comm_ops *ops, int n_ranks, int rank);

/* pinned (page-locked) host memory for callers that stage their own bulk blobs -- the acados-side adapter is plain C
 * and links no HIP; NULL (with a message) on failure */
void *ocp_qp_gpu_host_alloc(size_t bytes);
void ocp_qp_gpu_host_free(void *p);

/* bytes of HBM held by the batch */
size_t ocp_qp_gpu_batch_bytes(const ocp_qp_gpu_batch *b);
/* raw HIP stream handle (hipStream_t) the batch launches on, for event timing by the caller */
void *ocp_qp_gpu_batch_stream(ocp_qp_gpu_batch *b);
/* name of the kernel instantiation serving this batch, e.g. "1tpi<NX=8,NU=3,NG=0,NS=0>" */
const char *ocp_qp_gpu_batch_kernel_name(const ocp_qp_gpu_batch *b);

#ifdef __cplusplus
}
#endif
#endif
