/*
 * ocp_qp_host_internal.h -- what the two host translation units share (ocp_qp_host.cpp: containers + the inner
 * qp_solver_config plugin; ocp_qp_xcond.cpp: the condensing module, the outer xcond-solver vtable and the
 * acados_c-shaped convenience layer).  Host code only.
 */
#ifndef OCP_QP_HOST_INTERNAL_H_
#define OCP_QP_HOST_INTERNAL_H_

#include <stdint.h>

#include "acados_amd/ocp_qp_gpu_batch.h"
#include "acados_amd/ocp_qp_interface.h"

namespace gqp_host
{

inline char *align8(char *p) { return (char *) (((uintptr_t) p + 7) & ~(uintptr_t) 7); }

/* what the xcond level asks of one evaluate: travels as an ARGUMENT from the caller's opts / memory (no state outside
 * opts / mem / work, SURVEY 8b "Threading") */
struct cond_request
{
    int N2;                /* requested number of condensed stages; <= 0 or >= N: full space */
    const int *block_size; /* N2 + 1 entries or NULL (ocp_qp_partial_condensing.c:305-313) */
    int init_from_qp_out;  /* initialize_next_xcond_qp_from_qp_out: qp_out holds the guess (pi, lam, t) */
    int phase;             /* 0: condense + solve + expand; 1: condense_lhs only; 2: condense_rhs + solve + expand */
    int dense;             /* FULL_CONDENSING past what one condensed stage may carry: condense to ONE dense problem inside the solve
                              (device option "full_dense", dense_kernels.hpp); N2 is 0 then */
};

/* the evaluate of the inner plugin with the condensing request made explicit; `cr` may be NULL */
int gpu_ipm_evaluate_impl(void *config, int n, void **qp_in, void **qp_out, void *opts, void **mem, void *work,
                          int *status, const cond_request *cr);

/* (qp_in [, qp_out]) of one QP -> a one-instance device batch (created / re-created when the structure changes);
 * used by the condensing module and the residual workspace */
struct single_batch
{
    ocp_qp_gpu_batch *batch;
    int *sig;     /* structure signature of the QP the batch was built for (malloc'ed with the batch) */
    int sig_len;
    int generation; /* bumped whenever `batch` is (re)created: a new batch often lands on the old one's address */
};
int structure_sig_len(const ocp_qp_dims *d);
int structure_sig_fill(const ocp_qp_in *in, int *dst);
/* returns 0 on success; packs every numeric field of `in` */
int single_batch_load_in(single_batch *sb, const ocp_qp_in *in);
/* iterate of `out` -> device (x u sl su pi lam t) / device -> `out` */
void single_batch_push_out(ocp_qp_gpu_batch *b, const ocp_qp_dims *d, const ocp_qp_out *out);
void single_batch_pull_out(ocp_qp_gpu_batch *b, const ocp_qp_dims *d, ocp_qp_out *out);
/* numeric fields of a (condensed) device batch -> a host container of the same dims, incl. the index sets */
void single_batch_read_in(ocp_qp_gpu_batch *b, ocp_qp_in *in, int what); /* what: 1 matrices, 2 vectors, 3 both */
void single_batch_free(single_batch *sb);

} // namespace gqp_host

#endif
