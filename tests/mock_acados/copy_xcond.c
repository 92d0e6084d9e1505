/*
 * TEST INFRASTRUCTURE ONLY -- an ocp_qp_xcond_config (acados/ocp_qp/ocp_qp_common.h:84-107, 20 slots) that "condenses" with
 * N2 = N: the condensed QP is a COPY of the original one in the module's own containers, the expansion copies the solution
 * back.  That is what the reference's partial condensing amounts to at its default N2 = N (ocp_qp_partial_condensing.c:243-265
 * runs HPIPM's d_part_cond_qp_cond with blocks of one stage) -- HPIPM is absent, so the reference's module cannot be built;
 * this stand-in lets the reference's UNMODIFIED ocp_qp_xcond_solver.c run its whole orchestration (dims, opts routing, memory
 * carving, condense -> qp_solver->evaluate -> expand, info, memory_get) around this repository's plugin.  The containers are
 * created by the reference's own ocp_qp_in_assign / ocp_qp_out_assign / ocp_qp_seed_assign (ocp_qp_common.c).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acados/ocp_qp/ocp_qp_common.h"
#include "acados/utils/mem.h"
#include "acados/utils/timing.h"
#include "blasfeo/include/blasfeo_d_aux.h"

typedef struct { ocp_qp_dims *orig_dims; } copy_xcond_dims;
typedef struct { int N2; int ric_alg; } copy_xcond_opts;
typedef struct
{
    ocp_qp_in *xcond_qp_in;
    ocp_qp_out *xcond_qp_out;
    ocp_qp_seed *xcond_seed;
    qp_info *qp_out_info; /* = xcond_qp_out->misc: what the inner solver fills */
    double time_qp_xcond;
} copy_xcond_memory;

static acados_size_t cx_dims_calculate_size(void *config, int N) { return sizeof(copy_xcond_dims) + ocp_qp_dims_calculate_size(N) + 16; }
static void *cx_dims_assign(void *config, int N, void *raw)
{
    char *c = (char *) raw;
    copy_xcond_dims *d = (copy_xcond_dims *) c;
    c += sizeof(copy_xcond_dims);
    align_char_to(8, &c);
    d->orig_dims = ocp_qp_dims_assign(N, c);
    return d;
}
static void cx_dims_set(void *config, void *dims_, int stage, const char *field, int *value)
{
    ocp_qp_dims_set(config, ((copy_xcond_dims *) dims_)->orig_dims, stage, field, value);
}
static void cx_dims_get(void *config, void *dims_, const char *field, void *value)
{
    if (!strcmp(field, "xcond_dims")) *(void **) value = ((copy_xcond_dims *) dims_)->orig_dims;
    else { printf("copy_xcond: dims_get field %s\n", field); exit(1); }
}
static acados_size_t cx_opts_calculate_size(void *dims) { return sizeof(copy_xcond_opts) + 8; }
static void *cx_opts_assign(void *dims, void *raw) { char *c = (char *) raw; align_char_to(8, &c); return c; }
static void cx_opts_initialize_default(void *dims_, void *opts_)
{
    copy_xcond_opts *o = (copy_xcond_opts *) opts_;
    o->N2 = ((copy_xcond_dims *) dims_)->orig_dims->N;
    o->ric_alg = 0;
}
static void cx_opts_update(void *dims, void *opts) {}
static void cx_opts_set(void *opts_, const char *field, void *value)
{
    copy_xcond_opts *o = (copy_xcond_opts *) opts_;
    if (!strcmp(field, "N")) o->N2 = *(int *) value;
    else if (!strcmp(field, "ric_alg")) o->ric_alg = *(int *) value;
    else { printf("copy_xcond: opts_set field %s\n", field); exit(1); }
}
static acados_size_t cx_memory_calculate_size(void *dims_, void *opts_)
{
    ocp_qp_dims *d = ((copy_xcond_dims *) dims_)->orig_dims;
    if (((copy_xcond_opts *) opts_)->N2 != d->N) { printf("copy_xcond: only N2 = N\n"); exit(1); }
    return sizeof(copy_xcond_memory) + ocp_qp_in_calculate_size(d) + ocp_qp_out_calculate_size(d) + ocp_qp_seed_calculate_size(d) + 64;
}
static void *cx_memory_assign(void *dims_, void *opts_, void *raw)
{
    ocp_qp_dims *d = ((copy_xcond_dims *) dims_)->orig_dims;
    char *c = (char *) raw;
    copy_xcond_memory *m = (copy_xcond_memory *) c;
    c += sizeof(copy_xcond_memory);
    align_char_to(8, &c);
    m->xcond_qp_in = ocp_qp_in_assign(d, c); c += ocp_qp_in_calculate_size(d);
    align_char_to(8, &c);
    m->xcond_qp_out = ocp_qp_out_assign(d, c); c += ocp_qp_out_calculate_size(d);
    align_char_to(8, &c);
    m->xcond_seed = ocp_qp_seed_assign(d, c); c += ocp_qp_seed_calculate_size(d);
    m->qp_out_info = (qp_info *) m->xcond_qp_out->misc;
    m->time_qp_xcond = 0.0;
    return m;
}
static void cx_memory_get(void *config, void *mem_, const char *field, void *value)
{
    copy_xcond_memory *m = (copy_xcond_memory *) mem_;
    if (!strcmp(field, "xcond_qp_in")) *(ocp_qp_in **) value = m->xcond_qp_in;
    else if (!strcmp(field, "xcond_qp_out")) *(ocp_qp_out **) value = m->xcond_qp_out;
    else if (!strcmp(field, "xcond_seed")) *(ocp_qp_seed **) value = m->xcond_seed;
    else if (!strcmp(field, "qp_out_info")) *(qp_info **) value = m->qp_out_info;
    else if (!strcmp(field, "time_qp_xcond")) *(double *) value = m->time_qp_xcond;
    else { printf("copy_xcond: memory_get field %s\n", field); exit(1); }
}
static acados_size_t cx_workspace_calculate_size(void *dims, void *opts) { return 0; }

static void copy_vectors(ocp_qp_in *a, ocp_qp_in *b)
{
    ocp_qp_dims *d = a->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const int nv = d->nu[k] + d->nx[k], nx1 = k < d->N ? d->nx[k + 1] : 0, nct = 2 * (d->nb[k] + d->ng[k] + d->ns[k]);
        blasfeo_dveccp(nx1, a->b + k, 0, b->b + k, 0);
        blasfeo_dveccp(nv + 2 * d->ns[k], a->rqz + k, 0, b->rqz + k, 0);
        blasfeo_dveccp(nct, a->d + k, 0, b->d + k, 0);
        blasfeo_dveccp(nct, a->d_mask + k, 0, b->d_mask + k, 0);
        blasfeo_dveccp(nct, a->m + k, 0, b->m + k, 0);
    }
}
static void copy_matrices(ocp_qp_in *a, ocp_qp_in *b)
{
    ocp_qp_dims *d = a->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const int nv = d->nu[k] + d->nx[k], nx1 = k < d->N ? d->nx[k + 1] : 0, nb = d->nb[k], ng = d->ng[k];
        blasfeo_dgecp(nv + 1, nx1, a->BAbt + k, 0, 0, b->BAbt + k, 0, 0);
        blasfeo_dgecp(nv + 1, nv, a->RSQrq + k, 0, 0, b->RSQrq + k, 0, 0);
        blasfeo_dgecp(nv, ng, a->DCt + k, 0, 0, b->DCt + k, 0, 0);
        blasfeo_dveccp(2 * d->ns[k], a->Z + k, 0, b->Z + k, 0);
        memcpy(b->idxb[k], a->idxb[k], sizeof(int) * (size_t) nb);
        memcpy(b->idxs_rev[k], a->idxs_rev[k], sizeof(int) * (size_t) (nb + ng));
        memcpy(b->idxe[k], a->idxe[k], sizeof(int) * (size_t) (d->nbxe[k] + d->nbue[k] + d->nge[k]));
        b->diag_H_flag[k] = a->diag_H_flag[k];
    }
}
static int cx_condense_lhs(void *qp_in, void *xin, void *opts, void *mem, void *work) { copy_matrices(qp_in, xin); return ACADOS_SUCCESS; }
static int cx_condense_rhs(void *qp_in, void *xin, void *opts, void *mem, void *work) { copy_vectors(qp_in, xin); return ACADOS_SUCCESS; }
static int cx_condensing(void *qp_in, void *xin, void *opts, void *mem_, void *work)
{
    acados_timer t; acados_tic(&t);
    copy_matrices(qp_in, xin); copy_vectors(qp_in, xin);
    ((copy_xcond_memory *) mem_)->time_qp_xcond = acados_toc(&t);
    return ACADOS_SUCCESS;
}
static int cx_condense_rhs_seed(void *qp_in, void *seed_, void *xseed_, void *opts, void *mem, void *work)
{
    ocp_qp_seed *a = seed_, *b = xseed_;
    ocp_qp_dims *d = a->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const int nct = 2 * (d->nb[k] + d->ng[k] + d->ns[k]);
        blasfeo_dveccp(d->nu[k] + d->nx[k] + 2 * d->ns[k], a->seed_g + k, 0, b->seed_g + k, 0);
        blasfeo_dveccp(k < d->N ? d->nx[k + 1] : 0, a->seed_b + k, 0, b->seed_b + k, 0);
        blasfeo_dveccp(nct, a->seed_d + k, 0, b->seed_d + k, 0);
    }
    return ACADOS_SUCCESS;
}
static int cx_condense_qp_out(void *qp_in, void *xin, void *qp_out, void *xout, void *opts, void *mem, void *work)
{
    ocp_qp_out_copy(qp_out, xout);
    return ACADOS_SUCCESS;
}
static int cx_expansion(void *xout_, void *qp_out_, void *opts, void *mem, void *work)
{
    ocp_qp_out_copy(xout_, qp_out_);
    return ACADOS_SUCCESS;
}

void copy_xcond_config_initialize_default(void *config_)
{
    ocp_qp_xcond_config *c = (ocp_qp_xcond_config *) config_;
    c->dims_calculate_size = &cx_dims_calculate_size; c->dims_assign = &cx_dims_assign; c->dims_set = &cx_dims_set; c->dims_get = &cx_dims_get;
    c->opts_calculate_size = &cx_opts_calculate_size; c->opts_assign = &cx_opts_assign; c->opts_initialize_default = &cx_opts_initialize_default;
    c->opts_update = &cx_opts_update; c->opts_set = &cx_opts_set;
    c->memory_calculate_size = &cx_memory_calculate_size; c->memory_assign = &cx_memory_assign; c->memory_get = &cx_memory_get;
    c->workspace_calculate_size = &cx_workspace_calculate_size;
    c->condensing = &cx_condensing; c->condense_rhs = &cx_condense_rhs; c->condense_rhs_seed = &cx_condense_rhs_seed;
    c->condense_lhs = &cx_condense_lhs; c->condense_qp_out = &cx_condense_qp_out; c->expansion = &cx_expansion; c->expand_sol_seed = &cx_expansion;
}
