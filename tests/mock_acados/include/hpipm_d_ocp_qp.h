/*
 * TEST INFRASTRUCTURE ONLY -- the HPIPM structs behind acados' ocp_qp_dims / ocp_qp_in / ocp_qp_out
 * (acados/ocp_qp/ocp_qp_common.h:49-51), restated from the fields acados touches:
 * print.c:197-209, 220-429; ocp_qp_common.c:166-169, 874-921; ocp_qp_clarabel.c:205-683.  HPIPM itself is an empty
 * submodule in /root/reference.
 */
#ifndef MOCK_HPIPM_D_OCP_QP_H_
#define MOCK_HPIPM_D_OCP_QP_H_

#include "blasfeo_d_aux.h"

struct d_ocp_qp_dim
{
    int *nx, *nu, *nb, *nbx, *nbu, *ng, *ns, *nbxe, *nbue, *nge;
    int N;
};

struct d_ocp_qp
{
    struct d_ocp_qp_dim *dim;
    struct blasfeo_dmat *BAbt;   /* [B'; A'; b'], (nu+nx+1) x nx+ ; LAST ROW NOT AUTHORITATIVE (ocp_nlp writes b only to the vector) */
    struct blasfeo_dmat *RSQrq;  /* lower triangle of [[R,S],[S',Q]] + row [r' q'], (nu+nx+1) x (nu+nx); last row not authoritative */
    struct blasfeo_dmat *DCt;    /* [D'; C'], (nu+nx) x ng */
    struct blasfeo_dvec *b;      /* nx+ */
    struct blasfeo_dvec *rqz;    /* [r; q; zl; zu] */
    struct blasfeo_dvec *d;      /* [lb; lg; -ub; -ug; ls; us] */
    struct blasfeo_dvec *d_mask; /* same shape, 1.0 / 0.0 */
    struct blasfeo_dvec *m;      /* complementarity rhs */
    struct blasfeo_dvec *Z;      /* [Zl; Zu] */
    int **idxb;                  /* nb, into [u; x] */
    int **idxs_rev;              /* nb+ng, slack index or -1 */
    int **idxe;                  /* positions (in the bound list) of equality-flagged rows */
    int *diag_H_flag;
};

struct d_ocp_qp_sol
{
    struct d_ocp_qp_dim *dim;
    struct blasfeo_dvec *ux;     /* [u; x; sl; su] */
    struct blasfeo_dvec *pi;
    struct blasfeo_dvec *lam;    /* [lb lg ub ug ls us], >= 0 */
    struct blasfeo_dvec *t;
    void *misc;                  /* qp_info */
};

/* seed of a sensitivity solve (hpipm_d_ocp_qp_seed.h): fields acados writes, ocp_nlp_common.c:4057-4081, 4133 */
struct d_ocp_qp_seed
{
    struct d_ocp_qp_dim *dim;
    struct blasfeo_dvec *seed_g; /* like rqz */
    struct blasfeo_dvec *seed_b; /* like b */
    struct blasfeo_dvec *seed_d; /* like d: [lb; lg; -ub; -ug; ls; us] */
    struct blasfeo_dvec *seed_m;
};

#endif
