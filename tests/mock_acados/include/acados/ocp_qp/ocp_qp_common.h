/* TEST INFRASTRUCTURE ONLY -- the part of acados/ocp_qp/ocp_qp_common.h an inner QP plugin compiles against
 * (:49-54 typedefs, :60-79 qp_solver_config, :114-122 qp_info), restated */
#ifndef MOCK_ACADOS_OCP_QP_COMMON_H_
#define MOCK_ACADOS_OCP_QP_COMMON_H_

#include "acados/utils/types.h"
#include "hpipm_d_ocp_qp.h"

typedef struct d_ocp_qp_dim ocp_qp_dims;
typedef struct d_ocp_qp ocp_qp_in;
typedef struct d_ocp_qp_sol ocp_qp_out;
typedef struct d_ocp_qp_seed ocp_qp_seed;

#ifndef QP_SOLVER_CONFIG_
#define QP_SOLVER_CONFIG_
typedef struct
{
    void (*dims_set)(void *config_, void *dims_, int stage, const char *field, int *value);
    acados_size_t (*opts_calculate_size)(void *config, void *dims);
    void *(*opts_assign)(void *config, void *dims, void *raw_memory);
    void (*opts_initialize_default)(void *config, void *dims, void *opts);
    void (*opts_update)(void *config, void *dims, void *opts);
    void (*opts_set)(void *config_, void *opts_, const char *field, void *value);
    void (*opts_get)(void *config_, void *opts_, const char *field, void *value);
    acados_size_t (*memory_calculate_size)(void *config, void *dims, void *opts);
    void *(*memory_assign)(void *config, void *dims, void *opts, void *raw_memory);
    void (*memory_get)(void *config_, void *mem_, const char *field, void *value);
    acados_size_t (*workspace_calculate_size)(void *config, void *dims, void *opts);
    int (*evaluate)(void *config, void *qp_in, void *qp_out, void *opts, void *mem, void *work);
    void (*solver_get)(void *config_, void *qp_in_, void *qp_out_, void *opts_, void *mem_, const char *field, int stage, void *value,
                       int size1, int size2);
    void (*memory_reset)(void *config, void *qp_in, void *qp_out, void *opts, void *mem, void *work);
    void (*eval_forw_sens)(void *config, void *qp_in, void *seed, void *qp_out, void *opts, void *mem, void *work);
    void (*eval_adj_sens)(void *config, void *qp_in, void *seed, void *qp_out, void *opts, void *mem, void *work);
    void (*terminate)(void *config, void *mem, void *work);
} qp_solver_config;
#endif

#ifndef QP_INFO_
#define QP_INFO_
typedef struct
{
    double solve_QP_time;
    double condensing_time;
    double interface_time;
    double total_time;
    int num_iter;
    int t_computed;
} qp_info;
#endif

#endif
