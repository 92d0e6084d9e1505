"""Loader of the HIP shared library (C-ABI in include/acados_amd/*.h).

There is NO CPU fallback: if the gfx950 library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C acados_amd/csrc`)
importing the solver classes works but creating a solver raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libacados_amd_qp.so")
_LIB = None


def bind(L):
    """Attach argtypes/restypes of the batch C-ABI to a loaded library."""
    ip = C.POINTER(C.c_int)
    L.ocp_qp_gpu_batch_create.restype = C.c_void_p
    L.ocp_qp_gpu_batch_create.argtypes = [C.c_int, ip, ip, ip, ip, ip, ip, C.c_int, C.c_int]
    L.ocp_qp_gpu_batch_destroy.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_set_int.argtypes = [C.c_void_p, C.c_char_p, C.c_int, ip, C.c_int]
    L.ocp_qp_gpu_batch_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    L.ocp_qp_gpu_batch_opts_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    L.ocp_qp_gpu_batch_solve.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_condense_lhs.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_condense_rhs_and_solve.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_get.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    L.ocp_qp_gpu_batch_get_info.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    L.ocp_qp_gpu_batch_get_stat.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int]
    L.ocp_qp_gpu_batch_get_scalar.argtypes = [C.c_void_p, C.c_char_p]
    L.ocp_qp_gpu_batch_get_scalar.restype = C.c_double
    L.ocp_qp_gpu_batch_bytes.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_bytes.restype = C.c_size_t
    L.ocp_qp_gpu_batch_stream.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_stream.restype = C.c_void_p
    L.ocp_qp_gpu_batch_kernel_name.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_kernel_name.restype = C.c_char_p
    L.ocp_qp_gpu_batch_sens_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p]
    L.ocp_qp_gpu_batch_sens_solve.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_condense.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_condense.restype = C.c_void_p
    L.ocp_qp_gpu_batch_expand.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_get_dims.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
    L.ocp_qp_gpu_batch_get_int.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.ocp_qp_gpu_batch_condense_rhs.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_condense_rhs.restype = C.c_void_p
    L.ocp_qp_gpu_batch_condensed.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_condensed.restype = C.c_void_p
    L.ocp_qp_gpu_batch_set_bulk_out.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ocp_qp_gpu_batch_set_bulk_vec.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ocp_qp_gpu_batch_get_bulk_in.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ocp_qp_gpu_batch_set_bulk_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.ocp_qp_gpu_batch_set_bulk_staged.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_host_register.argtypes = [C.c_void_p, C.c_size_t]
    L.ocp_qp_gpu_host_unregister.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_gather_tables.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ocp_qp_gpu_batch_gather_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.ocp_qp_gpu_batch_condense_sol.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_comm_unique_id.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.ocp_qp_gpu_comm_create.restype = C.c_void_p
    L.ocp_qp_gpu_comm_destroy.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ocp_qp_gpu_batch_gather_root.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ocp_qp_gpu_batch_gather_v.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ocp_qp_gpu_comm_create_from_ops.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ocp_qp_gpu_comm_create_from_ops.restype = C.c_void_p
    L.ocp_qp_gpu_batch_bulk_len.argtypes = [C.c_void_p, C.c_int]
    L.ocp_qp_gpu_batch_bulk_offset.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.ocp_qp_gpu_batch_sens_bulk_len.argtypes = [C.c_void_p, C.c_int]
    L.ocp_qp_gpu_batch_sens_bulk_offset.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.ocp_qp_gpu_batch_sens_set_bulk.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ocp_qp_gpu_batch_sens_get_bulk.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ocp_qp_gpu_host_alloc.argtypes = [C.c_size_t]
    L.ocp_qp_gpu_host_alloc.restype = C.c_void_p
    L.ocp_qp_gpu_host_free.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_res_compute.argtypes = [C.c_void_p]
    L.ocp_qp_gpu_batch_res_nrm_inf.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    return L


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"acados_amd: HIP library {LIB_PATH} not built; run __graft_entry__.build() "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            # PyTorch ships its own copy of the HIP runtime; loading it first makes this library
            # and torch (device memory, streams, torch.distributed) share ONE runtime instance.
            import torch  # noqa: F401
        except ImportError:
            pass
        _LIB = bind(C.CDLL(LIB_PATH))
    return _LIB
