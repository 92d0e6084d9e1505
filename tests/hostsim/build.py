"""Builds tests/hostsim/libgqp_hostsim.so: the UNCHANGED kernel sources compiled with g++
against the host-simulation shim (tests/hostsim/include/hip/hip_runtime.h).
TEST INFRASTRUCTURE ONLY -- lets the CPU test tier exercise kernel and host logic against
the oracle; never loaded by the acados_amd package."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libgqp_hostsim.so")
SRCS = [os.path.join(ROOT, "acados_amd", "csrc", f) for f in ("gpu_batch.hip", "gpu_shapes_large.hip", "ocp_qp_host.cpp", "ocp_qp_xcond.cpp")]
DEPS = SRCS + [os.path.join(ROOT, "acados_amd", "csrc", f) for f in ("ipm_kernels.hpp", "ipm_kernels_box.hpp", "ipm_kernels_wpi.hpp", "ipm_kernels_w16.hpp", "ipm_kernels_w16r.hpp", "ipm_kernels_w16t.hpp", "mfma4.hpp", "pcond_kernels.hpp", "pcond_kernels_w16.hpp", "pcond_kernels_mfma.hpp", "ipm_kernels_wpi_mfma.hpp", "res_kernels.hpp", "ocp_qp_host_internal.h", "kernel_sets.h", "gpu_ipm_internal.h")] + \
    [os.path.join(HERE, "include", "hip", "hip_runtime.h"),
     os.path.join(ROOT, "include", "acados_amd", "ocp_qp_gpu_batch.h"),
     os.path.join(ROOT, "include", "acados_amd", "ocp_qp_interface.h")]


def build(force=False):
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in DEPS):
        return SO
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-Wno-unknown-pragmas",
           "-I" + os.path.join(HERE, "include"), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "acados_amd", "csrc")] + SRCS + ["-o", SO]
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True))
