#!/usr/bin/env python3
"""A PATCHED COPY of the reference files integration/acados.patch touches (plus the three plugin files dropped into acados/ocp_qp/),
for the builds that compile the patched reference sources (integration/Makefile: _ref_build/acados_c_driver;
tests/test_patched_acados_c_layer.py).  Nothing of it is stored in this repository.

    python integration/patched_copy.py <reference tree> <destination directory>
"""
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def patched_copy(ref, dest):
    patch = os.path.join(HERE, "acados.patch")
    files = [m.group(1) for m in re.finditer(r"^--- a/(\S+)$", open(patch).read(), flags=re.M)]
    if os.path.isdir(dest):
        shutil.rmtree(dest)
    for f in files:
        src = os.path.join(ref, f)
        if os.path.exists(src):            # (the header the patch ADDS has no original)
            os.makedirs(os.path.dirname(os.path.join(dest, f)), exist_ok=True)
            shutil.copy(src, os.path.join(dest, f))
    subprocess.check_call(["patch", "-p1", "-s", "-i", patch], cwd=dest)
    for f in ("ocp_qp_gpu_ipm.c", "ocp_qp_gpu_pcond.c", "ocp_qp_gpu_segments.h"):
        shutil.copy(os.path.join(HERE, f), os.path.join(dest, "acados", "ocp_qp", f))
    return dest


if __name__ == "__main__":
    patched_copy(sys.argv[1], sys.argv[2])
