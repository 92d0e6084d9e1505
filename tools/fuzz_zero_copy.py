"""Development tool: the ZERO-COPY GATHER of the batch boundary against the blob path on random STRUCTURES (tests/random_qp.py: per-stage dims,
general rows, slacks -- shared ones too --, one-sided / masked rows, free and fixed x0).  Per seed: n capsules alternating between two random
structures, each an acados `ocp_qp_in` in panel-major BLASFEO storage (tests/mock_acados/driver.c against the restated headers), ONE call of
ocp_qp_gpu_ipm_acados_evaluate_batch -- once with the device gathering the QP data from the capsules' registered memory (the word tables of
integration/ocp_qp_gpu_ipm.c: zc_tables), once with the host threads filling the pinned blob (ACADOS_AMD_ZERO_COPY=0): the solutions, per-capsule
statuses and iteration counts must be the same byte for byte, and capsule 0 of each structure equal to the oracle's solution.

    python tools/fuzz_zero_copy.py hostsim 0 40          # CPU: host simulation (hipHostRegister is a no-op there: the tables are what is tested)
    python tools/fuzz_zero_copy.py gpu 0 200 [n]         # on the GPU box: real registration of malloc'ed capsule memory"""
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    tier, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    n = int(sys.argv[4]) if len(sys.argv) > 4 else (7 if tier == "hostsim" else 131)
    import test_mock_acados as T
    from random_qp import random_structure_qp
    from oracle.oracle import OracleQp, default_opts
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="fuzz_zc_", dir=os.path.join(ROOT, "gpurun_out")))
    if tier == "hostsim":
        from hostsim.build import build
        lib = build()
        lib = lib if isinstance(lib, str) else lib._name
    else:
        lib = os.path.join(ROOT, "acados_amd", "csrc", "libacados_amd_qp.so")
    exe = T._build(lib, tmp)
    fails, gathered = [], 0
    for seed in range(lo, hi):
        qa = random_structure_qp(seed, nx_max=(6, 8, 12)[seed % 3], nu_max=(3, 3, 4)[seed % 3])
        qb = random_structure_qp(seed + 100000, nx_max=(8, 12, 6)[seed % 3], nu_max=(3, 4, 3)[seed % 3])
        fa, fb = str(tmp / "qa.txt"), str(tmp / "qb.txt")
        T._write_qp(qa, fa); T._write_qp(qb, fb)
        try:
            iz, pz, _, rz = T._run_batch(exe, tmp, n, [fa, fb], sens=False, extra_env={"ACADOS_AMD_ZERO_COPY": "1"}, tag="zc")
            ib, pb, _, rb = T._run_batch(exe, tmp, n, [fa, fb], sens=False, extra_env={"ACADOS_AMD_ZERO_COPY": "0"}, tag="blob")
        except AssertionError as e:
            fails.append((seed, "driver failed: " + str(e)[-300:]))
            continue
        gathered += int(iz["zero_copy"] == 1)
        if iz["zero_copy"] != 1 or ib["zero_copy"] != 0:
            fails.append((seed, f"zero_copy flags {iz['zero_copy']} / {ib['zero_copy']}"))
        if pz != pb or rz.size != rb.size or not np.array_equal(rz, rb):
            fails.append((seed, f"gather and blob path differ: statuses equal {pz == pb}, max diff {float(np.max(np.abs(rz - rb))) if rz.size == rb.size else 'size'}"))
            continue
        # capsules 0 and 1 (one of each structure; the driver perturbs capsule i > 0: qp_loader.h mock_perturb) against the oracle
        p = 0
        for i in range(2):
            qp = T._perturbed(qb if i & 1 else qa, i)
            sol, used = T._split_bin(qp, rz[p:]); p += used
            o = OracleQp(qp)
            if o.solve(default_opts(tol_stat=1e-8, iter_max=80)) != 0 or pz[i][1] != 0:
                continue
            for k in range(qp.N + 1):
                ref = np.concatenate([o.get(k, "u"), o.get(k, "x"), o.get(k, "sl"), o.get(k, "su")])
                if not np.allclose(sol[("ux", k)], ref, rtol=3e-7, atol=3e-7):
                    fails.append((seed, f"capsule {i} stage {k}: {float(np.max(np.abs(sol[('ux', k)] - ref))):.2e} from the oracle"))
                    break
        if (seed - lo) % 20 == 19:
            print(f"seed {seed}: {len(fails)} failures, {gathered} gathered", flush=True)
    print(f"{hi - lo} seeds x {n} capsules of two structures: gathered zero-copy in {gathered}, {len(fails)} failures")
    for f in fails:
        print("  ", f)


if __name__ == "__main__":
    main()
