"""n C3-shaped capsules (N = 50, nx = 8, nu = 3, cond_N = 10), each holding the reference's 22-slot solver object around the two plugin slots,
through the fused batch route (ocp_qp_gpu_xcond_solver_acados_evaluate_batch): ms per call, best of 5 (integration/_ref_build/ref_xcond_driver,
built by integration/Makefile where the reference tree exists):   python tools/orchestration_latency.py [n ...]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from acados_amd.generators import lqr_instance_qp, random_lqr_batch
from test_mock_acados import _write_qp
exe = os.path.join(ROOT, "integration", "_ref_build", "ref_xcond_driver")
d = tempfile.mkdtemp()
f = os.path.join(d, "qp.txt")
_write_qp(lqr_instance_qp(random_lqr_batch(N=50, batch=1, seed=5), 0, 50), f)
env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("ORCH_THREADS", "16"))
env.pop("ACADOS_AMD_WPI_BATCH_MAX", None)
for n in [int(a) for a in sys.argv[1:]] or [1024, 4096]:
    for label, extra in (("default (zero-copy gather by size)", {}), ("zero-copy gather at every size (ACADOS_AMD_ZERO_COPY=1)", {"ACADOS_AMD_ZERO_COPY": "1"}),
                         ("panel runs on the host (ACADOS_AMD_ZERO_COPY=0)", {"ACADOS_AMD_ZERO_COPY": "0"}), ("blasfeo_unpack_* per block (ACADOS_AMD_LA_API=1)", {"ACADOS_AMD_LA_API": "1"})):
        r = subprocess.run([exe, "batch", str(n), f, os.path.join(d, "b.bin"), "--cond-N", "10", "5"], capture_output=True, text=True, env=dict(env, **extra))
        print(label, "|", r.stdout.splitlines()[0] if r.stdout else r.stderr[-300:], flush=True)
