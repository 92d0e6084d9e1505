"""ocp_qp_gpu_ipm_acados_evaluate_batch on acados structs (integration/ocp_qp_gpu_ipm.c compiled against the stand-in
HPIPM / BLASFEO declarations of tests/mock_acados): ms per call for n C2-shaped QPs under several host-thread settings
-- OpenMP workers that spin after a parallel region compete with the thread that drives the device loop.
usage (GPU box): python tools/adapter_batch_latency.py [n ...]"""
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import subprocess
from acados_amd import _lib
from acados_amd.generators import lqr_instance_qp, random_lqr_batch
from test_mock_acados import _build, _write_qp
os.environ.pop("ACADOS_AMD_WPI_BATCH_MAX", None)   # tests/conftest.py pins the small-batch dispatch off for the CPU tier

tmp = pathlib.Path(tempfile.mkdtemp())
exe = _build(_lib.LIB_PATH, tmp)
qp = lqr_instance_qp(random_lqr_batch(N=50, batch=1, seed=5), 0, 50)
f = str(tmp / "qp.txt")
_write_qp(qp, f)
for n in [int(a) for a in sys.argv[1:]] or [1024]:
    for env_add in ({"OMP_NUM_THREADS": "16"}, {"OMP_NUM_THREADS": "16", "OMP_WAIT_POLICY": "passive"},
                    {"OMP_NUM_THREADS": "8", "OMP_WAIT_POLICY": "passive"}, {"OMP_NUM_THREADS": "4", "OMP_WAIT_POLICY": "passive"},
                    {"OMP_NUM_THREADS": "1"}):
        r = subprocess.run([exe, "batch", str(n), f, "-", str(tmp / "o.bin"), "5"], capture_output=True, text=True, env=dict(os.environ, **env_add))
        print(n, env_add, r.stdout.splitlines()[0] if r.stdout else r.stderr[-300:], r.stderr.strip().splitlines()[-1:], flush=True)
