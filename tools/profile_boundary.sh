#!/bin/bash
# Development tool: per-kernel time of the through-the-boundary batch call (integration/_ref_build/ref_xcond_driver: n C3-shaped capsules,
# reference 22-slot solver objects, 5 one-call + 5 preparation / feedback pairs + 2 per-capsule solves) against its wall time:
#     bash tools/profile_boundary.sh [n ...]        (on the GPU box; summary on stdout)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
python - <<PY
import os, sys
sys.path.insert(0, "$R"); sys.path.insert(0, os.path.join("$R", "tests"))
from acados_amd.generators import lqr_instance_qp, random_lqr_batch
from test_mock_acados import _write_qp
_write_qp(lqr_instance_qp(random_lqr_batch(N=50, batch=1, seed=5), 0, 50), "/tmp/qp.txt")
PY
for n in ${@:-256 1024 4096}; do
  rm -rf /tmp/prof_$n
  OMP_NUM_THREADS=16 rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -- $R/integration/_ref_build/ref_xcond_driver batch $n /tmp/qp.txt /tmp/b.bin --cond-N 10 5 > /tmp/run_$n.txt 2> /tmp/err_$n.txt
  echo "== n $n"; grep "^batch n" /tmp/run_$n.txt | cut -c1-60; grep -o "device_solve_ms [0-9.]*\|rti_feedback_ms [0-9.]*" /tmp/run_$n.txt
  db=$(find /tmp/prof_$n -name "*.db" | head -1)
  python $R/profiles/summarize.py kernel $db | head -22
  echo "-- the launches of one call (the 3rd one-call evaluate)"; python $R/profiles/summarize.py timeline $db 2
done
