#!/bin/bash
# Development tool: the through-the-boundary call (tools/orchestration_latency.py, default path and zero-copy at every size) against the number
# of host threads:     bash tools/orch_threads_sweep.sh [n]         (on the GPU box)
n=${1:-4096}
for t in 2 4 8 16 32; do
  for zc in "" 1; do
    echo "threads $t ${zc:+ACADOS_AMD_ZERO_COPY=1}"
    ORCH_THREADS=$t ACADOS_AMD_ZERO_COPY=$zc python tools/orchestration_latency.py $n 2>&1 | head -1 | grep -o "ms_per_call [0-9.]*\|unpack_in_ms [0-9.]*\|copy_and_device_ms [0-9.]*\|rti_feedback_ms [0-9.]*\|fb_unpack_in_ms [0-9.]*" | tr '\n' ' '; echo
  done
done
