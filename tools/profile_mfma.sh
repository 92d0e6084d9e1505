#!/bin/bash
# Matrix-pipe utilisation of the kernels that issue MFMAs (km_pcond: partial condensing; kt_factor: Riccati factor sweep of the
# condensed QP) over one C3 solve; run on the GPU box:   gpurun -- 'bash tools/profile_mfma.sh r04_v2 <commit>'
# writes gpurun_out/<tag>_mfma_util.json (copy into profiles/: bench.py reads the newest profiles/*_mfma_util.json)
tag=${1:-rXX}; commit=${2:-unknown}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mf1 /tmp/mf2
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU -d /tmp/mf1 -- python $root/tools/c3_once.py 65536 1 > $out/${tag}_mfma_pmc.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/mf2 -- python $root/tools/c3_once.py 65536 1 > $out/${tag}_mfma_trace.log 2>&1
p=$(find /tmp/mf1 -name "*.db" | head -1); t=$(find /tmp/mf2 -name "*.db" | head -1)
python $root/profiles/summarize.py mfma $p $t $commit > $out/${tag}_mfma_util.json
cat $out/${tag}_mfma_util.json
