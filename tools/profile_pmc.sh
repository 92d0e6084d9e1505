#!/bin/bash
# HBM traffic of the sweep kernels from PMC counters (separate passes, counters only -- no trace domains), run on
# the GPU box:  gpurun -- 'bash tools/profile_pmc.sh r01_v3'
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -- python $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/${tag}_pmc_f.log
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -- python $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/${tag}_pmc_w.log
f=$(find /tmp/pmc_f -name "*.db" | head -1); w=$(find /tmp/pmc_w -name "*.db" | head -1)
python $root/profiles/summarize.py pmc $f > $out/${tag}_pmc_fetch.txt
python $root/profiles/summarize.py pmc $w > $out/${tag}_pmc_write.txt
python $root/profiles/summarize.py traffic $f $w > $out/${tag}_pmc_traffic.json
