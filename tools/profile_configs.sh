#!/bin/bash
# kernel times + HBM traffic (PMC, separate passes) of the non-headline configurations (tools/config_rates.py);
# run on the GPU box:  gpurun -- 'bash tools/profile_configs.sh r01_v7'
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cfg_k /tmp/cfg_f /tmp/cfg_w
rocprofv3 --kernel-trace --stats -d /tmp/cfg_k -- python $root/tools/config_rates.py 4096 > $out/${tag}_config_rates.txt 2> $out/${tag}_cfg_k.log
rocprofv3 --pmc FETCH_SIZE -d /tmp/cfg_f -- python $root/tools/config_rates.py 4096 > /dev/null 2> $out/${tag}_cfg_f.log
rocprofv3 --pmc WRITE_SIZE -d /tmp/cfg_w -- python $root/tools/config_rates.py 4096 > /dev/null 2> $out/${tag}_cfg_w.log
k=$(find /tmp/cfg_k -name "*.db" | head -1); f=$(find /tmp/cfg_f -name "*.db" | head -1); w=$(find /tmp/cfg_w -name "*.db" | head -1)
python $root/profiles/summarize.py kernel $k > $out/${tag}_config_kernel_stats.txt
python $root/profiles/summarize.py traffic $f $w > $out/${tag}_config_pmc_traffic.json
