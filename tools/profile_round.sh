#!/bin/bash
# rocprofv3 kernel-trace summary of the headline bench (C2) and the rates of the other configurations; run on the
# GPU box:   gpurun -- 'bash tools/profile_round.sh r01_v5 [full]'
# writes gpurun_out/<tag>_*.txt (copy what should be kept into profiles/).  `full` also traces the configuration run.
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c2 /tmp/prof_cfg
rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof_c2.log
db=$(find /tmp/prof_c2 -name "*.db" | head -1)
python $root/profiles/summarize.py kernel $db > $out/${tag}_kernel_stats.txt
if [ "$2" = "full" ]; then
    rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg -- python $root/tools/config_rates.py 4096 > $out/${tag}_config_rates.txt 2> $out/${tag}_rocprof_cfg.log
    db=$(find /tmp/prof_cfg -name "*.db" | head -1)
    python $root/profiles/summarize.py kernel $db > $out/${tag}_config_kernel_stats.txt
else
    cd $root && python tools/config_rates.py 4096 > $out/${tag}_config_rates.txt 2>&1
fi
cd $root && python bench.py 2>/dev/null | tail -1 > $out/${tag}_bench.json
