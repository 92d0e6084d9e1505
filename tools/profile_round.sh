#!/bin/bash
# Everything the bench line's roofline object refers to, for the commit it is run on; run on the GPU box:
#     gpurun -- 'bash tools/profile_round.sh r02_v1 <commit> [full]'
# writes gpurun_out/<tag>_*  (copy into profiles/; bench.py reads the newest profiles/*_pmc_traffic.json):
#   <tag>_kernel_stats.txt     rocprofv3 --kernel-trace --stats summary of the headline bench (C2)
#   <tag>_pmc_traffic.json     HBM bytes per launch of every sweep kernel: --pmc FETCH_SIZE / --pmc WRITE_SIZE, SEPARATE
#                              counter-only passes (no trace domains), gfx950 correction of MI355X_MICROARCH.md
#   <tag>_bench.json           the full result object of the same build (reads the traffic file just written if copied first)
#   <tag>_bench_line.json      the compact line the driver parses (stdout's last line)
# `full` also traces the configuration legs (C3 / C4 / C5) of the bench and collects their HBM traffic per configuration
#   <tag>_config_kernel_stats.txt, <tag>_config_pmc_traffic.json (sections cut by the marker launches of bench.py),
#   <tag>_mfma_util.json (tools/profile_mfma.sh: MFMA instructions / busy cycles of km_pcond and kt_factor over one C3 solve)
tag=${1:-rXX}
commit=${2:-unknown}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c2 /tmp/prof_cfg /tmp/pmc_f /tmp/pmc_w
quick="--steps 1 --warmup 0 --no-cpu-baseline --no-configs --check 0"
rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof_c2.log
db=$(ls -S $(find /tmp/prof_c2 -name "*.db") | head -1)
python $root/profiles/summarize.py kernel $db > $out/${tag}_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -- python $root/bench.py $quick > /dev/null 2> $out/${tag}_pmc_f.log
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -- python $root/bench.py $quick > /dev/null 2> $out/${tag}_pmc_w.log
f=$(find /tmp/pmc_f -name "*.db" | head -1); w=$(find /tmp/pmc_w -name "*.db" | head -1)
python $root/profiles/summarize.py traffic $f $w $commit > $out/${tag}_pmc_traffic.json
cp $out/${tag}_pmc_traffic.json $root/profiles/${tag}_pmc_traffic.json   # the bench below reads the file of THIS build
if [ "$3" = "full" ]; then
    rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --check 0 --check-configs 0 > $out/${tag}_bench_configs_under_rocprof.json 2> $out/${tag}_rocprof_cfg.log
    # (the boundary leg of the bench runs a child process with a trace database of its own: the bench's is the largest)
    db=$(ls -S $(find /tmp/prof_cfg -name "*.db") | head -1)
    python $root/profiles/summarize.py kernel $db > $out/${tag}_config_kernel_stats.txt
    # HBM traffic of the configuration legs: the same two counter-only passes over the bench WITH its configuration legs;
    # bench.py brackets every configuration's timed solves with marker launches, summarize.py cuts the sequence there
    cfgq="--steps 1 --warmup 0 --no-cpu-baseline --check 0 --check-configs 0"
    rm -rf /tmp/pmc_cf /tmp/pmc_cw
    rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_cf -- python $root/bench.py $cfgq > /dev/null 2> $out/${tag}_pmc_cf.log
    rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_cw -- python $root/bench.py $cfgq > /dev/null 2> $out/${tag}_pmc_cw.log
    f=$(find /tmp/pmc_cf -name "*.db" | head -1); w=$(find /tmp/pmc_cw -name "*.db" | head -1)
    python $root/profiles/summarize.py sections $f $w $commit > $out/${tag}_config_pmc_traffic.json
    cp $out/${tag}_config_pmc_traffic.json $root/profiles/${tag}_config_pmc_traffic.json
    # matrix-pipe utilisation of the kernels that issue MFMAs (km_pcond, kt_factor) over one C3 solve
    (cd $root && bash tools/profile_mfma.sh $tag $commit > /dev/null 2>&1)
    cp $out/${tag}_mfma_util.json $root/profiles/${tag}_mfma_util.json
fi
# the full object goes to <tag>_bench.json (what the tests and bench.py's readers load), the line the driver parses to <tag>_bench_line.json
cd $root && python bench.py --detail-file $out/${tag}_bench.json 2> $out/${tag}_bench_err.log | tail -1 > $out/${tag}_bench_line.json
