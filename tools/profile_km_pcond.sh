cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/p1 -- python $root/tools/c3_once.py 65536 both > $root/gpurun_out/km_pmc1.log 2>&1
db=$(find /tmp/p1 -name "*.db" | head -1); python $root/profiles/summarize.py pmc $db | grep -E "pcond|kernel " > $root/gpurun_out/r04_km_pcond_pmc.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM -d /tmp/p2 -- python $root/tools/c3_once.py 65536 both > $root/gpurun_out/km_pmc2.log 2>&1
db=$(find /tmp/p2 -name "*.db" | head -1); python $root/profiles/summarize.py pmc $db | grep -E "pcond" >> $root/gpurun_out/r04_km_pcond_pmc.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/p3 -- python $root/tools/c3_once.py 65536 both > $root/gpurun_out/km_pmc3.log 2>&1
db=$(find /tmp/p3 -name "*.db" | head -1); python $root/profiles/summarize.py pmc $db | grep -E "pcond" >> $root/gpurun_out/r04_km_pcond_pmc.txt
rm -rf /tmp/p4; rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d /tmp/p4 -- python $root/tools/c3_once.py 65536 both > $root/gpurun_out/km_pmc4.log 2>&1
db=$(find /tmp/p4 -name "*.db" | head -1); python $root/profiles/summarize.py pmc $db | grep -E "pcond" >> $root/gpurun_out/r04_km_pcond_pmc.txt
rm -rf /tmp/p5; rocprofv3 --kernel-trace --stats -d /tmp/p5 -- python $root/tools/c3_once.py 65536 both > $root/gpurun_out/km_trace.log 2>&1
db=$(find /tmp/p5 -name "*.db" | head -1); python $root/profiles/summarize.py kernel $db | grep -E "pcond|pexpand|kernel " >> $root/gpurun_out/r04_km_pcond_pmc.txt
cat $root/gpurun_out/r04_km_pcond_pmc.txt; tail -3 $root/gpurun_out/km_pmc2.log
