# usage: bash tools/profile_mfma_c4_c5.sh <commit> [tag]   -> gpurun_out/<tag>_mfma_util_c4_c5.json (copy into profiles/)
tag=${2:-rXX}
root=$(pwd); out=$root/gpurun_out; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/mg1 /tmp/mg2
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU -d /tmp/mg1 -- python $root/tools/mfma_once.py > $out/${tag}_mfma2_pmc.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/mg2 -- python $root/tools/mfma_once.py > $out/${tag}_mfma2_trace.log 2>&1
p=$(find /tmp/mg1 -name "*.db" | head -1); t=$(find /tmp/mg2 -name "*.db" | head -1)
python $root/profiles/summarize.py mfma $p $t $1 > $out/${tag}_mfma_util_c4_c5.json; cat $out/${tag}_mfma_util_c4_c5.json | grep -E "kt_factor|utilisation|TFLOPs|avg_us"
