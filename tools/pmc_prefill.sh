#!/bin/bash
# PMC + kernel-trace passes of unified_attention on one prefill shape: tools/pmc_prefill.sh <BxQxCTX> <set> <outdir>
set -e
export SHAPE=$1 SET=$2 TMPDIR=/tmp
out=$3
mkdir -p $out
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/prof_prefill.py > $out/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc1 -- python3 tools/prof_prefill.py > $out/pmc1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc2 -- python3 tools/prof_prefill.py > $out/pmc2.log 2>&1
python3 tools/rocprof_summary.py $out/trace $out/pmc1 $out/pmc2 > $out/summary.txt 2>&1
