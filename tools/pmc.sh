#!/bin/bash
# Three separate rocprofv3 --pmc passes (clock, SQ, TCC) over a python target; per-kernel averages by tools/pmc_summary.py
#   tools/pmc.sh <name> <kernel substring> <script.py> [args...]
name=$1; kern=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() {  # tag, counters...
  tag=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${name}_$tag -o $tag -- python3 $R/"${ARGS[0]}" "${ARGS[@]:1}" > $R/gpurun_out/pmc_${name}_$tag.log 2>&1 || exit 1
}
ARGS=("$@")
run clk GRBM_GUI_ACTIVE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
cd $R
python3 tools/pmc_summary.py "$kern" gpurun_out/pmc_${name}_clk gpurun_out/pmc_${name}_sq gpurun_out/pmc_${name}_sq2 gpurun_out/pmc_${name}_tcc | tee gpurun_out/pmc_${name}_summary.txt
find gpurun_out/pmc_${name}_* -name "*counter_collection.csv" -size +3M -delete
