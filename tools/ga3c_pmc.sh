#!/bin/bash
# SQ counters of cfg4's policy kernel (k_ga3c_act_h16): matrix-core and vector utilisation (rocprofv3 --pmc passes of the cfg4 bench, counters only)
O=${1:-gpurun_out/ga3c_pmc}
mkdir -p $O
export TMPDIR=/tmp
CMD="python3 bench.py --config cfg4 --steps 200 --warmup 50 --repeats 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/p1 -o run -- $CMD > /dev/null 2>&1
python tools/pmc_summary.py $O/p1 k_ga3c_act > $O/p1.txt
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $O/p3 -o run -- $CMD > $O/p3.log 2>&1 && python tools/pmc_summary.py $O/p3 k_ga3c_act > $O/p3.txt
cat $O/p1.txt $O/p3.txt 2>/dev/null
tail -3 $O/p3.log
