#!/bin/bash
# SQ counters of cfg4's env kernel (k_step3, OBST instantiation): what bounds it (VALU issue, LDS, waiting)
O=${1:-gpurun_out/cfg4_pmc}
mkdir -p $O
export TMPDIR=/tmp
CMD="python3 bench.py --config cfg4 --steps 200 --warmup 50 --repeats 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/p1 -o run -- $CMD > /dev/null 2>&1
python tools/pmc_summary.py $O/p1 k_step3 > $O/p1.txt
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/p2 -o run -- $CMD > $O/p2.log 2>&1 && python tools/pmc_summary.py $O/p2 k_step3 > $O/p2.txt
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -d $O/p3 -o run -- $CMD > $O/p3.log 2>&1 && python tools/pmc_summary.py $O/p3 k_step3 > $O/p3.txt
cat $O/p1.txt $O/p2.txt $O/p3.txt 2>/dev/null
