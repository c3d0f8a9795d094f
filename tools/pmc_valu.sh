#!/bin/bash
# VALU/SALU instruction counts and busy cycles of the rollout kernel: tools/pmc_valu.sh <tag> [worlds]
set -e
T=$1; N=${2:-4096}
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS"
rocprofv3 --kernel-trace --output-format csv --pmc $PMC -d gpurun_out/pmc/$T -o run -- python3 bench.py --steps 128 --warmup 64 --no-cpu-baseline --worlds $N --pool-factor 2 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc/$T k_rollout2 > gpurun_out/pmc/$T.txt
cat gpurun_out/pmc/$T.txt
