#!/bin/bash
# rocprofv3 passes of the cfg5 whole step (bench.py --config cfg5): kernel trace + stats, SQ counters, L2 counters (separate --pmc passes)
O=${1:-gpurun_out/cfg5_prof}
mkdir -p $O
export TMPDIR=/tmp
CMD="python3 bench.py --config cfg5 --steps 50 --warmup 10 --repeats 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $CMD > $O/bench_under_rocprof.json 2>/dev/null || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $O/pmc_sq -o run -- $CMD > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d $O/pmc_tcc -o run -- $CMD > /dev/null 2>&1 || exit 1
for k in k_dmcts_plan k_ig_rollouts k_ig_visible k_ig_update; do
  echo "== $k"; python tools/pmc_summary.py $O/pmc_sq $k; python tools/pmc_summary.py $O/pmc_tcc $k
done > $O/pmc_summary.txt
head -12 $O/stats/*kernel_stats.csv > $O/kernel_stats_head.csv 2>/dev/null
cat $O/pmc_summary.txt
