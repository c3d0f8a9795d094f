#!/bin/bash
# GPU-box job that regenerates everything under profiles/r4 (run once the kernels are final): GPU tests, bench lines, rocprofv3
# kernel-trace stats of the bench commands, PMC passes (FETCH_SIZE / WRITE_SIZE in their own passes, SQ counters), sweeps,
# cfg4 / cfg5 lines with their counters and timelines, the 2-rank rehearsal, launch-cost trace, smoke.
# The diagnostic libraries (wgtrace, dmstamps) are built in the development container (build.build_variant) and travel.
# Split in parts so that one gpurun call stays well inside its limit:  tools/final_profile_r4.sh [a|b|c]
set -e
O=gpurun_out/final_r4
mkdir -p $O
export TMPDIR=/tmp
PART=${1:-a}
if [ $PART = a ]; then
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -3 $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 2048 --warmup 512 --repeats 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats20 -o run -- python3 bench.py --steps 20 --warmup 5 --repeats 200 --no-cpu-baseline > $O/bench20_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o run -- python3 bench.py --steps 1024 --warmup 512 --repeats 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o run -- python3 bench.py --steps 1024 --warmup 512 --repeats 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch20 -o run -- python3 bench.py --steps 20 --warmup 5 --repeats 200 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write20 -o run -- python3 bench.py --steps 20 --warmup 5 --repeats 200 --no-cpu-baseline > /dev/null 2>&1
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS"
rocprofv3 --kernel-trace --output-format csv --pmc $PMC -d $O/pmc_sq_4096 -o run -- python3 bench.py --steps 1024 --warmup 512 --repeats 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc $PMC -d $O/pmc_sq_20 -o run -- python3 bench.py --steps 20 --warmup 5 --repeats 200 --no-cpu-baseline > /dev/null 2>&1
for d in pmc_fetch pmc_write pmc_fetch20 pmc_write20 pmc_sq_4096 pmc_sq_20; do python tools/pmc_summary.py $O/$d k_rollout3 > $O/$d.txt; done
echo pmc done
fi
if [ $PART = b ]; then
: > $O/sweep.txt
for N in 16384 65536 262144; do
  R=64; if [ $N -ge 262144 ]; then R=16; fi
  echo "N=$N roll=$R" >> $O/sweep.txt
  python bench.py --no-cpu-baseline --worlds $N --roll $R --steps 512 --warmup 128 --repeats 5 --pool-factor 2 --scenarios device >> $O/sweep.txt
done
for R in 20 64 128 256; do
  echo "N=4096 roll=$R (steps per launch)" >> $O/sweep.txt
  python bench.py --no-cpu-baseline --roll $R --repeats 5 >> $O/sweep.txt
done
echo "per-step launches (cagym_step_autoreset)" >> $O/sweep.txt
python bench.py --no-cpu-baseline --per-step-launch --steps 1024 --warmup 128 --repeats 5 >> $O/sweep.txt
echo "cfg2 4096x4 NonCooperative" >> $O/sweep.txt
python bench.py --no-cpu-baseline --config cfg2 --repeats 5 >> $O/sweep.txt
echo "2048x20 RVO (maxNeighbors 20)" >> $O/sweep.txt
python bench.py --no-cpu-baseline --worlds 2048 --agents 20 --roll 256 --repeats 5 >> $O/sweep.txt
echo sweep done
CAGYM_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/rehearsal_2ranks_gloo_driver_cmd.json 2> $O/rehearsal_2ranks_gloo_driver_cmd.err
echo rehearsal done
python tools/launch_cost.py > $O/launch_cost.txt 2>&1
echo launch cost done
tools/valu_breakdown.sh $O/valu 2>&1 | grep -v amdgpu.ids > $O/valu_breakdown.txt
echo breakdown done
fi
if [ $PART = c ]; then
python bench.py --config cfg4 --steps 200 --warmup 50 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --config cfg4 --cfg4-step split --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_cfg4_split.json 2> $O/bench_cfg4_split.err
python bench.py --config cfg4 --cfg4-step overlap --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_cfg4_overlap.json 2> $O/bench_cfg4_overlap.err
python bench.py --config cfg5 --steps 200 --warmup 50 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o run -- python3 bench.py --config cfg4 --steps 200 --warmup 50 --repeats 5 --no-cpu-baseline > $O/bench_cfg4_under_rocprof.json 2>/dev/null
tools/cfg5_profile.sh $O/cfg5_prof > $O/cfg5_prof.log 2>&1
python tools/cfg4_timeline.py 2>&1 | grep -v amdgpu.ids > $O/cfg4_timeline.txt
tools/cfg4_pmc.sh $O/cfg4_pmc 2>&1 | grep -v amdgpu.ids > $O/cfg4_pmc.txt
python tools/dmcts_phases.py 2>&1 | grep -v amdgpu.ids > $O/dmcts_phases.txt
echo rows done
fi
# the raw rocprofv3 output directories stay on the box (only summaries travel back)
find $O -name "*.csv" -size +2M -delete 2>/dev/null || true
