#!/bin/bash
# Round 4, second session (the GA3C-CADRL forward on the 16-bit matrix cores, cagym_ga3c_act in one launch): what changed under
# profiles/r4 - GPU tests, smoke, the headline lines again (their kernels are untouched), the cfg4 line with its kernel stats, the
# forward kernel's phase stamps and its A/B variants.  The diagnostic / A/B libraries are built in the development container:
#   build.build_variant('gastamps', ['-DCAGYM_STAMPS', '-DGA_STAMPS']);  build.py --alt scaled cagym_api -DGA16_SCALED;  build.py --alt nohoist cagym_api -DGA16_NO_HOIST
set -e
O=gpurun_out/final_r4b
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/gputests_full.log 2>&1 || { tail -30 $O/gputests_full.log; exit 1; }
grep -v "^$" $O/gputests_full.log | grep "passed\|failed\|max |p\|laserscan:" > $O/gputests.log; cat $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
echo headline done
python bench.py --config cfg4 --steps 200 --warmup 50 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o run -- python3 bench.py --config cfg4 --steps 200 --warmup 50 --repeats 5 --no-cpu-baseline > $O/bench_cfg4_under_rocprof.json 2>/dev/null
cp $(find $O/stats_cfg4 -name "*kernel_stats.csv" | head -1) $O/bench_cfg4_kernel_stats.csv
echo cfg4 done
python tools/ga3c_phases.py 2>&1 | grep -v amdgpu.ids > $O/ga3c_forward_phases_h16.txt
{ echo "cagym_ga3c_act, HIP events around 50 calls, median of 7 (tools/ga3c_time.py): the shipped library and its A/B builds interleaved on one box";
  echo "  shipped = split-f16 operands, low halves unscaled into ONE accumulator, first tile's weights requested at kernel start";
  echo "  scaled  = -DGA16_SCALED: low halves x 2^11 into a second accumulator;  nohoist = -DGA16_NO_HOIST: weights requested behind the state rows;";
  echo "  mfma32  = CAGYM_GA3C=mfma32: round 3's three-launch chain with the exact-fp32 matrix-core kernel";
  for rep in 1 2 3; do
    python tools/ga3c_time.py shipped 2>/dev/null | grep _us
    CAGYM_LIB=$PWD/gym-exploration-2d_amd/csrc/libcagym_hip_scaled.so python tools/ga3c_time.py scaled 2>/dev/null | grep _us
    CAGYM_LIB=$PWD/gym-exploration-2d_amd/csrc/libcagym_hip_nohoist.so python tools/ga3c_time.py nohoist 2>/dev/null | grep _us
    CAGYM_GA3C=mfma32 python tools/ga3c_time.py mfma32 2>/dev/null | grep _us
  done
  echo "accuracy against the fp64 restatement (tests/test_ga3c.py -k 'forward_kernels_agree or wide_ranges'):"
  for tag in shipped scaled; do
    if [ $tag = shipped ]; then L=; else L=$PWD/gym-exploration-2d_amd/csrc/libcagym_hip_$tag.so; fi
    CAGYM_LIB=$L python -m pytest tests/test_ga3c.py -x -q -m gpu -s -k "forward_kernels_agree or wide_ranges" 2>&1 | grep "max |p" | sed "s/^/  $tag: /"
  done; } > $O/ga3c16_ab.txt
cat $O/ga3c16_ab.txt
echo ga3c done
find $O -name "*.csv" -size +2M -delete 2>/dev/null || true
