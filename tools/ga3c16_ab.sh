#!/bin/bash
# A/B of the split-f16 GA3C forward: shipped library vs csrc/libcagym_hip_<tag>.so (python gym-exploration-2d_amd/build.py --alt <tag> cagym_api <flags>),
# timing (tools/ga3c_time.py) and accuracy against the fp64 restatement (the test prints it).  Usage: tools/ga3c16_ab.sh <out dir> <tag>...
out=$1; shift
mkdir -p "$out"
timeout -k 10 120 python tools/ga3c_time.py shipped > "$out/time_shipped.json" 2>&1 || exit 1
timeout -k 10 200 python -m pytest tests/test_ga3c.py -x -q -m gpu -s -k "forward_kernels_agree or wide_ranges" > "$out/acc_shipped.log" 2>&1
for tag in "$@"; do
  CAGYM_LIB=$PWD/gym-exploration-2d_amd/csrc/libcagym_hip_$tag.so timeout -k 10 120 python tools/ga3c_time.py $tag > "$out/time_$tag.json" 2>&1 || exit 1
  CAGYM_LIB=$PWD/gym-exploration-2d_amd/csrc/libcagym_hip_$tag.so timeout -k 10 200 python -m pytest tests/test_ga3c.py -x -q -m gpu -s -k "forward_kernels_agree or wide_ranges" > "$out/acc_$tag.log" 2>&1
done
grep -h "_us" "$out"/time_*.json
grep -H "max |p\|passed\|failed" "$out"/acc_*.log
