#!/bin/bash
# same-box timing A/B of cagym_ga3c_act: shipped library vs csrc/libcagym_hip_<tag>.so, interleaved three times.  Usage: tools/ga3c16_time_ab.sh <out file> <tag>...
out=$1; shift
: > "$out"
for rep in 1 2 3; do
  timeout -k 10 120 python tools/ga3c_time.py shipped 2>/dev/null | grep _us >> "$out" || exit 1
  for tag in "$@"; do
    CAGYM_LIB=$PWD/gym-exploration-2d_amd/csrc/libcagym_hip_$tag.so timeout -k 10 120 python tools/ga3c_time.py $tag 2>/dev/null | grep _us >> "$out" || exit 1
  done
done
cat "$out"
