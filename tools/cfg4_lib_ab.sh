#!/bin/bash
# same-box A/B of the cfg4 step (bench.py --config cfg4, the policy in front of a cold env step): shipped library vs csrc/libcagym_hip_<tag>.so, interleaved.
# Usage: tools/cfg4_lib_ab.sh <out file> <tag>...
out=$1; shift
: > "$out"
for rep in 1 2 3; do
  python bench.py --config cfg4 --steps 200 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('shipped ms_per_step %.5f ga3c_ms %.5f env_ms %.5f' % (d['ms_per_step'], d['cfg4']['ga3c_ms_per_step'], d['cfg4']['env_kernel_ms_per_step']))" >> "$out" || exit 1
  for tag in "$@"; do
    CAGYM_LIB=$PWD/gym-exploration-2d_amd/csrc/libcagym_hip_$tag.so python bench.py --config cfg4 --steps 200 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$tag ms_per_step %.5f ga3c_ms %.5f env_ms %.5f' % (d['ms_per_step'], d['cfg4']['ga3c_ms_per_step'], d['cfg4']['env_kernel_ms_per_step']))" >> "$out" || exit 1
  done
done
cat "$out"
