#!/bin/bash
# headline bench for several alternative libraries: tools/abn.sh lib1.so lib2.so ...
mkdir -p gpurun_out/exp
for L in "$@"; do
  T=$(basename $L .so)
  CAGYM_LIB=$L python bench.py --no-cpu-baseline > gpurun_out/exp/$T.json || exit 1
  CAGYM_LIB=$L python bench.py --no-cpu-baseline --worlds 65536 --steps 512 --warmup 128 --pool-factor 2 > gpurun_out/exp/${T}_64k.json || exit 1
  python - <<PY
import json
for f in ("$T", "${T}_64k"):
    d = json.load(open("gpurun_out/exp/%s.json" % f))
    print(f, "%.1f M env-steps/s" % (d["value"] / 1e6), "launch %.3f ms" % d["roofline"]["launch_ms"], "frac %.4f" % d["roofline"]["frac"])
PY
done
