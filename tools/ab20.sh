#!/bin/bash
# same-box A/B of library builds on the driver's command (20-step launches) and the default bench, alternating twice
# usage: tools/ab20.sh alt1.so [alt2.so ...]   ("shipped" = the in-tree library)
mkdir -p gpurun_out/exp
for rep in 1 2; do
  for L in shipped "$@"; do
    if [ $L = shipped ]; then unset CAGYM_LIB; else export CAGYM_LIB=$L; fi
    python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/exp/a.json || exit 1
    python bench.py --no-cpu-baseline > gpurun_out/exp/b.json || exit 1
    python - <<PY
import json
a, b = json.load(open("gpurun_out/exp/a.json")), json.load(open("gpurun_out/exp/b.json"))
print("$L", "rep $rep", "20-step launch %.1f us by HIP events, %.1f us wall (%.1f M, frac %.4f)" % (a["roofline"]["launch_ms_hip_events"] * 1e3, a["roofline"]["launch_ms"] * 1e3, a["value"] / 1e6, a["roofline"]["frac"]),
      "| 512-step %.3f ms by HIP events, %.3f wall (%.1f M, frac %.4f)" % (b["roofline"]["launch_ms_hip_events"], b["roofline"]["launch_ms"], b["value"] / 1e6, b["roofline"]["frac"]))
PY
  done
done
