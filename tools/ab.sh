#!/bin/bash
# A/B of two builds of the library on the headline bench (diagnostics; run on the GPU box)
# usage: tools/ab.sh <alt_lib.so> [tag]
set -e
mkdir -p gpurun_out/exp
T=${2:-alt}
python bench.py --no-cpu-baseline > gpurun_out/exp/base.json
CAGYM_LIB=$1 python bench.py --no-cpu-baseline > gpurun_out/exp/$T.json
python bench.py --no-cpu-baseline --worlds 65536 --steps 512 --warmup 128 --pool-factor 2 > gpurun_out/exp/base64k.json
CAGYM_LIB=$1 python bench.py --no-cpu-baseline --worlds 65536 --steps 512 --warmup 128 --pool-factor 2 > gpurun_out/exp/${T}64k.json
python - <<PY
import json
for f in ("base", "$T", "base64k", "${T}64k"):
    d = json.load(open("gpurun_out/exp/%s.json" % f))
    print(f, "%.1f M env-steps/s" % (d["value"] / 1e6), "launch %.3f ms" % d["roofline"]["launch_ms"], "frac %.4f" % d["roofline"]["frac"])
PY
