#!/bin/bash
# 2048 x 20 RVO bench for several alternative libraries
mkdir -p gpurun_out/exp
python bench.py --no-cpu-baseline --worlds 2048 --agents 20 > gpurun_out/exp/m20_base.json || exit 1
python -c "
import json;d=json.load(open('gpurun_out/exp/m20_base.json'));print('base', '%.1f M env-steps/s'%(d['value']/1e6), 'launch %.3f ms'%d['roofline']['launch_ms'])"
for L in "$@"; do
  T=$(basename $L .so)
  CAGYM_LIB=$L python bench.py --no-cpu-baseline --worlds 2048 --agents 20 > gpurun_out/exp/$T.json || exit 1
  CAGYM_LIB=$L python bench.py --no-cpu-baseline --worlds 16384 --agents 20 --steps 512 --warmup 64 --pool-factor 2 > gpurun_out/exp/${T}_16k.json || exit 1
  python - <<PY
import json
for f in ("$T", "${T}_16k"):
    d = json.load(open("gpurun_out/exp/%s.json" % f))
    print(f, "%.1f M env-steps/s" % (d["value"] / 1e6), "launch %.3f ms" % d["roofline"]["launch_ms"])
PY
done
