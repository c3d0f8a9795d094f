#!/bin/bash
# cfg4's env kernel at 2 vs 3 workgroups per CU, same kernel, same worlds.  4 worlds per workgroup (CAGYM_WPW10=4) and worlds with 2..4
# rectangles need 50.0 KB of LDS (3 per CU); CAGYM_LDS_PAD (diagnostic build: python build.py --alt ldspad cagym_api -DCAGYM_DIAG_LDS_PAD)
# adds unused bytes to force 2 per CU.  Then BASELINE's 2..10 rectangles with 4 and 5 worlds per workgroup (61.1 / 75.9 KB: 2 per CU both).
mkdir -p gpurun_out/exp
export CAGYM_LIB=gym-exploration-2d_amd/csrc/libcagym_hip_ldspad.so
run() {  # label, K, wpw, pad
  CAGYM_WPW10=$3 CAGYM_LDS_PAD=$4 python bench.py --config cfg4 --max-obstacles $2 --no-cpu-baseline > gpurun_out/exp/c4.json || exit 1
  python - <<PY
import json
a = json.load(open("gpurun_out/exp/c4.json"))
print("$1: %.4f ms per step, env kernel %.1f us, ga3c %.1f us" % (a["ms_per_step"], a["cfg4"]["env_kernel_ms_per_step"] * 1e3, a["cfg4"]["ga3c_ms_per_step"] * 1e3))
PY
}
for rep in 1 2; do
  run "2..4 rectangles, 4 worlds/wg, pad 0    (50.0 KB, 3 per CU) rep $rep" 4 4 0
  run "2..4 rectangles, 4 worlds/wg, pad 6144 (56.0 KB, 2 per CU) rep $rep" 4 4 6144
  run "2..4 rectangles, 5 worlds/wg, pad 0    (59.2 KB, 2 per CU) rep $rep" 4 5 0
  run "2..10 rectangles, 4 worlds/wg (61.1 KB, 2 per CU) rep $rep" 10 4 0
  run "2..10 rectangles, 5 worlds/wg (75.9 KB, 2 per CU; shipped choice) rep $rep" 10 5 0
done
