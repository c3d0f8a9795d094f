#!/bin/bash
# four rocprofv3 PMC passes of tools/valu_breakdown.py -> wave-VALU instructions per workgroup-step of each variant
# usage (GPU box): tools/valu_breakdown.sh [outdir]        (CAGYM_LIB selects an alternative library)
O=${1:-gpurun_out/valu}
mkdir -p $O
export TMPDIR=/tmp
for pol in rvo noncoop; do for ob in obs noobs; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU -d $O/${pol}_${ob} -o run -- python3 tools/valu_breakdown.py $pol $ob > /dev/null 2>&1 || exit 1
  python tools/pmc_summary.py $O/${pol}_${ob} k_rollout3 > $O/${pol}_${ob}.txt
done; done
python - "$O" <<'PY'
import re, sys
o = sys.argv[1]
def get(f):
    t = open(f).read()
    v = float(re.search(r"SQ_INSTS_VALU\s+(\S+)", t).group(1))
    a = float(re.search(r"SQ_ACTIVE_INST_VALU\s+(\S+)", t).group(1))
    g = float(re.search(r"GRBM_GUI_ACTIVE\s+(\S+)", t).group(1))
    ns = float(re.search(r"avg_ns=(\S+)", t).group(1))
    return v, a, g, ns
wg_steps = 1024 * 128.0
res = {}
for pol in ("rvo", "noncoop"):
    for ob in ("obs", "noobs"):
        v, a, g, ns = get("%s/%s_%s.txt" % (o, pol, ob))
        res[pol, ob] = v / wg_steps
        print("%-8s %-6s  VALU/wg-step %7.1f   us/step %6.3f   VALU busy %4.1f %%" % (pol, ob, v / wg_steps, ns / 128e3, 100 * 4 * a / (1024 * g / 8)))
print("rows (OAS + ego obs)            %7.1f" % (res["rvo", "obs"] - res["rvo", "noobs"]))
print("ORCA (half-planes, rank, LP...) %7.1f" % (res["rvo", "noobs"] - res["noncoop", "noobs"]))
print("rest (S1, pairs, S2, ego frame) %7.1f" % res["noncoop", "noobs"])
PY
