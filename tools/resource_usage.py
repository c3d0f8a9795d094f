"""Registers / scratch / occupancy of every generation-3 kernel specialisation, as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage on each cagym_k3_tu.hip unit; runs in the development container, no GPU).
usage: python tools/resource_usage.py [extra hipcc flags ...] > profiles/r4/kernel_resources.txt"""
import concurrent.futures
import importlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("gym-exploration-2d_amd.build")


def one(job):
    obj, src, defs, extra = job
    with tempfile.TemporaryDirectory() as td:
        cmd = ["hipcc", "--offload-arch=" + b.ARCH, "-c"] + b.FLAGS + defs + extra + ["-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(td, "x.o"), os.path.join(b.CSRC, src)]
        p = subprocess.run(cmd, capture_output=True, text=True)
    rows = []
    cur = None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None and key not in cur:
                cur[key] = int(m.group(1))
    return obj, rows, p.returncode


def main():
    extra = sys.argv[1:]
    jobs = [(o, s, d, extra) for (o, s, d, _h) in b.units() if o.startswith("k3_")]
    with concurrent.futures.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        res = list(ex.map(one, jobs))
    print("%-64s %5s %5s %5s %8s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ"))
    for obj, rows, rc in res:
        if rc:
            print(obj, "COMPILE FAILED")
        for r in rows:
            n = re.sub(r"^void ", "", r["name"])
            n = re.sub(r"\(.*$", "", n)
            print("%-64s %5d %5d %5d %8d %4d" % (n, r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1), r.get("scratch", -1), r.get("occ", -1)))


if __name__ == "__main__":
    main()
