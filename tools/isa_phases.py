#!/usr/bin/env python3
"""Static instruction accounting of a kernel compiled with -DCAGYM_PMARK: VALU / SALU / LDS / VMEM instruction counts
between consecutive `; PMARK <name>` comments of the ISA listing (straight-line estimate: loops and branches are NOT
unrolled - read together with the per-phase trip counts in DESIGN.md).
usage: tools/isa_phases.py <file.s> [kernel-symbol-substring]"""
import re
import sys


def classify(op):
    if op.startswith(("v_",)):
        return "valu"
    if op.startswith(("s_",)):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return None


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().split("\n")
    inside = not pat
    name = "<start>"
    acc = {}
    order = []
    for ln in lines:
        if pat and re.match(r"^_Z\w+:", ln):
            inside = pat in ln
            continue
        if not inside:
            continue
        m = re.search(r"; PMARK (\w+)", ln)
        if m:
            name = m.group(1) + "@%d" % len(order)
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c = classify(op)
        if c is None:
            continue
        if name not in acc:
            acc[name] = dict(valu=0, salu=0, lds=0, vmem=0, trans=0, f64=0, dpp=0)
            order.append(name)
        acc[name][c] += 1
        if op.startswith(("v_sqrt", "v_rcp", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")):
            acc[name]["trans"] += 1
        if "_f64" in op:
            acc[name]["f64"] += 1
    print("%-28s %6s %6s %5s %5s %6s %6s" % ("after mark", "VALU", "SALU", "LDS", "VMEM", "trans", "f64"))
    for n in order:
        a = acc[n]
        print("%-28s %6d %6d %5d %5d %6d %6d" % (n, a["valu"], a["salu"], a["lds"], a["vmem"], a["trans"], a["f64"]))
    tot = {k: sum(a[k] for a in acc.values()) for k in ("valu", "salu", "lds", "vmem")}
    print("total", tot)


if __name__ == "__main__":
    main()
