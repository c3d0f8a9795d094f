"""Build libcagym_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The library is linked against the HIP runtime PyTorch-ROCm bundles (torch/lib/libamdhip64.so,
SONAME libamdhip64.so.7) so that torch streams and device pointers are valid inside it.

Translation units (compiled in parallel, objects cached under csrc/build/ by source + flag hash):
  cagym_api.hip                      the C ABI, generation-1 kernels, reset / laserscan / raster / generator, GA3C, IG, Dec-MCTS
  cagym_k3_tu.hip x 12               one per generation-3 specialisation (cagym_launch3.h: CAGYM_K3_SPECS x OBST)
"""
import concurrent.futures
import functools
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libcagym_hip.so")
K3_HEADERS = ["cagym_device.h", "cagym_trace.h", "cagym_spin.h", "cagym_orca.h", "cagym_kernels.h", "cagym_kernels3.h", "cagym_split3.h", "cagym_launch3.h",
              "../../include/cagym.h"]
HEADERS = K3_HEADERS + ["cagym_ig.h", "cagym_ga3c.h", "cagym_ga3c16.h", "cagym_gen.h", "cagym_dmcts.h"]


def _k3_specs():
    """(lanes, compile-time M, worlds per workgroup) rows of CAGYM_K3_SPECS, read from cagym_launch3.h: ONE list for the C ABI's
    dispatch table and for the translation units built here."""
    text = open(os.path.join(CSRC, "cagym_launch3.h")).read()
    m = re.search(r"#define\s+CAGYM_K3_SPECS\(X\)(.*)", text)
    rows = [tuple(int(v) for v in r) for r in re.findall(r"X\(\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*\)", m.group(1))] if m else []
    if not rows:
        raise RuntimeError("CAGYM_K3_SPECS not found in cagym_launch3.h")
    return rows


K3_SPECS = _k3_specs()
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-value",
         # MachineLICM hoists ~70 fp64 polynomial literals (sincos/atan2) out of the rollout's step loop and
         # keeps them in VGPRs: 168 VGPRs + spills instead of 106 (measured)
         "-mllvm", "-disable-machine-licm",
         # no automatic packing of fp32 pairs: on gfx950 a v_pk_mul/add_f32 issues in 4.5 - 4.9 cycles against 2.6 - 3.0 for the scalar
         # op (tools/micro/pk_f32_rate.hip), and the SLP vectorizer's 168 packed operations + their register shuffling cost the
         # headline kernel 1.9 % (profiles/r4/ab_compiler_flags.txt: 512-step launch 3.873 -> 3.804 ms)
         "-fno-slp-vectorize"]


def units():
    """(object name, source, extra -D flags, headers it depends on)"""
    u = [("cagym_api.o", "cagym_api.hip", [], HEADERS)]
    for nt, mt, wp in K3_SPECS:
        for ob in (0, 1):
            u.append(("k3_%d_%d_%d_%d.o" % (nt, mt, wp, ob), "cagym_k3_tu.hip",
                      ["-DK3_NT=%d" % nt, "-DK3_MT=%d" % mt, "-DK3_WP=%d" % wp, "-DK3_OBST=%d" % ob], K3_HEADERS))
    return u


def _torch_lib_dir():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "lib")


@functools.lru_cache(maxsize=None)
def _hipcc_version():
    try:
        return subprocess.run([os.environ.get("HIPCC", "hipcc"), "--version"], capture_output=True, text=True).stdout
    except OSError:
        return "?"


def _digest(src, defs, headers, extra):
    h = hashlib.sha256()
    h.update(_hipcc_version().encode())  # objects of another compiler are stale
    for f in [src] + headers:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS + defs + extra + [ARCH]).encode())
    return h.hexdigest()


def _stale(extra):
    out = []
    for obj, src, defs, headers in units():
        d = _digest(src, defs, headers, extra)
        stamp = os.path.join(OBJ, obj + ".sha")
        if not (os.path.exists(os.path.join(OBJ, obj)) and os.path.exists(stamp) and open(stamp).read() == d):
            out.append((obj, src, defs, d))
    return out


def needs_build(extra=()):
    return not os.path.exists(LIB) or bool(_stale(list(extra)))


def _compile(job):
    obj, src, defs, digest, extra, verbose = job
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc, "--offload-arch=" + ARCH, "-c"] + FLAGS + defs + extra + ["-o", os.path.join(OBJ, obj), os.path.join(CSRC, src)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(os.path.join(OBJ, obj + ".sha"), "w") as fh:
        fh.write(digest)
    return obj


def build(force=False, verbose=False, extra=(), jobs=None):
    """extra: additional compiler flags for every unit (diagnostic builds)."""
    extra = list(extra)
    os.makedirs(OBJ, exist_ok=True)
    todo = [(o, s, d, g) for (o, s, d, _h) in units() for g in [_digest(s, d, _h, extra)]] if force else _stale(extra)
    if not todo and os.path.exists(LIB):
        return LIB
    jobs = jobs or min(len(todo), max(1, (os.cpu_count() or 2)))
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(_compile, [(o, s, d, g, extra, verbose) for (o, s, d, g) in todo]))
    tl = _torch_lib_dir()
    cmd = ["g++", "-shared", "-o", LIB] + [os.path.join(OBJ, u[0]) for u in units()] + ["-L" + tl, "-lamdhip64", "-Wl,-rpath," + tl]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


def build_variant(tag, extra=(), verbose=False):
    """Diagnostic library csrc/libcagym_hip_<tag>.so: ONE translation unit (-DCAGYM_MONOLITHIC: the trace buffers are
    device globals shared by the C ABI's debug hooks and the kernels) compiled with the extra flags, e.g.
    build_variant("wavetrace", ["-DCAGYM_WAVETRACE"]).  Used by tools/*.py; never the shipped library."""
    lib = os.path.join(CSRC, "libcagym_hip_%s.so" % tag)
    os.makedirs(OBJ, exist_ok=True)
    obj = os.path.join(OBJ, "mono_%s.o" % tag)
    d = _digest("cagym_api.hip", ["-DCAGYM_MONOLITHIC"] + list(extra), HEADERS + ["cagym_k3_tu.hip", "cagym_k3_all.inc"], [])
    stamp = obj + ".sha"
    if not (os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == d):
        hipcc = os.environ.get("HIPCC", "hipcc")
        cmd = [hipcc, "--offload-arch=" + ARCH, "-c", "-DCAGYM_MONOLITHIC"] + list(extra) + FLAGS + ["-o", obj, os.path.join(CSRC, "cagym_api.hip")]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        tl = _torch_lib_dir()
        subprocess.check_call(["g++", "-shared", "-o", lib, obj, "-L" + tl, "-lamdhip64", "-Wl,-rpath," + tl])
        with open(stamp, "w") as fh:
            fh.write(d)
    return lib


def build_alt(tag, extra, only):
    """A/B library csrc/libcagym_hip_<tag>.so: the units whose object name contains one of `only` are recompiled with the
    extra flags (objects under csrc/build/alt_<tag>/), every other unit is the default build's object.  E.g.
    build_alt("minw4", ["-DCAGYM_MINW_WIDE=4"], ["k3_256_20_2_0"]).  Select with CAGYM_LIB=<path>."""
    build()
    alt = os.path.join(OBJ, "alt_" + tag)
    os.makedirs(alt, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    objs = []
    jobs = []
    for obj, src, defs, _h in units():
        if any(o in obj for o in only):
            out = os.path.join(alt, obj)
            jobs.append([hipcc, "--offload-arch=" + ARCH, "-c"] + FLAGS + defs + list(extra) + ["-o", out, os.path.join(CSRC, src)])
            objs.append(out)
        else:
            objs.append(os.path.join(OBJ, obj))
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 2))) as ex:
        list(ex.map(subprocess.check_call, jobs))
    lib = os.path.join(CSRC, "libcagym_hip_%s.so" % tag)
    tl = _torch_lib_dir()
    subprocess.check_call(["g++", "-shared", "-o", lib] + objs + ["-L" + tl, "-lamdhip64", "-Wl,-rpath," + tl])
    return lib


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "--alt":  # python build.py --alt <tag> <unit-substring> [flags...]
        print(build_alt(sys.argv[2], sys.argv[4:], [sys.argv[3]]))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":  # python build.py --variant <tag> [flags...]
        print(build_variant(sys.argv[2], sys.argv[3:], verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
