"""Build libcagym_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The library is linked against the HIP runtime PyTorch-ROCm bundles (torch/lib/libamdhip64.so,
SONAME libamdhip64.so.7) so that torch streams and device pointers are valid inside it.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libcagym_hip.so")
SOURCES = ["cagym_api.hip"]
HEADERS = ["cagym_device.h", "cagym_orca.h", "cagym_kernels.h", "cagym_kernels2.h", "cagym_kernels3.h", "cagym_ig.h", "cagym_ga3c.h", "cagym_gen.h", "cagym_dmcts.h", "../../include/cagym.h"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-value",
         # MachineLICM hoists ~70 fp64 polynomial literals (sincos/atan2) out of the rollout's step loop and
         # keeps them in VGPRs: 168 VGPRs + spills instead of 106 (measured)
         "-mllvm", "-disable-machine-licm"]


def _torch_lib_dir():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "lib")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    tl = _torch_lib_dir()
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=" + ARCH, "-c"] + FLAGS + ["-o", obj, os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = ["g++", "-shared", "-o", LIB] + objs + ["-L" + tl, "-lamdhip64", "-Wl,-rpath," + tl]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
