"""Build libtqdne_hip.so (hipcc, gfx950 only) in-tree under tqdne_amd/lib/."""

from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libtqdne_hip.so")
# (the conv kernel template, csrc/conv1d_kernel.hpp, is instantiated by five translation units so that the build runs in parallel)
SOURCES = ["conv1d_fwd_k5a.hip", "conv1d_fwd_k5b.hip", "conv1d_fwd_k13.hip", "conv1d_resample.hip", "conv1d_dgrad.hip", "conv1d_mfma.hip",
           "small_ops.hip", "attention.hip", "backward.hip"]
# Opt-in kernels that lost their A/B against the default ones: the one-wave-per-SIMD conv (conv1d_w4.hip), the
# slim 64-channel tile and the in-launch GroupNorm fold (TqGnFuse) inside conv1d_mfma.hip.  TQDNE_BUILD_EXPERIMENTS=1 compiles them
# (-DTQ_BUILD_EXPERIMENTS) into a library of its own, libtqdne_hip_exp.so, which `_lib` then loads; the default library does not
# carry them.
EXPERIMENTS = os.environ.get("TQDNE_BUILD_EXPERIMENTS", "0") == "1"
EXP_SOURCES = ["conv1d_w4.hip"]
EXP_LIBPATH = os.path.join(LIBDIR, "libtqdne_hip_exp.so")
ARCH = "gfx950"
# conv1d_w4.hip interleaves scalar fp32 VALU with MFMAs: packed fp32 ops (what the SLP vectoriser makes of adjacent scalar ones) cost
# extra cycles beside MFMAs (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
FILE_FLAGS = {"conv1d_w4.hip": ("-fno-slp-vectorize",)}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "tqdne_hip.h")]


def needs_build(path: str = None) -> bool:
    path = path or LIBPATH
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in _deps() if os.path.exists(d))


def build(force: bool = False, verbose: bool = True, extra_flags=(), out_name=None, experiments=None) -> str:
    """Compile every HIP source for gfx950 and link the C-ABI shared library (``experiments``: default = TQDNE_BUILD_EXPERIMENTS)."""
    experiments = EXPERIMENTS if experiments is None else experiments
    if out_name is None and experiments:
        out_name, default_out = os.path.basename(EXP_LIBPATH), True
    else:
        default_out = out_name is None
    global_out = LIBPATH if out_name is None else os.path.join(LIBDIR, out_name)
    if default_out and not force and not needs_build(global_out):
        return global_out
    os.makedirs(LIBDIR, exist_ok=True)
    tag = "" if out_name is None else "." + out_name.replace(".so", "")
    if experiments:
        extra_flags = tuple(extra_flags) + ("-DTQ_BUILD_EXPERIMENTS",)
    objs = []
    procs = []
    for src in SOURCES + (EXP_SOURCES if experiments else []):
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(LIBDIR, src.replace(".hip", tag + ".o"))
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-ignored-attributes", "-Wno-cuda-compat", *FILE_FLAGS.get(src, ()), *extra_flags,
               "-c", path, "-o", obj]
        if verbose:
            print("[tqdne_amd build]", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
        if verbose and out.strip():
            print(out)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", global_out] + objs
    if verbose:
        print("[tqdne_amd build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return global_out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
