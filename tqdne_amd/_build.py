"""Build libtqdne_hip.so (hipcc, gfx950 only) in-tree under tqdne_amd/lib/."""

from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libtqdne_hip.so")
SOURCES = ["conv1d_mfma.hip", "conv1d_w4.hip", "small_ops.hip", "attention.hip", "backward.hip"]
ARCH = "gfx950"
# conv1d_w4.hip interleaves scalar fp32 VALU with MFMAs: packed fp32 ops (what the SLP vectoriser makes of adjacent scalar ones) cost
# extra cycles beside MFMAs (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
FILE_FLAGS = {"conv1d_w4.hip": ("-fno-slp-vectorize",)}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "tqdne_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True, extra_flags=(), out_name=None) -> str:
    """Compile every HIP source for gfx950 and link the C-ABI shared library."""
    global_out = LIBPATH if out_name is None else os.path.join(LIBDIR, out_name)
    if out_name is None and not force and not needs_build():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    tag = "" if out_name is None else "." + out_name.replace(".so", "")
    objs = []
    procs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(LIBDIR, src.replace(".hip", tag + ".o"))
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", *FILE_FLAGS.get(src, ()), *extra_flags,
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
    build(force="--force" in sys.argv)
    print(LIBPATH)
