"""Builds the in-tree HIP library (libavmoe_hip.so, gfx950 only) with hipcc.

`python -m avmoe_amd.build` or `avmoe_amd.build.build()`; hipcc cross-compiles without a GPU.  The
built .so is git-ignored but travels to the GPU box with the repo snapshot."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libavmoe_hip.so")
STAMP = os.path.join(LIBDIR, "libavmoe_hip.stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"]


def sources():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".cpp")) and not f.startswith("host_"):
            out.append(os.path.join(CSRC, f))
    return out


def _digest():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    files.append(os.path.join(os.path.dirname(HERE), "include", "avmoe.h"))
    for p in files:
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode())      # not the absolute path: the GPU box has the repo elsewhere
            with open(p, "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_fresh() -> bool:
    if not (os.path.isfile(LIB) and os.path.isfile(STAMP)):
        return False
    with open(STAMP) as fh:
        return fh.read().strip() == _digest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every .hip/.cpp under csrc/ into one shared library.  Objects are built in parallel
    (one hipcc per translation unit) and then linked."""
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and is_fresh():
        return LIB
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]
    procs = []
    objs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [HIPCC] + cflags + ["-c", src, "-o", obj]
        if verbose:
            print("[avmoe_amd.build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print("[avmoe_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(_digest())
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
