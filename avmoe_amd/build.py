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
if os.environ.get("AVMOE_DEV_BUILD"):        # development build: the A/B switches of scripts/README.md are compiled in (csrc/common.h: dev_env)
    FLAGS.append("-DAVMOE_DEV")


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


def _obj_digest(src):
    """source + every header / include file of csrc/ + the public header + the flags: what an object depends on"""
    h = hashlib.sha256()
    deps = [src] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "avmoe.h"))
    for p in deps:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every .hip/.cpp under csrc/ into one shared library.  Objects are built in parallel
    (one hipcc per translation unit; an object whose source, headers and flags are unchanged is kept) and then linked."""
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and is_fresh():
        return LIB
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]
    procs = []
    objs = []
    jobs = int(os.environ.get("AVMOE_BUILD_JOBS", str(max(1, min(8, os.cpu_count() or 1)))))
    pending = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        dig, stamp = _obj_digest(src), obj + ".stamp"
        if not force and os.path.isfile(obj) and os.path.isfile(stamp) and open(stamp).read().strip() == dig:
            continue
        pending.append((src, obj, stamp, dig))
    pending.sort(key=lambda t: -os.path.getsize(t[0]) if not os.path.basename(t[0]).startswith("tile_gen") else -10 ** 9)   # longest first
    running = []

    def reap(block):
        for item in list(running):
            src, proc, stamp, dig = item
            if block:
                proc.wait()
            if proc.poll() is None:
                continue
            running.remove(item)
            if proc.returncode != 0:
                for _s, pr, _a, _b in running:
                    pr.wait()
                raise RuntimeError(f"hipcc failed on {src}")
            with open(stamp, "w") as fh:
                fh.write(dig)
            if block:
                return
    import time as _t
    for src, obj, stamp, dig in pending:
        while len(running) >= jobs:
            reap(False)
            if len(running) >= jobs:
                _t.sleep(0.05)
        cmd = [HIPCC] + cflags + ["-c", src, "-o", obj]
        if verbose:
            print("[avmoe_amd.build]", " ".join(cmd), flush=True)
        running.append((src, subprocess.Popen(cmd), stamp, dig))
    while running:
        reap(True)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print("[avmoe_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(_digest())
    return LIB


HOST_LIB = os.path.join(LIBDIR, "libavmoe_host.so")


def build_host(force: bool = False, verbose: bool = True) -> str:
    """g++ the CPU implementation of the ABI (csrc/host_*.cpp, include/avmoe_host.h) into libavmoe_host.so: TEST / CI infrastructure
    (tests/test_host_golden.py), never loaded by the product path."""
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.startswith("host_") and f.endswith(".cpp")]
    inc = os.path.join(os.path.dirname(HERE), "include")
    deps = srcs + [os.path.join(inc, "avmoe.h"), os.path.join(inc, "avmoe_host.h")]
    if not force and os.path.isfile(HOST_LIB) and all(os.path.getmtime(HOST_LIB) >= os.path.getmtime(p) for p in deps):
        return HOST_LIB
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-fopenmp", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-parameter"] + srcs + ["-o", HOST_LIB]
    if verbose:
        print("[avmoe_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return HOST_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
    print(build_host(force="--force" in sys.argv))
