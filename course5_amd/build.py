"""In-tree build of the gfx950 shared library (hipcc, no cmake needed).

    python -m course5_amd.build          # builds course5_amd/libcourse5_hip.so (+ the `course` CLI)

Objects go to course5_amd/_build/ (git-ignored); the .so stays in-tree so it travels to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "_build")
LIB = os.path.join(PKG, "libcourse5_hip.so")
CLI = os.path.join(PKG, "course")
ARCH = "gfx950"

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wextra",
          "-Wno-unused-parameter", "-I", os.path.join(PKG, "..", "include")]

# (source, extra flags)
LIB_SOURCES = [
    ("exact_kernels.hip", ["-ffp-contract=off"]),   # must round like the reference's host build
    ("walk_kernels.hip", []),
    ("c_api.hip", []),
    ("adjacency.cpp", ["-x", "c++", "-fopenmp"]),
]


DEVICE_SOURCES = ("device_types.hpp", "kernels.hpp", "walk_common.hpp", "exact_kernels.hip", "walk_kernels.hip")


def kernel_source_hash() -> str:
    """sha256 over the device-side sources with comments and white space removed: committed profiles carry the hash
    of the kernels they were measured on, and bench.py says when the kernels have moved on since (an edited comment or
    a change to the host side of the ABI does not make a profile stale)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in DEVICE_SOURCES:
        with open(os.path.join(CSRC, name)) as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update(name.encode())
        h.update("".join(text.split()).encode())
    return h.hexdigest()[:16]


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd: list[str], verbose: bool) -> None:
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build_library(verbose: bool = False, force: bool = False) -> str:
    os.makedirs(OUT, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith((".hpp", ".h"))]
    headers.append(os.path.join(PKG, "..", "include", "course5_hip.h"))
    objs = []
    for src, extra in LIB_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OUT, os.path.splitext(src)[0] + ".o")
        if force or _newer(o, [s] + headers):
            _run([HIPCC] + COMMON + extra + ["-c", s, "-o", o], verbose)
        objs.append(o)
    if force or _newer(LIB, objs):
        _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-fopenmp"], verbose)
    return LIB


def build_cli(verbose: bool = False, force: bool = False) -> str | None:
    host_dir = os.path.join(CSRC, "host")
    main = os.path.join(host_dir, "main.cpp")
    if not os.path.exists(main):
        return None
    srcs = sorted(os.path.join(host_dir, f) for f in os.listdir(host_dir) if f.endswith(".cpp"))
    deps = srcs + [os.path.join(host_dir, f) for f in os.listdir(host_dir) if f.endswith(".hpp")] + [LIB]
    if force or _newer(CLI, deps):
        _run(["g++", "-O3", "-std=c++17", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(PKG, "..", "include"),
              "-I", "/opt/rocm/include", "-I", host_dir, "-o", CLI] + srcs +
             [f"-L{PKG}", "-lcourse5_hip", "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-Wl,-rpath,$ORIGIN",
              "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-fopenmp", "-lz"], verbose)
    return CLI


def build_all(verbose: bool = False, force: bool = False) -> None:
    build_library(verbose, force)
    build_cli(verbose, force)


if __name__ == "__main__":
    build_all(verbose=True, force="--force" in sys.argv)
