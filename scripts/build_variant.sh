#!/bin/bash
# Scratch: build a variant of the library with extra -D flags for walk_kernels.hip.
#   scripts/build_variant.sh NAME -DC5_ELECT_LEADERS=0 ...   -> course5_amd/libcourse5_hip_NAME.so
# (A/B against the in-tree build with scripts/ab_probe.py)
set -e
cd "$(dirname "$0")/.."
name="$1"; shift
python3 -m course5_amd.build >/dev/null
cd course5_amd
mkdir -p _build/var_$name
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wextra -Wno-unused-parameter -I ../include "$@" -c csrc/walk_kernels.hip -o _build/var_$name/walk_kernels.o
hipcc -shared -fPIC --offload-arch=gfx950 -o libcourse5_hip_$name.so _build/exact_kernels.o _build/var_$name/walk_kernels.o  _build/c_api.o _build/adjacency.o -fopenmp
ls -la libcourse5_hip_$name.so
