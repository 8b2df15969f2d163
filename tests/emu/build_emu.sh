#!/bin/sh
# Host-emulation build of the kernels (CPU test-suite only; see csrc/drgnn_rt.h).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/build"
g++ -O1 -g -std=c++17 -fPIC -shared -x c++ -DDRGNN_EMU -Wall -Wno-unused-function -Wno-unused-variable \
    -o "$HERE/build/libdrgnn_emu.so" "$ROOT/deeprank-gnn_amd/csrc/drgnn_capi.hip"
