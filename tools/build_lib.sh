#!/bin/bash
# usage: tools/build_lib.sh [output .so] [extra hipcc flags...]   (default: the product library, same flags as __graft_entry__.build)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-$root/jackalope_amd/csrc/libjackalope_hip.so}
shift || true
cd "$root/jackalope_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-literal-range "$@" -o "$out" jk_api.hip -lz -lpthread
