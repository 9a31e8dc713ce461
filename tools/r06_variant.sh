#!/bin/bash
# build an alternative library from the same sources: tools/r06_variant.sh <name> [extra hipcc flags...]  -> build/<name>.so (in-tree: travels with gpurun)
cd "$(dirname "$0")/.." && mkdir -p build
NAME=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -x hip -ffp-contract=on "$@" scpp_amd/csrc/scpp_hip.cpp -o build/$NAME.so 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^\|generated\|^In file\|note:" | head -20
ls -la build/$NAME.so
