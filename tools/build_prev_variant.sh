#!/bin/bash
# Build lib/variants/prev.so from the kernel sources of a git revision (default HEAD) for same-box A/B runs against the working tree.
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/ab_prev; rm -rf $T; mkdir -p $T/pkg $T/include
git -C $ROOT archive $REV meshgraphnets.jl_amd/csrc include | tar -x -C $T
mkdir -p $ROOT/meshgraphnets.jl_amd/lib/variants
cd $T/meshgraphnets.jl_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -I $T/include $(ls *.hip *.cpp) -o $ROOT/meshgraphnets.jl_amd/lib/variants/prev.so -ldl -lrt -lpthread 2>&1 | grep -i " error" ; ls -la $ROOT/meshgraphnets.jl_amd/lib/variants/prev.so
