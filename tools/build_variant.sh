#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> [-Dflag ...]  -> tools/ab/lib_<name>.so (A/B through AB_LIB=...)
set -e
cd "$(dirname "$0")/../sparenet_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../../tools/ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src -o /tmp/ab_${name}.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/lib_${name}.so $objs /tmp/ab_${name}.o
echo built tools/ab/lib_${name}.so
