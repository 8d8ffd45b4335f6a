#!/bin/bash
# A variant library for A/B probes (scripts/*_probe.py --lib): ONE source of the library compiled with extra flags, linked with the
# objects of the product build:   scripts/build_variant.sh <name> <source.hip> <flags...>   ->  .scratch/lib_<name>.so
set -e
cd "$(dirname "$0")/../locityper_amd/csrc"
NAME=$1; SRC=$2; shift; shift
make -s >/dev/null
mkdir -p ../../.scratch/$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wextra -Wno-unused-parameter "$@" -c $SRC -o ../../.scratch/$NAME/${SRC%.hip}.o
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,now -o ../../.scratch/lib_$NAME.so ../../.scratch/$NAME/${SRC%.hip}.o $OBJS -L/opt/rocm/lib -lrccl -lz -ldl -Wl,-rpath,/opt/rocm/lib
ls -la ../../.scratch/lib_$NAME.so
