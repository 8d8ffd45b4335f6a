#!/bin/bash
# Developer experiment: liblocityper_hip built with ONE source file patched by a sed script (results may be wrong on purpose; only kernel
# times are read, by the probes' --lib).   scripts/build_experiment.sh <name> <file.hip> <sed expression> [<sed expression> ...]
set -e
NAME=$1; FILE=$2; shift; shift
cd "$(dirname "$0")/../locityper_amd/csrc"
mkdir -p ../exp
ARGS=(); for e in "$@"; do ARGS+=(-e "$e"); done
sed "${ARGS[@]}" "$FILE" > /tmp/exp_$NAME.hip
echo "$NAME: $(diff "$FILE" /tmp/exp_$NAME.hip | grep -c '^>') lines patched"
OBJS=$(ls *.o | grep -v "${FILE%.hip}.o")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I. -c /tmp/exp_$NAME.hip -o /tmp/exp_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/liblocityper_hip_$NAME.so $OBJS /tmp/exp_$NAME.o -L/opt/rocm/lib -lrccl -lz -ldl -Wl,-rpath,/opt/rocm/lib
