#!/bin/bash
# A/B builds of libasd_hip.so: tools/build_variant.sh NAME "-DFLAG=..." [file.hip]  ->  scaledreamer_amd/variants/libasd_hip_NAME.so
# (the named source is recompiled with the extra flags, every other object comes from the regular build)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; SRC=${3:-gemm.hip}
CS=$ROOT/scaledreamer_amd/csrc
mkdir -p $ROOT/scaledreamer_amd/variants
make -s -C $CS >/dev/null
OBJ=/tmp/asd_variant_${NAME}_${SRC%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function $FLAGS -c $CS/$SRC -o $OBJ
OTHERS=$(ls $CS/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scaledreamer_amd/variants/libasd_hip_$NAME.so $OTHERS $OBJ
echo built $NAME
