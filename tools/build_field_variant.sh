#!/bin/bash
# A/B build of libasd_hip.so with other field / paged-scatter constants: tools/build_field_variant.sh NAME "-DASD_FIELD_NAGG=6 -DASD_PG_NF=10"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); NAME=$1; FLAGS=$2; CS=$ROOT/scaledreamer_amd/csrc
mkdir -p $ROOT/scaledreamer_amd/variants
make -s -C $CS >/dev/null 2>&1
for f in field field_paged; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function $FLAGS -c $CS/$f.hip -o /tmp/asd_variant_${NAME}_$f.o
done
OTHERS=$(ls $CS/*.o | grep -v "/field.o\|/field_paged.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scaledreamer_amd/variants/libasd_hip_$NAME.so $OTHERS /tmp/asd_variant_${NAME}_field.o /tmp/asd_variant_${NAME}_field_paged.o
echo built $NAME
