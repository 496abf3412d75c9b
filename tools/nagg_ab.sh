#!/bin/bash
# A/B of the run-aggregated level count of the hash-grid gradient scatter (ASD_FIELD_NAGG) on the SDF-mode (Hyper-iNGP) and the
# headline step: tools/build_variant.sh nagg12 "-DASD_FIELD_NAGG=12" field.hip etc. first.  Writes gpurun_out/$1/*.json
OUT=gpurun_out/${1:-nagg}; mkdir -p $OUT
V=scaledreamer_amd/variants
for rep in 1 2; do
  for name in default nagg12 nagg16; do
    lib=""; [ $name != default ] && lib=$PWD/$V/libasd_hip_$name.so
    ASD_HIP_LIB=$lib python bench.py --workload asd_sd_hyper_ingp --no-cpu-baseline > $OUT/hyper_${name}_$rep.json 2>> $OUT/err.log
    python - <<PY
import json; d=json.load(open("$OUT/hyper_${name}_$rep.json")); print("hyper", "$name", $rep, d["value"], d["ms_per_step"])
PY
  done
done
for name in default nagg12; do
  lib=""; [ $name != default ] && lib=$PWD/$V/libasd_hip_$name.so
  ASD_HIP_LIB=$lib python bench.py --no-cpu-baseline > $OUT/c2_${name}.json 2>> $OUT/err.log
  python - <<PY
import json; d=json.load(open("$OUT/c2_${name}.json")); print("c2", "$name", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])
PY
done
