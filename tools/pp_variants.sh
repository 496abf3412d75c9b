# A/B of the ping-pong window convolution's schedule switches on the same box: bash tools/pp_variants.sh OUT   (build the variants first, see below)
#   tools/build_variant.sh ppnostagger "-DASD_PP_NO_STAGGER" gemm_pp.hip; tools/build_variant.sh ppnoprio "-DASD_PP_NO_PRIO" gemm_pp.hip
O=gpurun_out/${1:-ppv}; mkdir -p $O
for i in 1 2; do
  for v in base ppnostagger ppnoprio; do
    if [ $v = base ]; then L=scaledreamer_amd/libasd_hip.so; else L=scaledreamer_amd/variants/libasd_hip_$v.so; fi
    echo "== $v" >> $O/variants.txt
    ASD_HIP_LIB=$L python tools/pp_ab.py quick 2>/dev/null | grep -o "^([^:]*\|pp[0-9x]*/s[0-9]*: *[0-9.]* us *[0-9.]* PF" | tr '\n' ' ' >> $O/variants.txt; echo >> $O/variants.txt
  done
done
cat $O/variants.txt
