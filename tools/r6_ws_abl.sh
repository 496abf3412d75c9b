# timing-only ablations of the weight-streaming convolution (tools/build_variant.sh ws_X "-DWS_ABL_X" gemm_ws.hip): bash tools/r6_ws_abl.sh OUT
O=gpurun_out/${1:-r6_ws_abl}; mkdir -p $O
WS_ONLY=10 python tools/r6_ws_time.py $O/product.txt 2>/dev/null | cut -c1-40,150-260
for v in NOW NOA NOMMA NOSTORE; do
  ASD_HIP_LIB=$PWD/scaledreamer_amd/variants/libasd_hip_ws_$v.so WS_ONLY=10 python tools/r6_ws_time.py $O/$v.txt 2>/dev/null | cut -c1-40,150-260
done
cat $O/*.txt > $O/all.txt
