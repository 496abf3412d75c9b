# HBM traffic of bench.py's three single-shape roofline loops (roofline_gemm / _vae_conv / _unet_conv): separate FETCH_SIZE / WRITE_SIZE passes per shape
#   bash tools/r6_pmc_shapes.sh OUT   ->  gpurun_out/OUT/pmc_shapes.json  (copied to profiles/r06_pmc_shapes.json, which bench.py reads)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r6_pmc_shapes}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_shapes
for shape in gemm vae512 unet64; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_shapes/${shape}_$c -- python $R/tools/pmc_shapes.py $shape > /dev/null 2> $O/pmc_shapes_${shape}_$c.err
  echo "$shape $c exit $?"
done; done
python $R/tools/pmc_shapes.py collect /tmp/pmc_shapes $O/pmc_shapes.json
