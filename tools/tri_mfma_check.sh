# MFMA tri-plane field against float64 + A/B with the vector-pipe kernels:  bash tools/tri_mfma_check.sh OUT [args]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-tfm}; mkdir -p $O; shift
cd $R
( echo "== matrix pipe"; timeout 300 python tools/tri_mfma_check.py "$@"; echo "== vector pipe"; ASD_TRI_MFMA=0 timeout 300 python tools/tri_mfma_check.py "$@" ) > $O/check.txt 2>&1
tail -40 $O/check.txt
