# packed-polynomial GEGLU epilogue: parity tests, per-shape timing, step time     bash tools/geglu_poly_ab.sh OUT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-geglu}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_diffusion_ops.py tests/test_gpu_diffusion_goldens.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt; cat $O/tests.txt
timeout 300 python tools/geglu_ab.py 2>&1 | cut -c1-160 | tee $O/shapes.txt
for i in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done | tee $O/steps.txt
