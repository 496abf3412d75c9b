# C5 (asd_mv_triplane) A/B over environment settings:  bash tools/c5_ab.sh OUT "ENV1" "ENV2" ...   (each a string of VAR=value pairs, "-" = none)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-c5ab}; mkdir -p $O; shift
cd $R
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  echo "== env: $e" >> $O/ab.txt
  env $e timeout 600 python bench.py --workload asd_mv_triplane --steps ${STEPS:-8} --warmup ${WARMUP:-4} --no-cpu-baseline ${EXTRA} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms')" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
