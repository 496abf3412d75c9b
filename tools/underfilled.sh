R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-uf}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_uf
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_uf -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench.json 2>/tmp/uf.err
python $R/tools/underfilled.py /tmp/prof_uf > $O/underfilled.txt; head -60 $O/underfilled.txt | cut -c1-170
