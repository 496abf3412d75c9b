# driver-style check of the tree: the GPU tests with -x, then the default bench line
mkdir -p gpurun_out/${1:-final}
O=gpurun_out/${1:-final}
(timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo EXIT $? >> $O/gputests.log); grep -E "passed|failed|EXIT" $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-330 $O/bench.json
