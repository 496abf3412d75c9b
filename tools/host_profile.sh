# host-side profile (cProfile) of a secondary workload's steps: bash tools/host_profile.sh OUT WORKLOAD
O=gpurun_out/${1:-hostprof}; mkdir -p $O
python -c "
import cProfile, pstats, sys
sys.argv = ['bench.py', '--workload', '$2', '--steps', '24', '--warmup', '8', '--no-cpu-baseline']
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    st = pstats.Stats(pr, stream=open('$O/$2_host_profile.txt', 'w'))
    st.sort_stats('tottime').print_stats(60)
    st.sort_stats('cumtime').print_stats(70)
" > $O/$2_bench.json 2> $O/$2_err.txt
grep -v "weights.py\|engine.py\|importlib\|sympy\|tokenize\|marshal" $O/$2_host_profile.txt | head -45 | cut -c1-150
