# shader clock / power sampled by rocm-smi while the headline step runs: bash tools/clock_probe.sh OUT
O=gpurun_out/${1:-clk}; mkdir -p $O
(python bench.py --steps 4000 --warmup 20 --no-cpu-baseline > $O/bench.json 2>/dev/null) &
BP=$!
sleep 35
for i in $(seq 1 12); do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | tr '\n' ';' >> $O/clock_samples.txt; echo >> $O/clock_samples.txt; sleep 0.5; done
wait $BP
cat $O/clock_samples.txt | cut -c1-400 | head -14; cut -c1-150 $O/bench.json
rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -2
