"""Join the launch order of the step's GEMM / conv calls (gpurun_out/gemm_order.txt, tools/gemm_shapes.py) with their durations inside the
step (gpurun_out/<brk>/launch_list.txt from tools/step_breakdown.sh with the pattern 'gemm_f16_kernel|conv3x3_') and their standalone
times (gpurun_out/gemm_shapes.txt): which launches lose most inside the step?     python tools/gemm_in_step.py gpurun_out/brk6"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
order = [tuple(int(v) for v in l.split()) for l in open(os.path.join(ROOT, "gpurun_out", "gemm_order.txt"))]
durs = [float(v) for v in open(os.path.join(sys.argv[1], "launch_list.txt")).read().strip().splitlines()[-1].split()]
alone = {}
for l in open(os.path.join(ROOT, "gpurun_out", "gemm_shapes.txt")):
    if l.startswith("#"):
        continue
    f = l.split()
    alone[tuple(int(v) for v in f[5:])] = float(f[2])
print(len(order), "calls traced,", len(durs), "kernels in the step")
assert len(order) == len(durs), "the two orders do not line up"
agg = {}
for k, d in zip(order, durs):
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += d
rows = sorted(((v[1] - v[0] * alone.get(k, 0.0), k, v) for k, v in agg.items()), reverse=True)
print("# in-step minus standalone (us per step) | count | in-step us | standalone us (main kernel + split-K epilogue, cold operands) | M N K conv Hin Win Cin Hout Wout s p u cfg split act res f32 gn")
for lost, k, v in rows[:40]:
    print(f"{lost:8.1f} {v[0]:3d} {v[1] / v[0]:8.1f} {alone.get(k, float('nan')):8.1f}   " + " ".join(str(x) for x in k))
print(f"total in-step {sum(durs):.0f} us, standalone {sum(v[0] * alone.get(k, 0.0) for k, v in agg.items()):.0f} us (standalone includes the split-K epilogue launches, in-step does not)")
