# per-kernel time of one small-M conv shape under chosen plans: bash tools/small_m_prof.sh OUT B hw cin cout cfg:split ...
O=gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for p in "${@:5}"; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/p_$p -o r -- python $GRAFT_REPO_ROOT/tools/small_m_one.py $1 $2 $3 $4 $p > /dev/null 2>&1
  echo "== $p"; python - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/$O/p_$p/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if int(r["Calls"])>=40 and ("gemm" in r["Name"] or "conv3x3" in r["Name"] or "splitk" in r["Name"]): print(f'  {float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]}  {r["Name"][:90]}')
PY
done
