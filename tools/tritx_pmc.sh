# matrix-pipe / wave counters of the tri-plane transformer's kernels (tools/tritx_time.py under rocprofv3 --pmc, SQ counters only)
O=gpurun_out/${1:-tritx_pmc}; mkdir -p $O
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tx_pmc; timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/tx_pmc -- python $R/tools/tritx_time.py > $R/$O/time.txt 2>&1
python $R/tools/pmc_table.py /tmp/tx_pmc | grep -E "tx_attn_(fwd|bwd)|gemm_f16" > $R/$O/tritx_sq_counters.txt
python - "$R/$O/tritx_sq_counters.txt" <<'PY' | tee $R/$O/tritx_sq_summary.txt
import sys, re, collections
rows = collections.defaultdict(dict)
for ln in open(sys.argv[1]):
    m = re.match(r"(.{60}) grid\s+(\d+) (\S+)\s+([\d.]+)", ln)
    if m: rows[(m.group(1).strip()[:48], m.group(2))][m.group(3)] = float(m.group(4))
for k, v in rows.items():
    busy, mf = v.get("SQ_BUSY_CU_CYCLES", 0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    wc = v.get("SQ_WAVE_CYCLES", 0)
    print(f"{k[0]:48s} grid {k[1]:>8s}  matrix pipe busy per SIMD (SQ_VALU_MFMA_BUSY_CYCLES / 4 SQ_BUSY_CU_CYCLES) {mf / (4 * busy) if busy else 0:5.3f}  wait_any/wave {v.get('SQ_WAIT_ANY', 0) / wc if wc else 0:5.3f}  wait_inst/wave {v.get('SQ_WAIT_INST_ANY', 0) / wc if wc else 0:5.3f}  valu/wave {v.get('SQ_ACTIVE_INST_VALU', 0) / wc if wc else 0:5.3f}  lds/wave {v.get('SQ_ACTIVE_INST_LDS', 0) / wc if wc else 0:5.3f}")
PY
