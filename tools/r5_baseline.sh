# round-5 start-of-round state on one box: GPU tests, default bench line, per-step kernel breakdown, FETCH_SIZE / WRITE_SIZE passes
bash tools/final_validation.sh r5_base
bash tools/step_breakdown.sh r5_base_brk
bash tools/r5_pmc.sh r5_base_pmc FETCH_SIZE WRITE_SIZE
