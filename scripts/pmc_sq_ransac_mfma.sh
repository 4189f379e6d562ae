# SQ counters of the RANSAC scoring kernel with the pre-test on the matrix cores (two passes of at most 8 SQ_
# counters), on 64 pairs of the stress shape.  Output: gpurun_out/pmc_sq_ransac_mfma_{a,b}.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
run() { # tag, counters
  rm -rf gpurun_out/pmc_mf_$1; timeout -s KILL 170 rocprofv3 --kernel-trace --pmc $2 GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc_mf_$1 -- python3 bench.py --images 20 --steps 1 --warmup 0 --ransac-pairs 64 --no-cpu-baseline --no-extensions --no-pipeline --no-regimes > gpurun_out/pmc_mf_$1.log 2>&1 || return 1
  python3 scripts/summarise_pmc.py "$(find gpurun_out/pmc_mf_$1 -name '*counter_collection.csv' | head -1)" gpurun_out/pmc_sq_ransac_mfma_$1.json ransac_score > /dev/null
}
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" &&
run b "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" &&
python3 - <<'PY'
import json
for t in "ab":
    d = json.load(open(f"gpurun_out/pmc_sq_ransac_mfma_{t}.json"))
    for k, v in d.items():
        steps = 64 * 64 * 1250  # waves (64 pairs x 4096 rows / 64) x steps of 16 correspondences
        print(k[:60], {x: round(y, 3) for x, y in v.items()})
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA"):
            if c in v: print("   per wave-step:", c, round(v[c] / steps, 2))
PY
