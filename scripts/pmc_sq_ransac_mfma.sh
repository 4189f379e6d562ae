# SQ counters of the RANSAC scoring kernel with the pre-test on the matrix cores (two passes of at most 8 SQ_
# counters), on 1024 pairs of the stress shape (slices of 256 and 3840 rows: whole rounds of blocks, as in the
# benchmark's batch of 10 000).  Output: gpurun_out/pmc_sq_ransac_mfma_{a,b}.json and their merge, gpurun_out/pmc_sq_ransac_mfma.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
run() { # tag, counters
  rm -rf gpurun_out/pmc_mf_$1; timeout -s KILL 170 rocprofv3 --kernel-trace --pmc $2 GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc_mf_$1 -- python3 bench.py --images 20 --steps 1 --warmup 0 --ransac-pairs 1024 --no-cpu-baseline --no-extensions --no-pipeline --no-regimes > gpurun_out/pmc_mf_$1.log 2>&1 || return 1
  python3 scripts/summarise_pmc.py "$(find gpurun_out/pmc_mf_$1 -name '*counter_collection.csv' | head -1)" gpurun_out/pmc_sq_ransac_mfma_$1.json ransac_score > /dev/null
}
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" &&
run b "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" &&
python3 - <<'PY'
import json
a = json.load(open("gpurun_out/pmc_sq_ransac_mfma_a.json"))
b = json.load(open("gpurun_out/pmc_sq_ransac_mfma_b.json"))
out = {}
for k, v in a.items():
    if "ransac_score_mfma_kernel" not in k:
        continue
    m = dict(v)
    m.update({kk: vv for kk, vv in b.get(k, {}).items() if kk not in m})
    wsd = m["SQ_WAVES"] * 1250  # mean waves per dispatch x steps of 16 correspondences (20 000 per pair)
    den = m["GRBM_GUI_ACTIVE"] * 128  # SIMD cycles per XCD while the kernel runs
    m["shape"] = ("1024 pairs x 20000 correspondences x 4096 rows at 15 % inliers, slices of 256 and 3840 rows per call "
                  "(the figures are means per dispatch); one wave-step = 16 correspondences x 64 rows")
    m["wave_steps_per_dispatch"] = int(wsd)
    for name, c in (("vector", "SQ_INSTS_VALU"), ("mfma", "SQ_INSTS_MFMA"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS")):
        m[name + "_instructions_per_wave_step"] = m[c] / wsd
    m["valu_busy_fraction_of_simd_cycles"] = m["SQ_ACTIVE_INST_VALU"] * 4 / den
    m["mfma_busy_fraction_of_simd_cycles"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / den
    out[k] = m
    print(k, {x: round(m[x], 4) for x in ("vector_instructions_per_wave_step", "mfma_instructions_per_wave_step",
                                          "valu_busy_fraction_of_simd_cycles", "mfma_busy_fraction_of_simd_cycles",
                                          "clock_GHz_under_load", "mean_duration_ns_under_pmc", "dispatches")})
json.dump(out, open("gpurun_out/pmc_sq_ransac_mfma.json", "w"), indent=1)
PY
