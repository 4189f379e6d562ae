# SQ counters of the RANSAC kernels (one pass; at most 8 SQ_ counters, 2 GRBM_), with kernel durations
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --images 20 --steps 1 --warmup 0 --ransac-pairs 64 --no-cpu-baseline --no-extensions --no-pipeline --no-regimes > gpurun_out/pmc_sq.log 2>&1
echo "status $?"
find gpurun_out/pmc_sq -name "*.csv" | head
python3 scripts/summarise_pmc.py "$(find gpurun_out/pmc_sq -name '*counter_collection.csv' | head -1)" gpurun_out/pmc_sq_ransac.json ransac > /dev/null
python3 - <<'PY'
# per wave and four correspondences: the stress shape has 64 pairs x 4096 rows = 4096 waves x 5000 groups of four
import json
p = "gpurun_out/pmc_sq_ransac.json"
d = json.load(open(p))
for k, v in d.items():
    if "ransac_score_kernel<0, false, true>" in k:
        v["valu_instructions_per_wave_per_four_correspondences"] = v["SQ_INSTS_VALU"] / (4096 * 5000)
        v["salu_instructions_per_wave_per_four_correspondences"] = v["SQ_INSTS_SALU"] / (4096 * 5000)
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        v["valu_busy_fraction_of_simd_cycles"] = v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)
        print(k, {x: v[x] for x in ("dispatches", "mean_duration_ns_under_pmc", "valu_instructions_per_wave_per_four_correspondences", "valu_busy_fraction_of_simd_cycles", "clock_GHz_under_load")})
json.dump(d, open(p, "w"), indent=1)
PY
