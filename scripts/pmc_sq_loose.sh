# SQ counters of the two-phase sweeps of the loose-threshold path (24 images x 8192, threshold 0.5): matrix pipe and
# vector ALU busy cycles per kernel.  Output: gpurun_out/pmc_sq_loose.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_sq_loose
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq_loose -- python3 bench.py --images 24 --threshold 0.5 --steps 1 --warmup 0 --no-cpu-baseline --no-ransac --no-extensions --no-pipeline --no-regimes > gpurun_out/pmc_sq_loose.log 2>&1
echo "status $?"
python3 scripts/summarise_pmc.py "$(find gpurun_out/pmc_sq_loose -name '*counter_collection.csv' | head -1)" gpurun_out/pmc_sq_loose.json "match_sim_walk_kernel" > /dev/null
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_sq_loose.json"))
for k, v in d.items():
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    v["valu_busy_fraction_of_simd_cycles"] = v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)
    print(k[:70], v["dispatches"], round(v["mean_duration_ns_under_pmc"] / 1e3, 1), "us; matrix pipe busy", round(v.get("mfma_busy_fraction_of_simd_cycles", 0), 3),
          "vector ALU busy", round(v["valu_busy_fraction_of_simd_cycles"], 3), "clock", round(v["clock_GHz_under_load"], 3),
          "vector instructions per MFMA", round(v["SQ_INSTS_VALU"] / max(v["SQ_INSTS_MFMA"], 1) - 1, 2))
json.dump(d, open("gpurun_out/pmc_sq_loose.json", "w"), indent=1)
PY
