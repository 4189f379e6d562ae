# SQ counters of the matcher's similarity kernel, second pass: where its waves wait.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_sq_match_waits
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq_match_waits -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ransac --no-extensions --no-pipeline --no-regimes > gpurun_out/pmc_sq_match_waits.log 2>&1
echo "status $?"
python3 scripts/summarise_pmc.py "$(find gpurun_out/pmc_sq_match_waits -name '*counter_collection.csv' | head -1)" gpurun_out/pmc_sq_match_waits.json "match_sim_kernel<5" | head -40
