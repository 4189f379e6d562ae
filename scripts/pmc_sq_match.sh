# SQ counters of the matcher's similarity kernel (one pass: 8 SQ_ counters, 1 GRBM_): issue and busy cycles of the
# matrix pipe and of the vector ALU.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_sq_match
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq_match -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ransac --no-extensions --no-pipeline --no-regimes > gpurun_out/pmc_sq_match.log 2>&1
echo "status $?"
python3 scripts/summarise_pmc.py "$(find gpurun_out/pmc_sq_match -name '*counter_collection.csv' | head -1)" gpurun_out/pmc_sq_match.json "match_sim_kernel<5" | head -40
