# SQ counters of the RANSAC kernels (one pass; at most 8 SQ_ counters, 2 GRBM_), with kernel durations
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --images 20 --steps 1 --warmup 0 --ransac-pairs 64 --no-cpu-baseline --no-extensions --no-pipeline > gpurun_out/pmc_sq.log 2>&1
echo "status $?"
find gpurun_out/pmc_sq -name "*.csv" | head
