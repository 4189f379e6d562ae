# HBM traffic of the similarity kernel: FETCH_SIZE and WRITE_SIZE in two separate counter passes over one step of the
# headline loop (the e2m3 instance, first pass of every pair), summarised per launch and per pair (pairs per launch = 19900 / 11 launches of the step).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout -s KILL 170 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ransac --no-extensions --no-pipeline --no-regimes > gpurun_out/pmc_$c.log 2>&1 || { echo "pass $c failed"; tail -3 gpurun_out/pmc_$c.log; exit 1; }
done
python3 scripts/pmc_traffic.py "$(find gpurun_out/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find gpurun_out/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" "match_sim_kernel<5" $(python3 -c "print(19900 / 11)") gpurun_out/pmc_traffic_f6.json
