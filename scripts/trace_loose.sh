# Kernel statistics of the matcher at the reference's default threshold (0.5): 24 images x 8192, 276 pairs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/trace_loose
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_loose -- python3 bench.py --images 24 --threshold ${1:-0.5} --steps 2 --warmup 1 --no-cpu-baseline --no-ransac --no-pipeline --no-extensions --no-regimes > gpurun_out/trace_loose.json 2>gpurun_out/trace_loose.err
echo "status $?"
python3 -c "
import json;d=json.load(open('gpurun_out/trace_loose.json'));print(d['value'], d['ms_per_step'], d['config']['matches_rank0'], d['config']['candidates_rank0'])"
python3 scripts/trace_summary.py gpurun_out/trace_loose 12
