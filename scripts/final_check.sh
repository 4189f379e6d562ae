# Whole GPU suite, smoke, the default bench line, and the same bench under rocprofv3 (kernel stats as csv)
TAG=${1:-final}
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/${TAG}_tests.txt 2>&1
st=$?; echo "tests status $st"; tail -3 gpurun_out/${TAG}_tests.txt; [ $st -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.txt 2>&1 || { tail -5 gpurun_out/${TAG}_smoke.txt; exit 1; }
tail -1 gpurun_out/${TAG}_smoke.txt
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_line.json 2>gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/${TAG}_bench_line.json'));print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'], d.get('ransac',{}).get('value'), d.get('pipeline',{}).get('ms'))"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_line_under_rocprof.json 2>gpurun_out/${TAG}_prof.err
echo "rocprof status $?"
find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -2
