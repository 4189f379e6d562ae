# Kernel timeline of the RANSAC stress shape only (bench.py without the matcher's secondary legs): totals per kernel
# and what one timed batch consists of.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/trace_ransac
timeout -s KILL 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/trace_ransac -- python3 bench.py --steps 2 --warmup 1 --images 8 --no-cpu-baseline --no-pipeline --no-extensions --no-regimes > gpurun_out/trace_ransac.json 2>gpurun_out/trace_ransac.err
echo "status $?"
python3 scripts/trace_summary.py gpurun_out/trace_ransac 12
python3 - <<'PY'
import csv, glob, json
d = json.load(open('gpurun_out/trace_ransac.json'))
print('hyps/s', d['ransac']['value'], 'score kernel ms per step', d['ransac']['score_kernel_ms_per_step'])
ev = []
for f in glob.glob('gpurun_out/trace_ransac/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
for f in glob.glob('gpurun_out/trace_ransac/*/*_memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
ev.sort()
score = [i for i, e in enumerate(ev) if 'ransac_score_kernel' in e[2] and e[1] - e[0] > 20_000_000]
# the last batch: from the prep kernel before its first scoring launch to the last inlier kernel after it
last3 = score[-3:]
a = max(i for i in range(last3[0]) if 'ransac_prep_kernel' in ev[i][2])
b = min(len(ev) - 1, last3[-1] + 12)
t0 = ev[a][0]
busy = 0
for s, e, n in ev[a:b + 1]:
    busy += e - s
    print(f'  {(s - t0) / 1e3:10.1f} +{(e - s) / 1e3:10.1f}  {n}')
print('span ms', (ev[b][1] - t0) / 1e6, 'busy ms', busy / 1e6)
PY
