# Kernel and copy timeline of the headline matcher loop: what runs between two launches of the similarity kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/trace_headline
timeout -s KILL 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_headline -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ransac --no-pipeline --no-extensions --no-regimes > gpurun_out/trace_headline.json 2>gpurun_out/trace_headline.err
echo "status $?"
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob('gpurun_out/trace_headline/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70], r.get('Stream_Id', r.get('Queue_Id', ''))))
for f in glob.glob('gpurun_out/trace_headline/*/*_memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', ''), ''))
ev.sort()
sims = [i for i, e in enumerate(ev) if 'match_sim_kernel<5, 1, false, false, 0>' in e[2] and e[1] - e[0] > 2_000_000]
gaps = [(ev[sims[k + 1]][0] - ev[sims[k]][1]) / 1e3 for k in range(len(sims) - 1)]
print('full-size launches', len(sims), 'mean duration us', sum(ev[i][1] - ev[i][0] for i in sims) / len(sims) / 1e3)
print('gaps between consecutive launches (us):', ' '.join(f'{g:.0f}' for g in gaps[-22:]))
# everything between the last two full-size launches of the run
a, b = sims[-3], sims[-2]
t0 = ev[a][1]
print('between two launches (us after the first ends; start, duration, queue):')
for s, e, n, q in ev[a + 1:b + 1]:
    print(f'  {(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  {q:>4s}  {n}')
PY
