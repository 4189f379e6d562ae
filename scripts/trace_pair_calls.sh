# Kernel and copy timeline of the single-pair entry points (one pair per call, host buffers in and out).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# usage: trace_pair_calls.sh [match|ransac ...] ; PROBE_ARG=pinned keeps the matcher's tables in page-locked memory
python3 scripts/probe_match_pair_once.py
python3 scripts/probe_match_pair_once.py pinned
python3 scripts/probe_ransac_pair_once.py
for w in ${@:-match ransac}; do
  rm -rf gpurun_out/trace_pair_$w
  timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_pair_$w -- python3 scripts/probe_${w}_pair_once.py $PROBE_ARG > gpurun_out/trace_pair_$w.txt 2>&1 || exit 1
  python3 - $w <<'PY'
import csv, glob, sys
w = sys.argv[1]
ev = []
for f in glob.glob(f'gpurun_out/trace_pair_{w}/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
for f in glob.glob(f'gpurun_out/trace_pair_{w}/*/*_memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', r.get('Kind', ''))))
ev.sort()
# a call begins with the first event after a device-to-host copy and ends with its last device-to-host copy
starts = [i for i, e in enumerate(ev) if 'DEVICE_TO_HOST' not in e[2] and i > 0 and 'DEVICE_TO_HOST' in ev[i - 1][2]]
call = ev[starts[-1]:]
period_from = ev[starts[-2]][0]
t0 = call[0][0]
print(w, 'last call: span', (call[-1][1] - t0) / 1e3, 'us; busy', sum(e - s for s, e, _ in call) / 1e3, 'us;', len(call), 'events; period', (call[0][0] - period_from) / 1e3)
for s, e, n in call:
    print(f'  {(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} us  {n}')
PY
done
