# A/B of library builds on the RANSAC stress shape: kernel durations from a trace per build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  cp sfm_amd/lib/ab/$v.so sfm_amd/lib/libsfm_amd.so
  rm -rf gpurun_out/abr_$v
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abr_$v -- python3 bench.py --images 20 --steps 3 --warmup 1 --no-cpu-baseline --no-extensions --no-pipeline > gpurun_out/abr_$v.json 2>gpurun_out/abr_$v.err || exit 1
  python3 - $v <<'PY'
import csv, glob, sys, json, collections
v=sys.argv[1]
f=glob.glob('gpurun_out/abr_%s/*/*_kernel_trace.csv'%v)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'ransac' in r['Kernel_Name']:
        d[r['Kernel_Name'][:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
j=json.load(open('gpurun_out/abr_%s.json'%v))
print(v, 'hyps/s', j['ransac']['value'], {k: round(sum(x)/len(x),1) for k,x in d.items()})
PY
done
