# Per-kernel durations of the RANSAC stress shape (1024 pairs) for variants of the matrix-core pre-test:
# prepared operands (default), operands made by the blocks (SFM_RANSAC_MFMA_OPS_GIB=0), packed fp32 test (SFM_RANSAC_MFMA=0).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in ${VARIANTS:-1:64 1:0 0:64}; do
  set -- ${v/:/ }
  tag=m$1_g$2
  rm -rf gpurun_out/trab_$tag
  SFM_RANSAC_MFMA=$1 SFM_RANSAC_MFMA_OPS_GIB=$2 timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trab_$tag -- python3 bench.py --steps 2 --warmup 1 --images 8 --ransac-pairs 1024 --no-cpu-baseline --no-pipeline --no-extensions --no-regimes > gpurun_out/trab_$tag.json 2>gpurun_out/trab_$tag.err || exit 1
  echo "== mfma=$1 ops_gib=$2"
  python3 - <<PY
import csv, glob, json
d = json.load(open('gpurun_out/trab_$tag.json'))
print('hyps/s', d['ransac']['value'], 'score kernel ms per step', d['ransac']['score_kernel_ms_per_step'])
for f in glob.glob('gpurun_out/trab_$tag/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'ransac' in r['Name'] and float(r['Percentage']) > 0.3:
            print('   %-70s calls %5s avg %10.1f us total %9.2f ms' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done
