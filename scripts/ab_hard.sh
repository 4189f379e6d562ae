# Kernel statistics of scripts/probe_hard.py for the libraries in sfm_amd/lib/ab/*.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp sfm_amd/lib/libsfm_amd.so /tmp/keep.so
for f in sfm_amd/lib/ab/*.so; do
  v=$(basename $f .so)
  cp $f sfm_amd/lib/libsfm_amd.so
  python3 scripts/probe_hard.py
  rm -rf gpurun_out/hard_$v
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hard_$v -- python3 scripts/probe_hard.py > gpurun_out/hard_$v.txt 2>&1 || exit 1
  echo "== $v"; python3 scripts/trace_summary.py gpurun_out/hard_$v 8
done
cp /tmp/keep.so sfm_amd/lib/libsfm_amd.so
