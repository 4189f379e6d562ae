for fs in ${FS_LIST:-0 256 1024}; do
  timeout -k 10 120 python bench.py --images 40 --steps 3 --warmup 1 --no-cpu-baseline --first-slice $fs > gpurun_out/fs_$fs.json 2> gpurun_out/fs_$fs.err || exit 1
  python3 -c "
import json;d=json.load(open('gpurun_out/fs_$fs.json'));print($fs, d['ransac']['value'], d['ransac']['score_kernel_ms_per_step'], d['pipeline']['ms'])"
done
