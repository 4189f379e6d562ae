# Headline bench against the matcher's chunk size
for c in 2048 1024 512 2048 1024 256; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --pairs-per-launch $c --no-cpu-baseline --no-ransac --no-pipeline --no-extensions > gpurun_out/cs_$c.json 2>gpurun_out/cs_$c.err || exit 1
  python3 -c "
import json;d=json.load(open('gpurun_out/cs_$c.json'));print($c, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], d['roofline']['frac'])"
done
