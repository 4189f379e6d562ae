#!/bin/bash
# A/B of the RANSAC scoring kernel's pre-test: matrix cores against packed fp32, on the stress shape and on the
# shapes where a model is found.  Output: gpurun_out/ab_ransac_mfma_<tag>_{1,0}.json
tag=${1:-x}
mkdir -p gpurun_out
for v in ${VARIANTS:-1 0}; do
  SFM_RANSAC_MFMA=$v timeout -k 10 400 python bench.py --images 40 --no-cpu-baseline --no-pipeline --no-extensions \
    > gpurun_out/ab_ransac_mfma_${tag}_$v.json 2> gpurun_out/ab_ransac_mfma_${tag}_$v.err || exit 1
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_ransac_mfma_${tag}_$v.json"))
r = d["ransac"]
print("mfma=$v", round(r["value"] / 1e6, 1), "M hyps/s; score kernel ms/step", r.get("score_kernel_ms_per_step"))
for w in r.get("where_a_model_is_found", []):
    print("   ", {k: w[k] for k in w if k in ("inlier_ratio", "hyps_per_s", "pairs_per_s", "filter", "ms")})
PY
done
