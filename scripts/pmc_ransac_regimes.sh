# SQ counters of the RANSAC scoring kernels where a model is found: $1 = inlier ratio, $2 = pre-test (1/0), $3 = tag.
# Two passes of at most 8 SQ_ counters on 128 pairs of 20 000 correspondences x 4096 rows.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
run() { # tag, counters
  rm -rf gpurun_out/pmc_rr_$1; timeout -s KILL 170 rocprofv3 --kernel-trace --pmc $2 GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc_rr_$1 -- python3 scripts/probe_ransac_regimes.py $RATIO 128 $FILT 0 1 > gpurun_out/pmc_rr_$1.log 2>&1 || return 1
  python3 scripts/summarise_pmc.py "$(find gpurun_out/pmc_rr_$1 -name '*counter_collection.csv' | head -1)" gpurun_out/pmc_ransac_regime_$1.json ransac_score > /dev/null
}
RATIO=$1; FILT=$2; TAG=$3
run ${TAG}_a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" &&
run ${TAG}_b "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" &&
python3 - <<PY
import json
for t in "ab":
    d = json.load(open("gpurun_out/pmc_ransac_regime_${TAG}_%s.json" % t))
    for k, v in d.items():
        print(k[:70], {x: round(y, 3) for x, y in v.items()})
PY
