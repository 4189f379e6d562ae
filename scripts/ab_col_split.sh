# Column parts of small matcher launches: the default-run stand-in and one-pair calls for each setting (0 = chosen per call)
cd "$GRAFT_REPO_ROOT"
for s in 1 0 2 4 8; do
  SFM_MATCH_COL_SPLIT=$s timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-regimes --no-extensions --no-ransac > gpurun_out/ab_split_$s.json 2> gpurun_out/ab_split_$s.err || exit 1
  python3 - "$s" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_split_{sys.argv[1]}.json"))
p = d["pipeline"]
print("split", sys.argv[1], "pairs/s", d["value"], "stages", p.get("stage_ms"), "one pair", p.get("one_pair_per_call"))
PY
done
