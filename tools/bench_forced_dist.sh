#!/bin/bash
# one-rank RCCL run of bench.py (TETRIS_BENCH_FORCE_DIST=1) next to the plain single-rank run, same K:
#   bash tools/bench_forced_dist.sh <out_dir> [steps] [warmup]
OUT=${1:-gpurun_out/dist}; K=${2:-20}; W=${3:-5}
mkdir -p $OUT
X="--no-cpu-baseline --no-extras --steps $K --warmup $W"
for rep in 1 2 3; do
  timeout -k 10 200 python3 bench.py $X > $OUT/single_k${K}_$rep.json 2>/dev/null || exit 1
  TETRIS_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29500 + rep)) \
    timeout -k 10 200 python3 bench.py $X > $OUT/rccl_one_rank_k${K}_$rep.json 2>/dev/null || exit 1
done
python3 - $OUT $K <<'PY'
import json, sys, glob
out, k = sys.argv[1], sys.argv[2]
a = [json.loads([l for l in open(f) if l.startswith("{")][0])["value"] for f in sorted(glob.glob("%s/single_k%s_*.json" % (out, k)))]
b = [json.loads([l for l in open(f) if l.startswith("{")][0]) for f in sorted(glob.glob("%s/rccl_one_rank_k%s_*.json" % (out, k)))]
print("K=%s  single rank: %s G env-steps/s" % (k, " ".join("%.2f" % (x / 1e9) for x in a)))
print("K=%s  one-rank RCCL (payload: %s): %s G env-steps/s" % (k, b[0]["done_gather"]["payload"], " ".join("%.2f" % (x["value"] / 1e9) for x in b)))
print("ratio of medians: %.3f" % (sorted(x["value"] for x in b)[1] / sorted(a)[1]))
PY
