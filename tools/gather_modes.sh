#!/bin/bash
OUT=gpurun_out/r03g; mkdir -p $OUT
X="--no-cpu-baseline --no-extras --steps 20 --warmup 5"
for rep in 1 2 3; do
  timeout -k 10 200 python3 bench.py $X 2>/dev/null | grep '^{' > $OUT/single_$rep.json || exit 1
  for mode in side same linkonly; do
    TETRIS_BENCH_GATHER_MODE=$mode TETRIS_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + rep)) \
      timeout -k 10 200 python3 bench.py $X 2>/dev/null | grep '^{' > $OUT/${mode}_$rep.json || exit 1
  done
done
python3 - <<'PY'
import json, glob
for m in ("single", "side", "same", "linkonly"):
    v = [json.load(open(f))["value"] / 1e9 for f in sorted(glob.glob("gpurun_out/r03g/%s_*.json" % m))]
    print("%-9s %s" % (m, " ".join("%.2f" % x for x in v)))
PY
