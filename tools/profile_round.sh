#!/bin/bash
# Collects the evidence committed under profiles/ for one round, on the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r01f'
# then copy gpurun_out/<tag>/ into profiles/ (see the cp lines printed at the end).
# Counters are taken in their own passes (never combined with trace domains).
set -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"

# 1. bench lines (N = 1): headline, config 5 shape, 7-piece set, without obs, fused
X="--no-cpu-baseline --no-extras"
timeout -k 10 200 python3 bench.py $X --rows 40 > $OUT/bench_n1_10x40.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py $X --pieces standard7 > $OUT/bench_n1_10x20_standard7.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py $X --no-obs > $OUT/bench_n1_10x20_no_obs.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py $X --fuse 20 --steps 1000 --warmup 100 > $OUT/bench_n1_10x20_fuse20.json 2>/dev/null || exit 1
# two env shards per GPU on two HIP streams (the tail of one launch overlaps the ramp of the other)
timeout -k 10 200 python3 bench.py $X --streams 2 > $OUT/bench_n1_10x20_streams2.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py $X --streams 2 --rows 40 > $OUT/bench_n1_10x40_streams2.json 2>/dev/null || exit 1
# N = 2 ranks rehearsed on this ONE GPU (gloo moves the gathers through the host): checks the launcher and
# the N > 1 line; two ranks time-slice the device, so the rate is NOT a scaling number
TETRIS_BENCH_BACKEND=gloo timeout -k 10 200 python3 bench.py --gpus 2 --batch 524288 --steps 200 --warmup 50 > $OUT/bench_n2_one_gpu_rehearsal_gloo.json 2>/dev/null || exit 1
echo "bench done"

# 2. kernel trace of the same bench command (per-kernel average duration)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 50 > $OUT/trace.log 2>&1 || exit 1
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/step_kernel_stats_bench_steps300.csv
echo "trace done"

# 3. HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes, calibrated on refresh_kernel
timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_probe.py > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_probe.py > $OUT/pmc_write.log 2>&1 || exit 1
python3 tools/parse_pmc.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json || exit 1
# the headline line last: it quotes the traffic just measured (same sources: csrc_hash)
mkdir -p profiles && cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 300 python3 bench.py > $OUT/bench_n1_10x20.json 2> $OUT/bench_n1_10x20.err || exit 1
echo "pmc traffic + headline bench done"

# 4. per-wave SQ counters of the step kernel (two passes of 8 counters)
timeout -k 10 250 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq1 -- python3 tools/pmc_probe.py > $OUT/pmc_sq1.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq2 -- python3 tools/pmc_probe.py > $OUT/pmc_sq2.log 2>&1 || exit 1
python3 tools/parse_pmc_variant.py $OUT/pmc_sq1 $OUT/pmc_sq2 > $OUT/step_kernel_sq_counters_per_wave.txt || exit 1
echo "sq counters done"

# 5. afterstates kernel
timeout -k 10 200 python3 tools/bench_afterstates.py > $OUT/afterstates_10x20.json 2>/dev/null || exit 1
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_after -- python3 tools/bench_afterstates.py > $OUT/trace_after.log 2>&1 || exit 1
cp $(ls $OUT/trace_after/*/*kernel_stats.csv | head -1) $OUT/afterstates_kernel_stats.csv
echo "afterstates done"

# 6. policy-side kernels (greedy, fused greedy step, rollouts) and the store-path microbenchmarks
timeout -k 10 300 python3 tools/bench_policy.py > $OUT/policy_kernels_10x20.json 2>/dev/null || exit 1
timeout -k 10 100 python3 tools/ubench/fill_bw.py > $OUT/ubench_fill_bw.txt 2>/dev/null || exit 1
if [ -x build_variants/row_store ]; then timeout -k 10 60 ./build_variants/row_store > $OUT/ubench_row_store.txt 2>/dev/null || exit 1; fi
if [ -x build_variants/step_traffic ]; then timeout -k 10 120 ./build_variants/step_traffic > $OUT/ubench_step_traffic.txt 2>/dev/null || exit 1; fi
timeout -k 10 100 python3 tools/sweep_batch.py 65536 131072 262144 524288 1048576 2097152 4194304 > $OUT/sweep_batch.txt 2>/dev/null || exit 1
timeout -k 10 100 python3 tools/launch_overhead.py > $OUT/host_launch_overhead.txt 2>/dev/null || exit 1
timeout -k 10 100 python3 examples/example_play.py > $OUT/example_play.txt 2>/dev/null || exit 1
if [ -f build_variants/libtetris_stamps.so ]; then timeout -k 10 100 python3 tools/timeline.py > $OUT/step_kernel_timeline.txt 2>/dev/null || exit 1; fi
echo "policy + ubench done"
ls -la $OUT
