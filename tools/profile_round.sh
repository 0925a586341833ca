#!/bin/bash
# Collects the evidence committed under profiles/ for one round, on the GPU box:
#   gpurun --timeout 1150 -- 'bash tools/profile_round.sh r03'
# then copy gpurun_out/<tag>/ into profiles/ (file names get the tag as prefix).
# Counters are taken in their own passes (never combined with trace domains).
set -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
J() { grep '^{' ; }   # (RCCL / driver banners may precede the JSON line)

# 1. bench lines (N = 1): config 5 shape, 7-piece set, without obs, fused, two streams
X="--no-cpu-baseline --no-extras"
timeout -k 10 200 python3 bench.py $X --rows 40 2>/dev/null | J > $OUT/bench_n1_10x40.json || exit 1
timeout -k 10 200 python3 bench.py $X --pieces standard7 2>/dev/null | J > $OUT/bench_n1_10x20_standard7.json || exit 1
timeout -k 10 200 python3 bench.py $X --no-obs 2>/dev/null | J > $OUT/bench_n1_10x20_no_obs.json || exit 1
timeout -k 10 200 python3 bench.py $X --fuse 20 --steps 1000 --warmup 100 2>/dev/null | J > $OUT/bench_n1_10x20_fuse20.json || exit 1
timeout -k 10 200 python3 bench.py $X --streams 2 2>/dev/null | J > $OUT/bench_n1_10x20_streams2.json || exit 1
timeout -k 10 200 python3 bench.py $X --streams 2 --rows 40 2>/dev/null | J > $OUT/bench_n1_10x40_streams2.json || exit 1
echo "bench done"

# 2. kernel trace of the same bench command (per-kernel average duration)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 50 > $OUT/trace.log 2>&1 || exit 1
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/step_kernel_stats_bench_steps300.csv
echo "trace done"

# 3. HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes, calibrated on refresh_kernel
timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_probe.py > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_probe.py > $OUT/pmc_write.log 2>&1 || exit 1
python3 tools/parse_pmc.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json || exit 1
# the headline lines last: they quote the traffic just measured (same library: source hash)
mkdir -p profiles && cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 300 python3 bench.py 2> $OUT/bench_n1_10x20.err | J > $OUT/bench_n1_10x20.json || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 2>/dev/null | J > $OUT/bench_n1_10x20_driver_style_steps20.json || exit 1
echo "pmc traffic + headline bench done"

# 4. per-wave SQ counters of the step kernel (two passes of 8 counters)
timeout -k 10 250 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq1 -- python3 tools/pmc_probe.py > $OUT/pmc_sq1.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq2 -- python3 tools/pmc_probe.py > $OUT/pmc_sq2.log 2>&1 || exit 1
python3 tools/parse_pmc_variant.py $OUT/pmc_sq1 $OUT/pmc_sq2 > $OUT/step_kernel_sq_counters_per_wave.txt || exit 1
echo "sq counters done"

# 5. afterstate kernels: 10x20 and 10x40, timings, kernel trace, SQ counters
for r in 20 40; do
  ABL_ROWS=$r timeout -k 10 200 python3 tools/bench_afterstates.py 2>/dev/null | J > $OUT/afterstates_10x$r.json || exit 1
  ABL_ROWS=$r timeout -k 10 300 python3 tools/bench_policy.py 2>/dev/null | J > $OUT/policy_kernels_10x$r.json || exit 1
done
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_after -- python3 tools/bench_afterstates.py > $OUT/trace_after.log 2>&1 || exit 1
cp $(ls $OUT/trace_after/*/*kernel_stats.csv | head -1) $OUT/afterstates_kernel_stats.csv
ABL_ROWS=40 timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_after40 -- python3 tools/bench_afterstates.py > $OUT/trace_after40.log 2>&1 || exit 1
cp $(ls $OUT/trace_after40/*/*kernel_stats.csv | head -1) $OUT/afterstates_kernel_stats_10x40.csv
timeout -k 10 250 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_after_sq1 -- python3 tools/pmc_probe_after.py > $OUT/pmc_after_sq1.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_after_sq2 -- python3 tools/pmc_probe_after.py > $OUT/pmc_after_sq2.log 2>&1 || exit 1
python3 tools/parse_pmc_kernels.py afterstates_kernel 16384 $OUT/pmc_after_sq1 $OUT/pmc_after_sq2 > $OUT/afterstates_sq_counters_per_wave.txt || exit 1
python3 tools/parse_pmc_kernels.py greedy_kernel 16384 $OUT/pmc_after_sq1 $OUT/pmc_after_sq2 >> $OUT/afterstates_sq_counters_per_wave.txt || exit 1
if [ -f build_variants/libtetris_a3.so ]; then TETRIS_VARIANT_LIB=build_variants/libtetris_a3.so timeout -k 10 100 python3 tools/bench_afterstates.py 2>/dev/null | J > $OUT/afterstates_10x20_3waves_variant.json; fi
echo "afterstates done"

# 6. micro-benchmarks, sweeps, host overhead, the example
timeout -k 10 100 python3 tools/ubench/fill_bw.py > $OUT/ubench_fill_bw.txt 2>/dev/null || exit 1
if [ -x build_variants/row_store ]; then timeout -k 10 60 ./build_variants/row_store > $OUT/ubench_row_store.txt 2>/dev/null || exit 1; fi
if [ -x build_variants/step_traffic ]; then timeout -k 10 200 ./build_variants/step_traffic > $OUT/ubench_step_traffic.txt 2>/dev/null || exit 1; fi
timeout -k 10 100 python3 tools/sweep_batch.py 16384 65536 131072 262144 524288 1048576 2097152 4194304 2>/dev/null | grep '^B=' > $OUT/sweep_batch.txt || exit 1
timeout -k 10 100 python3 tools/launch_overhead.py > $OUT/host_launch_overhead.txt 2>/dev/null || exit 1
timeout -k 10 100 python3 examples/example_play.py > $OUT/example_play.txt 2>/dev/null || exit 1
echo "ubench + sweeps done"

# 7. the N > 1 code path on this one GPU: one-rank RCCL run next to the plain run, K = 20 and K = 1000
bash tools/bench_forced_dist.sh $OUT/dist 20 5 > $OUT/rccl_one_rank_vs_single_k20.txt 2>&1 || exit 1
bash tools/bench_forced_dist.sh $OUT/dist 1000 100 > $OUT/rccl_one_rank_vs_single_k1000.txt 2>&1 || exit 1
cp $OUT/dist/rccl_one_rank_k20_2.json $OUT/bench_n1_rccl_one_rank_forced_collectives_steps20.json
cp $OUT/dist/rccl_one_rank_k1000_2.json $OUT/bench_n1_rccl_one_rank_forced_collectives.json
echo "forced dist done"
ls -la $OUT
