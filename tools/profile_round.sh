#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the default bench command (graph replay, single-chain headline only)
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) on a short run, as MI355X_MICROARCH.md prescribes
#   3. one --pmc pass with the matrix-core counters (MFMA instruction / busy cycles) for k_pair_gram / k_factor
#   4. --kernel-trace --stats of the 8-chain batches (warm start and Nu_Z: BASELINE configs[4]) and of configs[2], [3]
# Outputs land in gpurun_out/prof_<tag>/; tools/summarize_profile.py condenses them into profiles/.
set -uo pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --profile-steps 0"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o run -- python3 "$ROOT/bench.py" $B > "$OUT/bench_trace.log" 2>&1 && echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o run -- python3 "$ROOT/bench.py" --steps 100 --warmup 20 $B > "$OUT/bench_fetch.log" 2>&1 && echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o run -- python3 "$ROOT/bench.py" --steps 100 --warmup 20 $B > "$OUT/bench_write.log" 2>&1 && echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -d "$OUT/pmc_mfma" -o run -- python3 "$ROOT/bench.py" --steps 100 --warmup 20 $B > "$OUT/bench_mfma.log" 2>&1 && echo "mfma done" || { echo "mfma pass failed:"; tail -5 "$OUT/bench_mfma.log"; }
# (chain batches normally run as two half-batches on two streams; the per-kernel averages below are taken with the batch on ONE
#  stream, BFMMM_BATCH_SPLIT=1, so that a kernel's duration is not stretched by the other half's kernels sharing the CUs)
export BFMMM_BATCH_SPLIT=1
for spec in warm:8 warm:32 nu_z:1 nu_z:8 config3:1 config4:1; do
  wl=${spec%%:*}; ch=${spec##*:}
  rocprofv3 --kernel-trace --stats -d "$OUT/${wl}_${ch}" -o run -- python3 "$ROOT/tools/prof_workload.py" --workload "$wl" --chains "$ch" --steps 300 > "$OUT/${wl}_${ch}.json" 2> "$OUT/${wl}_${ch}.err" && echo "$spec done: $(cat "$OUT/${wl}_${ch}.json")"
done
# FETCH_SIZE / WRITE_SIZE of the 8-chain Nu_Z batch (BASELINE configs[4] on one GPU): bench.py's config5.roofline.traffic
unset BFMMM_BATCH_SPLIT
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/nu_z_8_pmc_fetch" -o run -- python3 "$ROOT/tools/prof_workload.py" --workload nu_z --chains 8 --steps 100 > "$OUT/nu_z_8_fetch.log" 2>&1 && echo "nu_z:8 fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/nu_z_8_pmc_write" -o run -- python3 "$ROOT/tools/prof_workload.py" --workload nu_z --chains 8 --steps 100 > "$OUT/nu_z_8_write.log" 2>&1 && echo "nu_z:8 write done"
# FETCH_SIZE / WRITE_SIZE and the matrix-core counters of the 8-chain WARM-START batch on one stream (k_pair_gram_pack and the
# per-curve kernels at their batch sizes)
export BFMMM_BATCH_SPLIT=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/warm_8_pmc_fetch" -o run -- python3 "$ROOT/tools/prof_workload.py" --workload warm --chains 8 --steps 100 > "$OUT/warm_8_fetch.log" 2>&1 && echo "warm:8 fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/warm_8_pmc_write" -o run -- python3 "$ROOT/tools/prof_workload.py" --workload warm --chains 8 --steps 100 > "$OUT/warm_8_write.log" 2>&1 && echo "warm:8 write done"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -d "$OUT/warm_8_pmc_mfma" -o run -- python3 "$ROOT/tools/prof_workload.py" --workload warm --chains 8 --steps 100 > "$OUT/warm_8_mfma.log" 2>&1 && echo "warm:8 mfma done"
unset BFMMM_BATCH_SPLIT
# condensed summaries (small files): gpurun_out/profiles_$TAG/ -> copied to profiles/ by hand
python3 "$ROOT/tools/summarize_profile.py" "$OUT" "${TAG}_final" "$ROOT/gpurun_out/profiles_$TAG" > "$OUT/summary.log" 2>&1; tail -30 "$OUT/summary.log"
# the databases themselves are large: only the summaries travel back
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*.csv" -size +2M -delete
ls "$ROOT/gpurun_out/profiles_$TAG"
