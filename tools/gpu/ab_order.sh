#!/bin/bash
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras --profile-steps 0"
for i in 1 2 3 4; do
  a=$(python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['ms_per_step']*1e3,2), round(d['config']['device_ms_per_step']*1e3,2))")
  b=$(BFMMM_BENCH_WARMUP_FIRST=1 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['ms_per_step']*1e3,2), round(d['config']['device_ms_per_step']*1e3,2))")
  c=$(BFMMM_BENCH_WARMUP_FIRST=1 BFMMM_DRY_LAUNCH=0 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['ms_per_step']*1e3,2), round(d['config']['device_ms_per_step']*1e3,2))")
  echo "prepare-then-warmup: $a | warmup-then-prepare: $b | warmup-then-prepare, no dry launch (round 3): $c   [us/step wall, device]"
done
