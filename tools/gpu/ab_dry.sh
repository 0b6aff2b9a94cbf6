#!/bin/bash
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras --profile-steps 0"
py='import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d["ms_per_step"]*1e3,2), round(d["config"]["device_ms_per_step"]*1e3,2))'
for i in 1 2 3 4; do
  a=$(BFMMM_DRY_LAUNCH_MS=0 python bench.py $B 2>/dev/null | python -c "$py")
  b=$(BFMMM_DRY_LAUNCH_MS=4 python bench.py $B 2>/dev/null | python -c "$py")
  c=$(BFMMM_DRY_LAUNCH_MS=12 python bench.py $B 2>/dev/null | python -c "$py")
  echo "dry launch once: $a | 4 ms: $b | 12 ms: $c   [us/step wall, device]"
done
