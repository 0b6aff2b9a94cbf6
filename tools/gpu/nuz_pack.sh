#!/bin/bash
for wl in warm nu_z; do for ch in 2 4 6; do for pk in 0 1; do
  echo -n "$wl chains=$ch PACK=$pk: "; BFMMM_PG_PACK=$pk python tools/prof_workload.py --workload $wl --chains $ch --steps 300 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['chain_iterations_per_s']))"
done; done; done
