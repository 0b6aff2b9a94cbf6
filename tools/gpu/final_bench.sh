#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/gpu/show_bench.py $O/bench_default.json
python bench.py --steps 20 --warmup 5 > $O/bench_driver20.json 2> $O/bench_driver20.err; python tools/gpu/show_bench.py $O/bench_driver20.json
