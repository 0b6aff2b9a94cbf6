#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_prepare.py tests/test_gpu_chain_batch.py tests/test_gpu_disk_batches.py tests/test_gpu_tempered.py -m gpu -x -q > $O/t3.log 2>&1; echo "prepare rc=$?"; tail -6 $O/t3.log
python tools/gpu/trace_run.py 2>&1 | grep -v bfmmm_run | tail -11
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20_3.json 2> $O/b20_3.err; python tools/gpu/show_bench.py $O/b20_3.json
