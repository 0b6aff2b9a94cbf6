#!/bin/bash
# GPU session: full GPU suite, the suite's pair-Gram-heavy part with k_pair_gram_pack forced, a driver-form bench line
O=gpurun_out/r4; mkdir -p $O
TAG=${1:-1}
python -m pytest tests -m gpu -x -q > $O/t$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 $O/t$TAG.log
BFMMM_PG_PACK=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py tests/test_gpu_shapes.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_chain_batch.py tests/test_gpu_fullsize.py tests/test_gpu_tempered.py -m gpu -x -q > $O/t${TAG}p.log 2>&1; echo "forced-pack rc=$?"; tail -5 $O/t${TAG}p.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20_$TAG.json 2> $O/b20_$TAG.err
python tools/gpu/show_bench.py $O/b20_$TAG.json
