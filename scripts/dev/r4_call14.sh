#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4c14; mkdir -p $O
CN_DETERMINISTIC=1 timeout 1800 python -m pytest tests/test_nets_gpu.py tests/test_ops_gpu.py tests/test_steps_gpu.py -q -m gpu > $O/det_suite.txt 2>&1; tail -6 $O/det_suite.txt
timeout 600 python -m pytest tests/test_steps_gpu.py tests/test_bf16_gpu.py -q -m gpu -k "latent_gan or data_parallel" > $O/new_tests.txt 2>&1; tail -4 $O/new_tests.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
