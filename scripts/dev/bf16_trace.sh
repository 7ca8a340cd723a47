#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktb
timeout 600 rocprofv3 --kernel-trace -d /tmp/ktb -- python $R/bench.py --dtype bf16 --serial --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_bf16_serial.json 2>/dev/null
python $R/scripts/prof_summary.py $(ls /tmp/ktb/*/*.db | head -1) > $out/bf16_serial_kernel_trace.txt
