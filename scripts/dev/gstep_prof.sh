#!/bin/bash
# kernel composition of the generator step (eager, one stream) under rocprofv3: bash scripts/dev/gstep_prof.sh OUTDIR
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/gt -o run -- python $root/scripts/g_step_trace.py 20 > $out/gstep.log 2>&1
db=$(ls /tmp/gt/*/*.db /tmp/gt/*.db 2>/dev/null | tail -1)
python $root/scripts/prof_summary.py $db > $out/gstep_kernels.txt 2>&1
