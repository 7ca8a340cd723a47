#!/bin/bash
# kernel trace of ONE step function replayed as a HIP graph: busy fraction, gaps, top kernels.  usage: step_trace.sh d|sd|ld|g
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st
CN_USE_GRAPHS=1 rocprofv3 --kernel-trace -d /tmp/st -o st -- python $GRAFT_REPO_ROOT/scripts/run_step.py $1 12 > /tmp/st.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(ls /tmp/st/*/*.db /tmp/st/*.db 2>/dev/null | head -1)
python scripts/gap_analysis.py $DB ${2:-100} | head -14
python scripts/prof_summary.py $DB | head -28
