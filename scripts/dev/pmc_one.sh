#!/bin/bash
# Dev: SQ stall breakdown + effective clock of ONE convolution shape (forced tile via CN_CFG): scripts/dev/pmc_one.sh "fwd 16 64 64 256 256 3 1"
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
shp="$1"
rm -rf /tmp/p1 /tmp/p2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE -d /tmp/p1 -- python $R/scripts/conv_one.py $shp 10 > /dev/null 2>&1
python $R/scripts/pmc_sq.py /tmp/p1 | head -14
python - <<'PY'
import glob, sqlite3
cur = sqlite3.connect(glob.glob("/tmp/p1/**/*.db", recursive=True)[0]).cursor()
rows = cur.execute("select name, avg(end-start), count(*) from kernels where name like '%igemm%' or name like '%wino%' group by name").fetchall()
for n, d, c in rows:
    print("trace: %-70s avg %.1f us x %d" % (n[:70], d / 1e3, c))
PY
