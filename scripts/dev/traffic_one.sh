#!/bin/bash
# Dev: HBM-side traffic (PMC FETCH_SIZE x2 + WRITE_SIZE) of single convolution shapes against their algorithmic bytes.
cd /tmp; export TMPDIR=/tmp; R=/root/repo
for shp in "fwd 16 64 64 256 256 3 1" "wgrad 16 64 64 256 256 3 1" "dgrad 16 64 64 256 256 3 1" "fwd 16 128 128 48 96 3 2" "dgrad 16 128 128 48 96 3 2" "wgrad 16 128 128 48 96 3 2" "fwd 16 256 256 64 64 3 1"; do
  tag=$(echo $shp | tr " " "_")
  rocprofv3 --pmc FETCH_SIZE -d /tmp/tf_$tag -- python $R/scripts/conv_one.py $shp 5 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d /tmp/tw_$tag -- python $R/scripts/conv_one.py $shp 5 > /dev/null 2>&1
  python - "$shp" /tmp/tf_$tag /tmp/tw_$tag <<'PY'
import sys, glob, sqlite3
shp, fd, wd = sys.argv[1], sys.argv[2], sys.argv[3]
def load(d, c):
    cur = sqlite3.connect(glob.glob(d + "/**/*.db", recursive=True)[0]).cursor()
    rows = [(n, v) for n, v in cur.execute("select kernel_name, value from counters_collection where counter_name=?", (c,)) if "igemm" in n or "s2_image" in n or "c3_fwd" in n]
    return sum(v for _, v in rows) * 1024 / max(len(rows), 1)
kind, n, h, w, cin, cout, k, s = shp.split()
n, h, w, cin, cout, k, s = map(int, (n, h, w, cin, cout, k, s))
xin, yout, wb = n * h * w * cin * 4, n * (h // s) * (w // s) * cout * 4, k * k * cin * cout * 4
alg = {"fwd": xin + wb + yout, "dgrad": yout + wb + xin, "wgrad": xin + yout + wb}[kind]
f, wr = 2 * load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
print("%-32s algorithmic %7.1f MB | fetch(x2) %7.1f MB  write %7.1f MB  total %7.1f MB  = %.2fx" % (shp, alg / 1e6, f / 1e6, wr / 1e6, (f + wr) / 1e6, (f + wr) / alg))
PY
done
