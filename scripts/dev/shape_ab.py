"""Per-shape A/B of two scripts/conv_shapes_bench.py outputs: python scripts/dev/shape_ab.py base.txt new.txt [kinds]"""
import sys


def load(p):
    rows = {}
    for l in open(p):
        t = l.split()
        if len(t) >= 10 and t[0] in ("fwd", "dgrad", "wgrad"):
            key = (t[0], t[4], t[5], t[6], " ".join(t[7:-1]))
            rows[key] = (int(t[1]), float(t[2]), float(t[3]))
    return rows


a, b = load(sys.argv[1]), load(sys.argv[2])
kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fwd", "dgrad"]
tot_a = tot_b = 0.0
out = []
for k, (cnt, us, tf) in a.items():
    if k[0] not in kinds or k not in b:
        continue
    us2 = b[k][1]
    tot_a += cnt * us
    tot_b += cnt * us2
    out.append((cnt * (us - us2), k, cnt, us, us2))
for d, k, cnt, us, us2 in sorted(out, reverse=True):
    if abs(us - us2) / us > 0.03:
        print("%-6s cnt %2d  %8.1f -> %8.1f us (%+5.1f%%)  saved/iter %7.1f us   M %8s K %6s N %5s  %s" % (k[0], cnt, us, us2, 100 * (us2 - us) / us, d, k[1], k[2], k[3], k[4]))
print("total of the listed kinds: %.2f -> %.2f ms" % (tot_a / 1e3, tot_b / 1e3))
