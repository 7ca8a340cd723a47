"""Static check of a gfx950 ISA listing (hipcc -save-temps *.s): is a register that an in-flight ds_read writes READ (or overwritten)
before an s_waitcnt lgkmcnt(n) has retired that read?  The compiler cannot see a ds_read inside inline asm as asynchronous: a
copy it inserts between the asm and the hand-placed wait (tied operands of an "+v" constraint, live-range splits) reads the
register before the data has landed -- right most of the time, wrong when LDS is contended.
    python scripts/isa_lds_hazard.py file.s [kernel-name-substring]"""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(lines, name):
    pending = []          # in-flight LGKM operations in issue order: ("lds", destination registers) or ("smem", set())
    bad = 0
    for no, raw in lines:
        ins = raw.split(";")[0].strip()
        if not ins or ins.endswith(":") or ins.startswith("."):
            continue
        op, _, rest = ins.partition(" ")
        toks = [t for t in re.split(r"[ ,]+", rest) if t]
        if op.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", ins)
            if m:
                n = int(m.group(1))
                # SMEM may return out of order: only a wait for 0 retires scalar loads; LDS operations retire in order
                if n == 0:
                    pending = []
                else:
                    while len(pending) > n:
                        pending.pop(0)
            continue
        if op.startswith("s_load") or op.startswith("s_buffer_load"):
            pending.append(("smem", set()))
            continue
        used = set()
        for t in toks:
            used |= regs(t)
        if op.startswith("ds_read"):
            dst = regs(toks[0])
            src = set().union(*[regs(t) for t in toks[1:]]) if len(toks) > 1 else set()
            inflight = set().union(*[p[1] for p in pending]) if pending else set()
            if src & inflight or dst & inflight:
                bad += 1
                print("%s:%d  %s   <- touches in-flight %s" % (name, no, ins, sorted((src | dst) & inflight)))
            pending.append(("lds", dst))
            continue
        if op.startswith("ds_"):
            pending.append(("lds", set()))
        inflight = set().union(*[p[1] for p in pending]) if pending else set()
        if op.startswith("v_mad_u64_u32") and len(toks) >= 5:
            # 32-bit a * b + c compiled as the 64-bit multiply-add: the addend is a register PAIR whose high half is undefined (the
            # compiler takes whatever register follows c) and only the low half of the result is used; carries go upward only, so a
            # stale high addend cannot reach it.  Reading that register is benign; every other operand is checked as usual.
            hi = regs(toks[4]) - {min(regs(toks[4]))} if len(regs(toks[4])) == 2 else set()
            used = used - (hi - set().union(*[regs(t) for t in toks[:4]]))
        if used & inflight:
            bad += 1
            print("%s:%d  %s   <- touches in-flight %s" % (name, no, ins, sorted(used & inflight)))
    return bad


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    text = open(path).read().splitlines()
    total = 0
    cur, body = None, []
    for i, l in enumerate(text, 1):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, body = m.group(1), []
        if cur is not None:
            body.append((i, l))
            if "s_endpgm" in l:
                if want in cur:
                    n = check(body, cur[:60])
                    print("== %s: %d hazards" % (cur[:90], n))
                    total += n
                cur = None
    print("total hazards:", total)
    return total


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
