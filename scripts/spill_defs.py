"""Which scalar values a kernel spills to VGPR lanes: every v_writelane of the kernel in an ISA listing (scripts/hot_probe.py leaves one in /tmp/hot_probe.s)
with the instruction that defined the spilled register.  usage: python3 scripts/spill_defs.py /tmp/hot_probe.s <mangled-name-substring>"""
import re, sys
L = open(sys.argv[1]).read().splitlines()
st = next(i for i, l in enumerate(L) if l.startswith("_Z") and sys.argv[2] in l and ":" in l)
en = next(i for i in range(st, len(L)) if L[i].startswith(".Lfunc_end"))
for i in range(st, en):
    m = re.search(r"v_writelane_b32 (v\d+), (s\d+), (\d+)", L[i])
    if not m:
        continue
    n, d = int(m.group(2)[1:]), "(live-in)"
    for k in range(i - 1, st, -1):
        t = L[k].split(";")[0].strip()
        ops = t.split(None, 1)
        if len(ops) < 2 or t.endswith(":") or ops[0].startswith(("s_cmp", "s_cbranch", "buffer_store", "ds_write", "s_waitcnt", "v_writelane")):
            continue
        first = ops[1].split(",")[0].strip()
        mm = re.match(r"s\[(\d+):(\d+)\]", first)
        if first == m.group(2) or (mm and int(mm.group(1)) <= n <= int(mm.group(2))):
            d = f"{k + 1 - st}: {t}"
            break
    print(f"+{i + 1 - st:<6d} {m.group(1)}[{m.group(3):>2s}] <- {d[:110]}")
