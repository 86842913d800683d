"""Scalar-register liveness of one kernel in a gfx950 ISA listing (hipcc -S): where the pressure peaks and what is alive there.
usage: python3 scripts/sgpr_pressure.py /tmp/hot_probe.s <mangled-name-substring> [--top N] [--at LINE]
Approximate by construction (text-level parse: first operand of an s_* / v_cmp_e64 / v_readlane instruction is its definition, the carry operand of
the VALU carry forms likewise; vcc / exec / m0 are not counted), good enough to see which values stretch across the example loop."""
import re, sys
path, key = sys.argv[1], sys.argv[2]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
at = int(sys.argv[sys.argv.index("--at") + 1]) if "--at" in sys.argv else None
L = open(path).read().splitlines()
start = next(i for i, l in enumerate(L) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("E") and ":" in l)
end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
NODEST = ("s_cmp", "s_bitcmp", "s_waitcnt", "s_barrier", "s_nop", "s_branch", "s_cbranch", "s_endpgm", "s_setprio", "s_sleep", "s_setreg", "s_sendmsg", "s_dcache", "s_icache",
          "s_store", "s_buffer_store", "s_trap", "s_sethalt", "s_code_end", "s_set_gpr")
CARRY2 = ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")
RMW = ("s_cmov", "s_addk", "s_mulk", "s_bitset")
sre = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def regs(tok):
    out = []
    for m in sre.finditer(tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


ins = []  # (line_no, mnemonic, defs, uses, targets, falls)
labels = {}
for i in range(start + 1, end):
    l = L[i].split(";")[0].strip()
    if not l:
        continue
    if l.endswith(":"):
        labels[l[:-1]] = len(ins)
        continue
    if l.startswith("."):
        continue
    mn, _, rest = l.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest.strip() else []
    d, u = [], []
    if mn.startswith(("s_branch", "s_cbranch")):
        tgt = ops[-1] if ops else None
        ins.append((i + 1, mn, [], [], [tgt], not mn.startswith("s_branch")))
        continue
    if mn.startswith("s_") and not mn.startswith(NODEST):
        d = regs(ops[0]) if ops else []
        for o in ops[1:]:
            u += regs(o)
        if mn.startswith(RMW):
            u += d
    elif (mn.startswith("v_cmp") and mn.endswith("_e64")) or mn.startswith(("v_readlane", "v_readfirstlane")):
        d = regs(ops[0])
        for o in ops[1:]:
            u += regs(o)
    elif mn.startswith(CARRY2):
        d = regs(ops[1]) if len(ops) > 1 else []
        for o in ops[2:]:
            u += regs(o)
    else:
        for o in ops:
            u += regs(o)
    ins.append((i + 1, mn, d, u, [], mn != "s_endpgm"))
n = len(ins)
succ = [[] for _ in range(n)]
for j, (ln, mn, d, u, t, falls) in enumerate(ins):
    if falls and j + 1 < n:
        succ[j].append(j + 1)
    for x in t:
        if x in labels and labels[x] < n:
            succ[j].append(labels[x])
live_in = [0] * n  # bitmasks
changed = True
while changed:
    changed = False
    for j in range(n - 1, -1, -1):
        out = 0
        for s_ in succ[j]:
            out |= live_in[s_]
        dm = 0
        for r in ins[j][2]:
            dm |= 1 << r
        um = 0
        for r in ins[j][3]:
            um |= 1 << r
        v = (out & ~dm) | um
        if v != live_in[j]:
            live_in[j] = v
            changed = True
cnt = [bin(v).count("1") for v in live_in]
print(f"{key}: {n} instructions, peak {max(cnt)} live scalar registers")
# pressure profile: the peak of every stretch of `win` instructions, in program order
win = max(1, n // top)
for a in range(0, n, win):
    j = max(range(a, min(n, a + win)), key=lambda j: cnt[j])
    print(f"  line {ins[j][0]:6d}  live {cnt[j]:3d}   {ins[j][1]}")


def last_def(j, r):
    for k in range(j - 1, -1, -1):
        if r in ins[k][2]:
            return ins[k][0], L[ins[k][0] - 1].strip()
    return 0, "(live-in)"


def next_use(j, r):
    for k in range(j, n):
        if r in ins[k][3]:
            return ins[k][0]
        if r in ins[k][2]:
            return -ins[k][0]
    return 0


if at is not None:
    j = min(range(n), key=lambda j: abs(ins[j][0] - (start + 1 + at) if False else abs(ins[j][0] - at)))
    print(f"live at line {ins[j][0]} ({cnt[j]}):")
    for r in range(110):
        if live_in[j] >> r & 1:
            dl, dt = last_def(j, r)
            print(f"   s{r:<3d} def@{dl:<6d} next use@{next_use(j, r):<7d} {dt[:110]}")
