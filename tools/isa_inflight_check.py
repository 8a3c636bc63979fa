"""dev aid: in an assembly listing (tools/kernel_regs.py leaves /tmp/kernel_regs.s), flag instructions that READ a VGPR which an
inline-asm LDS read (ds_read_* inside ;;#ASMSTART .. ;;#ASMEND) has written and no s_waitcnt lgkmcnt has covered yet.  The
compiler does not know those reads are asynchronous: a copy it inserts before the wait carries the register's old content.
usage: python tools/isa_inflight_check.py [/tmp/kernel_regs.s]"""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/kernel_regs.s"
lines = open(path).read().split("\n")
def regs(tok):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out
inflight = []          # list of (set of regs) in issue order
in_asm = False
bad = 0
for i, l in enumerate(lines):
    t = l.strip()
    if t.startswith(";;#ASMSTART"): in_asm = True; continue
    if t.startswith(";;#ASMEND"): in_asm = False; continue
    if not t or t[0] in ";." or t.endswith(":"): 
        if t.endswith(":") and not t.startswith(";"): pass
        continue
    op = t.split()[0]
    args = t[len(op):]
    m = re.search(r"lgkmcnt\((\d+)\)", t)
    if op == "s_waitcnt" and m:
        n = int(m.group(1))
        while len(inflight) > n: inflight.pop(0)
        continue
    if op in ("s_barrier",): continue
    if op.startswith("ds_read") and in_asm:
        dst = args.split(",")[0]
        inflight.append(regs(dst))
        continue
    if op.startswith(("ds_", "s_load", "global_load_lds")) and not in_asm:
        # a compiler-visible LGKM op: it takes a slot in the counter (conservative for our purpose: ignore)
        continue
    if not inflight: continue
    parts = args.split(",")
    srcs = ",".join(parts[1:]) if len(parts) > 1 else ""
    if op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store")): srcs = args
    r = regs(srcs)
    fl = set().union(*inflight) if inflight else set()
    if r & fl:
        bad += 1
        print(f"{i + 1}: {t}    <- reads in-flight {sorted(r & fl)}")
print("flagged:", bad)
