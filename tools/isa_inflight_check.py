"""dev aid: in an assembly listing (tools/kernel_regs.py leaves /tmp/kernel_regs.s), flag instructions that READ a VGPR which an
inline-asm LDS read (ds_read_* inside ;;#ASMSTART .. ;;#ASMEND) has written and no s_waitcnt lgkmcnt has covered yet.  The
compiler does not know those reads are asynchronous: a copy it inserts before the wait carries the register's old content.
usage: python tools/isa_inflight_check.py [/tmp/kernel_regs.s]"""
import re, sys
def regs(tok):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out
def check(text):
    """-> list of (line number, instruction, registers) for every compiler-generated read of a register an asm LDS read still has in flight"""
    lines = text.split("\n")
    inflight = []          # list of (set of regs) in issue order
    in_asm = False
    out = []
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True; continue
        if t.startswith(";;#ASMEND"): in_asm = False; continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        args = t[len(op):]
        m = re.search(r"lgkmcnt\((\d+)\)", t)
        if op == "s_waitcnt" and m:
            n = int(m.group(1))
            while len(inflight) > n: inflight.pop(0)
            continue
        if op == "s_waitcnt" or op == "s_barrier": continue
        if op.startswith("ds_read") and in_asm:
            inflight.append(regs(args.split(",")[0]))
            continue
        if in_asm or not inflight: continue
        if op.startswith(("ds_", "s_load", "global_load_lds")):
            continue                                     # a compiler-visible LGKM op only adds to the counter: the asm waits stay conservative
        parts = args.split(",")
        srcs = ",".join(parts[1:]) if len(parts) > 1 else ""
        if op.startswith(("global_store", "scratch_store", "buffer_store")): srcs = args
        hit = regs(srcs) & set().union(*inflight)
        if hit: out.append((i + 1, t, sorted(hit)))
    return out


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/kernel_regs.s"
    res = check(open(path).read())
    for ln, t, r in res:
        print(f"{ln}: {t}    <- reads in-flight {r}")
    print("flagged:", len(res))
