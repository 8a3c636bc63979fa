"""dev aid: instruction mix of the LARGEST loop (by span) of every kernel in a .hip file
usage: python tools/isa_mix.py attn_bf16.hip [kernel-name-substring]"""
import collections, os, re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "meant_amd", "csrc")
asm = "/tmp/isa_mix.s"
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "-fno-slp-vectorize", "--cuda-device-only", src, "-o", asm],
                   cwd=root, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-3000:]); sys.exit(1)
lines = open(asm).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)] + [len(lines)]
for si in range(len(starts) - 1):
    name = lines[starts[si]].split(":")[0]
    if pat not in name:
        continue
    k = lines[starts[si]:starts[si + 1]]
    labels = {m.group(1): i for i, l in enumerate(k) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    be = []
    for i, l in enumerate(k):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            be.append((labels[m.group(1)], i))
    if not be:
        continue
    loops = [x for x in be if any("v_mfma" in l for l in k[x[0]:x[1] + 1])] or be
    a, b = max(loops, key=lambda x: x[1] - x[0])
    c = collections.Counter()
    for l in k[a:b + 1]:
        t = l.strip()
        if not t or t[0] in ";.":
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"): c["MFMA"] += 1
        elif op.startswith(("s_waitcnt", "s_barrier", "s_cbranch", "s_branch", "s_setprio", "s_nop")): c[op] += 1
        elif op.startswith("s_"): c["SALU"] += 1
        else: c[op] += 1
    tot = sum(c.values())
    valu = sum(v for kk, v in c.items() if kk.startswith("v_"))
    regs = [l.strip() for l in lines if name[:60] in l and ("NumVgprs" in l or "ScratchSize" in l or "Occupancy" in l)]
    print(f"== {name[:80]}\n   loop lines {a}-{b}: {tot} instr, VALU {valu}, MFMA {c['MFMA']}, SALU {c['SALU']}  {regs}")
    print("   " + ", ".join(f"{kk}:{v}" for kk, v in c.most_common(28)))
