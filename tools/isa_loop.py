"""dev aid: compile one .hip with -save-temps and print the instruction mix of a kernel's hottest loop
usage: python tools/isa_loop.py gemm_bf16.hip nt256w4 [loop_index]"""
import collections, re, subprocess, sys, os
src, pat = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 0
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "meant_amd", "csrc")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
                    "-Wno-unused-variable", "-save-temps=obj", "-c", src, "-o", "/tmp/isa_tmp.o"], cwd=root, capture_output=True, text=True)
print(r.stderr[-3000:])
asm = "/tmp/" + os.path.splitext(src)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s"
lines = open(asm).read().split("\n")
st = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + pat + r"\w*:", l)][0]
out = []
for l in lines[st:]:
    out.append(l)
    if "s_endpgm" in l:
        break
open("/tmp/kernel.s", "w").write("\n".join(out))
labels = {}
for i, l in enumerate(out):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
be = []
for i, l in enumerate(out):
    m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        be.append((labels[m.group(1)], i))
print("back edges:", be[:6])
a, b = be[which]
c = collections.Counter(l.split()[0] for l in out[a:b + 1] if l.strip() and not l.strip().startswith(";"))
print(c.most_common(16))
for l in lines:
    if pat in l and ("num_agpr" in l or "num_vgpr" in l or "private_seg_size" in l):
        print(l.strip())
print("loop written to /tmp/kernel.s lines", a, b)
