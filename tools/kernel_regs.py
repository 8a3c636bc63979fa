"""dev aid: VGPR / AGPR / SGPR counts, spills and LDS of every kernel of a .hip file (from the assembly's metadata)
usage: python tools/kernel_regs.py norm.hip [name-substring] [extra hipcc flags...]"""
import os, re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "meant_amd", "csrc")
asm = "/tmp/kernel_regs.s"
flags = ["-fno-slp-vectorize"] if "attn_bf16" in src else []
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", *flags, *sys.argv[3:], src, "-o", asm],
                   cwd=root, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-3000:]); sys.exit(1)
txt = open(asm).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\S+)", body) or [None, "?"])[1]
    sp = re.search(re.escape(name) + r".*?; ScratchSize: (\d+)", txt, re.S)
    vg = re.search(r"; NumVgprs: (\d+)\n; NumAgprs: (\d+)\n; TotalNumVgprs: (\d+)", txt[txt.find(name + ":"):])
    occ = re.search(r"; Occupancy: (\d+)", txt[txt.find(name + ":"):])
    print(f"{name[:110]:110s} vgpr {vg.group(1) if vg else '?':>4} agpr {vg.group(2) if vg else '?':>4} total {vg.group(3) if vg else '?':>4} scratch {sp.group(1) if sp else '?':>5} occ {occ.group(1) if occ else '?'} lds {g('group_segment_fixed_size')}")
