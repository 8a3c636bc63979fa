"""ISA guard for registers that are "in flight" behind hipcc's back.

The streaming NT GEMMs (meant_amd/csrc/gemm_bf16.hip, gemm_bf16_nt256s_kernel and its ping-pong form gemm_bf16_nt256p_kernel) request their next tile with inline-asm memory
operations whose results arrive during the K-step and are picked up behind the wait that ends the step.  hipcc believes the
destination registers are defined the moment the asm statement ends, so nothing stops it from copying or reusing them while
the load is still outstanding.  This script compiles the file to ISA and checks, for every such request (marked by its cache
scope bits: `global_load_dword ... sc1`, `global_atomic_add ... sc0` with an `off` address), that no instruction between the
request and the next `s_waitcnt vmcnt(...)` touches the destination register.

    python tools/isa_inflight.py            # prints one line per request, exit code 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "meant_amd", "csrc", "gemm_bf16.hip")
REQ = re.compile(r"^\s*(global_load_dword|global_atomic_add)\s+v(\d+),\s*v\[\d+:\d+\],.*\boff\b.*\bsc[01]\b")


def regs_of(line):
    """all VGPR numbers an instruction line mentions (v7, v[4:7])"""
    out = set()
    code = line.split(";")[0]
    for m in re.finditer(r"\bv(\d+)\b", code):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", code):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def masked(lines, j):
    """is instruction j bracketed by `s_mov_b32 exec_lo, ...` above and `s_mov_b64 exec, ...` below inside one asm statement?"""
    up = [l.strip() for l in lines[max(0, j - 3):j]]
    down = lines[j + 1].strip() if j + 1 < len(lines) else ""
    return any(l.startswith("s_mov_b32 exec_lo") for l in up) and down.startswith("s_mov_b64 exec,")


def check(asm_text, kernel_substr="gemm_bf16_nt256s_kernel"):
    """-> list of (kernel, request line, register, n instructions in flight, offending line or None)"""
    results = []
    kernel = None
    lines = asm_text.splitlines()
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel = m.group(1) if kernel_substr in m.group(1) else None
        if kernel:
            r = REQ.match(ln)
            if r:
                reg = int(r.group(2))
                bad, n = None, 0
                j = i + 1
                while j < len(lines):
                    cur = lines[j].strip()
                    if cur.startswith("s_waitcnt") and "vmcnt" in cur:
                        break
                    if cur.startswith("s_endpgm") or re.match(r"^_Z\w+:", lines[j]):
                        bad = "no wait before the end of the kernel"
                        break
                    if cur and not cur.startswith((";", ".")) and not cur.endswith(":"):
                        n += 1
                        # another request into the SAME register is by design: the ping-pong kernel issues both of its requests
                        # as EXEC-masked asm statements into one register per step (at most one of them is live in a step)
                        r2 = REQ.match(lines[j])
                        if r2 and int(r2.group(2)) == reg and masked(lines, j):
                            j += 1
                            continue
                        if reg in regs_of(cur):
                            bad = cur
                            break
                    j += 1
                results.append((kernel, ln.strip(), reg, n, bad))
        i += 1
    return results


def compile_to_isa():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "gemm.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
                        SRC, "-o", out], check=True, cwd=os.path.dirname(SRC), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read()


if __name__ == "__main__":
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 else compile_to_isa()
    res = check(text) + check(text, "gemm_bf16_nt256p_kernel")
    rc = 0
    for kernel, req, reg, n, bad in res:
        print(f"{kernel[:60]}: `{req}` v{reg} in flight over {n} instructions: {'OK' if bad is None else 'TOUCHED BY ' + bad}")
        rc |= bad is not None
    if not res:
        print("no in-flight requests found")
        rc = 1
    sys.exit(rc)
