"""lab: per-workgroup s_memtime sums of gemm_bf16_nt256p_kernel (lib built by `bash tools/lab/build_lab.sh PPSTAMP -DPP_LAB_STAMP`):
where a tile's time goes (K-steps, end-of-step waits, epilogue until its bias has arrived, epilogue rounds, barriers behind it) and
the clock the chip holds inside the kernel (s_memtime ticks per s_memrealtime tick x 100 MHz)"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
fine = os.environ.get("PROBE_FINE", "0") == "1"
os.environ["MEANT_LIB_PATH"] = os.path.join(ROOT, "tools", "lab", "lib_PPSTAMP2.so" if fine else os.environ.get("PROBE_LIB", "lib_PPSTAMP.so"))
sys.path.insert(0, ROOT)
from meant_amd._lib import lib, check
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
f = getattr(ctypes.CDLL(os.environ["MEANT_LIB_PATH"]), "meant_lab_pp_stamps2" if fine else "meant_lab_pp_stamps")
f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
pp = int(os.environ.get("PROBE_PP", "1"))
check(lib.meant_set_option(b"nt_pp", pp), "opt")
dyn = os.environ.get("PROBE_DYN")
if dyn is not None: check(lib.meant_set_option(b"nt_dynamic", int(dyn)), "opt")
cap = int(os.environ.get("PROBE_CAP", "0"))
if cap: check(lib.meant_set_option(b"nt_grid_cap", cap), "opt")
for (M, N, K, bias) in [(786432, 768, 768, 1), (786432, 768, 768, 0), (786432, 2304, 768, 1), (786432, 768, 3072, 1), (8192, 8192, 8192, 0)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.float32)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    def run():
        check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr() if bias else None, None, 0, y.data_ptr(), N, None, M, N, K, 0, 1, st), "lin")
    for _ in range(30): run()                       # long enough for the clock to settle
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = np.zeros(256 * 2 * 16, dtype=np.uint64)
    assert f(buf.ctypes.data, buf.nbytes) == 0
    a = buf.reshape(256, 2, 16).astype(np.float64)
    print(f"M={M} N={N} K={K} bias={bias} pp={pp} cap={cap}: {ms:.3f} ms = {2.0*M*N*K/ms/1e9:.1f} TF (stamped build)")
    if fine:
        for g, nm in ((0, "leader  "), (1, "follower")):
            v = a[:, g, :]
            v = v[v[:, 8] > 0]
            steps = v[:, 8].sum()
            names = ["L0", "bar", "M0", "bar", "L1", "bar", "M1", "rest+bar"]
            print(f"  {nm}: per step: " + "  ".join(f"{names[i]} {v[:, i].sum()/steps:6.0f}" for i in range(8)) + f"   sum {v[:, :8].sum()/steps:7.0f}")
        del x, w, y
        continue
    for g, nm in ((0, "leader  "), (1, "follower")):
        v = a[:, g, :]
        v = v[v[:, 7] > 0]
        tiles, steps = v[:, 7].sum(), v[:, 6].sum()
        clk = np.median(v[:, 8] / np.maximum(v[:, 9], 1)) * 100.0
        per = lambda i: v[:, i].sum() / tiles
        print(f"  {nm}: clock {clk:6.0f} MHz; per tile ({tiles/len(v):.1f} tiles on {len(v)} CUs, {steps/tiles:.0f} steps per tile): steps kt>0 {per(0):8.0f}  step kt=0 {per(1):7.0f}  "
              f"end-of-step waits {per(2):7.0f}  epilogue->bias {per(3):7.0f}  epilogue rounds {per(4):7.0f}  barriers behind {per(5):7.0f}  "
              f"total {v[:, 8].sum()/tiles:8.0f} ticks = {v[:, 8].sum()/tiles/clk:6.2f} us")
    del x, w, y
