"""dev probe: does the eval-mode forward capture into a HIP graph (torch.cuda.graph), and what does replay buy at small batch?"""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import meant_amd as M
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = M.meant(768, 768, 4, 224, 224, 16, 12, 2, torch.nn.Embedding(64001, 768), num_heads=12, num_encoders=1).to(dev).eval()
m.compute_dtype = torch.bfloat16
rs = np.random.RandomState(0)
ids = torch.from_numpy(rs.randint(0, 64001, (B, 12, 512))).to(dev)
img = torch.randn(B, 12, 4, 224, 224, device=dev)
mask = torch.ones(B, 12, 512, device=dev)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    ref = m(ids, img, mask).clone()
    t_eager = timeit(lambda: m(ids, img, mask))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): m(ids, img, mask)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = m(ids, img, mask)
    g.replay(); torch.cuda.synchronize()
    print("max diff vs eager", (out - ref).abs().max().item())
    t_graph = timeit(lambda: g.replay())
print(f"B={B}: eager forward {t_eager:.3f} ms, graph replay {t_graph:.3f} ms")
