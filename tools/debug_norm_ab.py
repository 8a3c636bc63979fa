#!/usr/bin/env python3
"""dev probe: every RMSNorm entry point on fixed inputs, outputs saved to argv[1] (run once per build with MEANT_LIB_PATH),
or with two files: compare them"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    for k in a:
        x, y = a[k].float(), b[k].float()
        print(f"{k:40s} max|a-b| {(x - y).abs().max().item():.3e}  rel norm {((x - y).norm() / max(y.norm().item(), 1e-30)).item():.3e}  |b| {y.norm().item():.3e}")
    sys.exit(0)
from meant_amd._lib import lib, check
dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
out = {}
for rows, d, S, DT in ((12288, 768, 512, 1), (4704, 768, 196, 1), (12288, 768, 512, 0), (24, 768, 12, 1), (1024, 768, 512, 1), (24, 1536, 12, 1), (392, 768, 196, 1), (24, 768, 12, 0), (24, 1536, 12, 0), (1024, 768, 512, 0), (26, 1536, 13, 0), (26, 1536, 13, 1)):
    gen = torch.Generator().manual_seed(rows + d)
    cast = (lambda z: z.bfloat16()) if DT else (lambda z: z.float())
    x = cast(torch.randn(rows, d, generator=gen).to(dev)); dy = cast(torch.randn(rows, d, generator=gen).to(dev))
    g = (1 + 0.1 * torch.randn(d, generator=gen)).to(dev); r = torch.empty(rows, device=dev)
    y = torch.empty_like(x); dx = torch.empty_like(x); ds = torch.empty(d, device=dev)
    wsb = lib.meant_rmsnorm_bwd_ws(rows, d); ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
    tag = f"{rows}x{d}{'b' if DT else 'f'}"
    check(lib.meant_rmsnorm_fwd(x.data_ptr(), g.data_ptr(), y.data_ptr(), r.data_ptr(), rows, d, 1e-8, 0.0, 0, DT, st))
    out[tag + " fwd y"] = y.clone(); out[tag + " fwd rinv"] = r.clone()
    for nm, dres, gp in (("bwd", None, None), ("bwd+dres", dy, None), ("bwd+gelu", dy, x)):
        check(lib.meant_rmsnorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), r.data_ptr(), dx.data_ptr(), ds.data_ptr(), rows, d, 1e-8, 0.0, 0,
                                    dres.data_ptr() if dres is not None else None, gp.data_ptr() if gp is not None else None, DT, ws.data_ptr(), wsb, st))
        out[f"{tag} {nm} dx"] = dx.clone(); out[f"{tag} {nm} dscale"] = ds.clone()
    if lib.meant_rmsnorm_pooled_ok(rows, d, S):
        G = rows // S
        pooled = torch.empty(G, d, device=dev); dyp = torch.randn(G, d, generator=gen).to(dev)
        for gelu in (0, 1):
            check(lib.meant_rmsnorm_fwd_pooled(x.data_ptr(), g.data_ptr(), None, r.data_ptr(), pooled.data_ptr(), rows, d, S, 0, gelu, 1e-8, 0.0, 0, DT, st))
            out[f"{tag} fwd_pooled gelu={gelu} pooled"] = pooled.clone(); out[f"{tag} fwd_pooled gelu={gelu} rinv"] = r.clone()
            check(lib.meant_rmsnorm_bwd_pooled(dyp.data_ptr(), 1, None if gelu else x.data_ptr(), g.data_ptr(), r.data_ptr(), dx.data_ptr(), ds.data_ptr(), rows, d, S, 1e-8, 0.0, 0,
                                               None, 0, x.data_ptr() if gelu else None, DT, ws.data_ptr(), wsb, st))
            out[f"{tag} bwd_pooled gelu={gelu} dx"] = dx.clone(); out[f"{tag} bwd_pooled gelu={gelu} dscale"] = ds.clone()
        check(lib.meant_rmsnorm_fwd_pooled(x.data_ptr(), g.data_ptr(), y.data_ptr(), r.data_ptr(), pooled.data_ptr(), rows, d, S, 1, 0, 1e-8, 0.0, 0, DT, st))
        out[f"{tag} fwd pool_input y"] = y.clone(); out[f"{tag} fwd pool_input pooled"] = pooled.clone()
        check(lib.meant_rmsnorm_bwd_pooled(dy.data_ptr(), 0, x.data_ptr(), g.data_ptr(), r.data_ptr(), dx.data_ptr(), ds.data_ptr(), rows, d, S, 1e-8, 0.0, 0,
                                           dyp.data_ptr(), 1, None, DT, ws.data_ptr(), wsb, st))
        out[f"{tag} bwd dres_pooled dx"] = dx.clone()
        check(lib.meant_rmsnorm_bwd_pooled(dyp.data_ptr(), 1, x.data_ptr(), g.data_ptr(), r.data_ptr(), dx.data_ptr(), ds.data_ptr(), rows, d, S, 1e-8, 0.0, 0,
                                           dyp.data_ptr(), 1, None, DT, ws.data_ptr(), wsb, st))
        out[f"{tag} bwd both pooled dx"] = dx.clone(); out[f"{tag} bwd both pooled dscale"] = ds.clone()
        check(lib.meant_rmsnorm_fwd_pooled(x.data_ptr(), g.data_ptr(), None, r.data_ptr(), pooled.data_ptr(), rows, d, S, 1, 0, 1e-8, 0.0, 0, DT, st))
        out[f"{tag} fwd stats+means rinv"] = r.clone(); out[f"{tag} fwd stats+means pooled"] = pooled.clone()
    check(lib.meant_rmsnorm_stats(x.data_ptr(), r.data_ptr(), rows, d, 1e-8, DT, st))
    out[f"{tag} stats rinv"] = r.clone()
torch.cuda.synchronize()
torch.save({k: v.cpu() for k, v in out.items()}, sys.argv[1])
