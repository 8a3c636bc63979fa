#!/usr/bin/env python3
"""Launch the streaming NT GEMM (gemm_bf16_nt256p_kernel) once per shape the bench step uses, 3 times each, so that
rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) can attribute HBM traffic per shape.  N = 2304 is the fused q|k|v
projection: it is launched as the step launches it (rotary epilogue) AND as a plain Linear of the same shape (what rounds 1-3
measured); the other shapes run the plain epilogue.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python tools/pmc_nt256.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meant_amd._lib import lib, check, BF16, EPI_NONE

# (M, N, K, how): "rot" = the fused q|k|v projection as the step launches it (meant_qkv_proj_fwd with the models' rotary tables,
# modules.py:332 / :292); "plain" = meant_linear_fwd without epilogue options.  tools/parse_pmc_traffic.py walks the same list.
LAUNCHES = [(786432, 2304, 768, "rot"), (786432, 2304, 768, "plain"), (786432, 768, 768, "plain"), (786432, 768, 2304, "plain"),
            (301056, 768, 1024, "plain"), (301056, 2304, 768, "rot"), (301056, 2304, 768, "plain"), (301056, 768, 768, "plain"),
            (301056, 768, 2304, "plain")]
if __name__ == "__main__":
    dev = "cuda"
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K, how) in LAUNCHES:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        if how == "rot":
            import meant_amd
            text = M == 786432
            S = 512 if text else 196
            rot = meant_amd.RotaryEmbedding(dim=48, use_xpos=True) if text else meant_amd.RotaryEmbedding(dim=32, freqs_for="pixel")
            qa, qb, ka, kb = rot.tables(S, torch.device(dev))
            bias = torch.zeros(N, device=dev)
            for _ in range(3):
                check(lib.meant_qkv_proj_fwd(x.data_ptr(), K, w.data_ptr(), bias.data_ptr(), y.data_ptr(), M, K, S, 12, 64, qa.shape[1],
                                             qa.data_ptr(), qb.data_ptr(), ka.data_ptr(), kb.data_ptr(), BF16, st))
        else:
            for _ in range(3):
                check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), None, None, 0, y.data_ptr(), N, None, M, N, K, EPI_NONE, BF16, st))
        torch.cuda.synchronize()
        del x, w, y
    print("done")
