"""Shared helpers for the GPU parity tests (HIP path vs the CPU oracle on the same seeded inputs)."""
import numpy as np
import torch

TOL = {  # (max-abs on outputs, relative on gradient norms, max-abs/scale on gradient tensors)
    torch.float32: dict(out=1e-3, gnorm=2e-3, gelem=2e-3),
    torch.bfloat16: dict(out=1e-2, gnorm=3e-2, gelem=6e-2),
}
DTYPES = [torch.float32, torch.bfloat16]
IDS = ["f32", "bf16"]


def t(a):
    return torch.from_numpy(np.asarray(a))


def maxerr(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def assert_close(a, b, tol, what=""):
    e = maxerr(a, b)
    assert e <= tol, f"{what}: max-abs error {e:.3e} > {tol:.1e}"


def assert_grad_close(a, b, tol, what=""):
    """tensor-level gradient check, normalised by the reference tensor's max magnitude"""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    scale = max(b.abs().max().item(), 1e-6)
    e = (a - b).abs().max().item() / scale
    assert e <= tol, f"{what}: relative max error {e:.3e} > {tol:.1e}"


def pair(oracle_mod, hip_mod, seed, dev):
    """fill the oracle with the deterministic recipe, copy its weights into the HIP module"""
    from oracle import meant_oracle as O
    O.fill_weights_(oracle_mod, seed)
    oracle_mod.eval()
    hip_mod.load_state_dict(oracle_mod.state_dict())
    return oracle_mod, hip_mod.to(dev).eval()


def norm_floor(ref_norms, dtype):
    """Gradients that are structurally ~0 in exact arithmetic (e.g. the key bias, to which a softmax is
    invariant) are pure rounding noise; norms are compared against max(ref, floor), floor being a small
    fraction of the largest gradient norm of the module."""
    big = max([float(x) for x in ref_norms] + [1e-12])
    return (1e-4 if dtype == torch.float32 else 2e-2) * big


def compare_param_grads(oracle_mod, hip_mod, dtype, what=""):
    tol = TOL[dtype]
    ref = dict(oracle_mod.named_parameters())
    floor = norm_floor([r.grad.double().norm().item() for r in ref.values() if r.grad is not None], dtype)
    for k, p in hip_mod.named_parameters():
        r = ref[k]
        if r.grad is None:
            assert p.grad is None or p.grad.abs().max().item() == 0, f"{what}:{k} unexpected grad"
            continue
        assert p.grad is not None, f"{what}:{k} missing grad"
        a, b = p.grad.detach().double().cpu().norm().item(), r.grad.double().norm().item()
        assert abs(a - b) <= tol["gnorm"] * max(b, floor) + 1e-7, f"{what}:{k} grad norm {a} vs {b}"
        if b > floor:
            assert_grad_close(p.grad, r.grad, tol["gelem"], f"{what}:{k}")
