"""Build-time guard (no GPU): registers that inline asm leaves "in flight" in the streaming NT GEMM must not be touched by
compiler-generated code before the wait that covers them (tools/isa_inflight.py explains why)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("isa_inflight", os.path.join(ROOT, "tools", "isa_inflight.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_checker_flags_a_touched_register():
    tool = _tool()
    good = """_Z23gemm_bf16_nt256s_kernelv:
	global_load_dword v9, v[2:3], off sc1
	v_add_u32_e32 v4, v5, v6
	s_waitcnt vmcnt(8)
	v_readfirstlane_b32 s4, v9
	s_endpgm
"""
    bad = good.replace("v_add_u32_e32 v4, v5, v6", "v_mov_b32_e32 v12, v9")
    ranged = good.replace("v_add_u32_e32 v4, v5, v6", "ds_write_b128 v1, v[8:11]")
    assert [r[4] for r in tool.check(good)] == [None]
    assert tool.check(bad)[0][4] is not None
    assert tool.check(ranged)[0][4] is not None


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_tile_draw_registers_stay_untouched_in_flight():
    tool = _tool()
    res = tool.check(tool.compile_to_isa())
    # six kernel instantiations (plain / rotary / extended epilogue, each with the DMA issue at the top of the K-step or split in time
    # between the two waves of a SIMD), each with the mailbox read and the counter draw
    assert len(res) == 12, res
    for kernel, req, reg, n, bad in res:
        assert bad is None, f"{kernel}: `{req}`: v{reg} touched while in flight by `{bad}`"
        assert n > 100, f"{kernel}: `{req}` is no longer issued ahead of the K-step body ({n} instructions)"


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_tile_draw_register_of_the_ping_pong_kernel_stays_untouched_in_flight():
    """gemm_bf16_nt256p_kernel: both requests are EXEC-masked asm statements into one register per step, in straight-line code (a
    branch join is where hipcc once copied the ticket while it was in flight); six epilogue modes x two requests"""
    tool = _tool()
    res = tool.check(tool.compile_to_isa(), "gemm_bf16_nt256p_kernel")
    assert len(res) == 12, res
    for kernel, req, reg, n, bad in res:
        assert bad is None, f"{kernel}: `{req}`: v{reg} touched while in flight by `{bad}`"


def test_checker_accepts_a_masked_second_request_into_the_same_register():
    tool = _tool()
    text = """_Z23gemm_bf16_nt256p_kernelv:
	s_mov_b64 s[70:71], exec
	s_mov_b32 exec_lo, s72
	s_mov_b32 exec_hi, 0
	global_atomic_add v220, v[2:3], v219, off sc0
	s_mov_b64 exec, s[70:71]
	s_mov_b64 s[70:71], exec
	s_mov_b32 exec_lo, s73
	s_mov_b32 exec_hi, s73
	global_load_dword v220, v[2:3], off sc1
	s_mov_b64 exec, s[70:71]
	v_add_u32_e32 v4, v5, v6
	s_waitcnt vmcnt(4)
	s_endpgm
"""
    assert [r[4] for r in tool.check(text, "gemm_bf16_nt256p_kernel")] == [None, None]
    unmasked = text.replace("\ts_mov_b32 exec_lo, s73\n\ts_mov_b32 exec_hi, s73\n", "")
    assert tool.check(unmasked, "gemm_bf16_nt256p_kernel")[0][4] is not None


# ---- attention kernels: LDS reads issued by inline asm, waited for by a later asm statement ---------------------------------
def _attn_tool():
    spec = importlib.util.spec_from_file_location("isa_inflight_check", os.path.join(ROOT, "tools", "isa_inflight_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_attention_checker_flags_a_copy_made_before_the_wait():
    tool = _attn_tool()
    good = """k:
	;;#ASMSTART
	ds_read_b64_tr_b16 v[8:9], v2 offset:0
	;;#ASMEND
	v_add_u32_e32 v4, v5, v6
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mov_b64_e32 v[20:21], v[8:9]
"""
    bad = good.replace("v_add_u32_e32 v4, v5, v6", "v_mov_b64_e32 v[20:21], v[8:9]")
    assert tool.check(good) == []
    assert [r[2] for r in tool.check(bad)] == [[8, 9]]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("src", ["attn_bwd1.hip", "attn_bf16.hip", "gemm_bf16.hip"])
def test_attention_kernels_do_not_touch_lds_reads_in_flight(src, tmp_path):
    """the compiler does not know that a register written by an asm `ds_read` is still on its way: a copy it inserts between the
    read and the wait carries the old content (that was one wrong 32 x 32 block of dQ in three million, now and then)"""
    import subprocess
    tool = _attn_tool()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "k.s"
    flags = [] if src == "gemm_bf16.hip" else ["-fno-slp-vectorize"]           # as meant_amd/csrc/Makefile builds each file
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", *flags,
                    os.path.join(ROOT, "meant_amd", "csrc", src), "-o", str(out)], check=True, cwd=os.path.join(ROOT, "meant_amd", "csrc"))
    res = tool.check(out.read_text())
    assert res == [], res[:5]
