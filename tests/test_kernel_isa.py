"""CPU tier (needs hipcc, no GPU): static checks of the generated gfx950 code that the pipelined kernels depend on --
tools/check_mp_isa.py: no register spill inside conv_mp's counted-wait K loop, and no compiler-inserted full vmcnt wait (or
spill) inside wgrad_wide_kernel's three-stage loop; and, over every translation unit, no VALU write to the data registers of a
12-/16-B buffer store fewer than two wait states behind it (the compiler skips that hazard for register-soffset stores, gfx950
does not)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_pipelined_kernels_keep_their_counted_waits():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mp_isa.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "wgrad_wide_kernel" in r.stdout and "conv_mp_kernel" in r.stdout and "conv_mq_kernel<" in r.stdout
    assert "STORE-DATA HAZARD" not in r.stdout and "conv_mp.hip" in r.stdout
    assert "conv0_bwd_fused_kernel" in r.stdout          # layer 0's one-pass backward: scalar group offsets (no waterfall loops), request pipeline intact
    assert "unguarded patches 0" in r.stdout and "conv_stem_pair_kernel" in r.stdout     # conv_stem.hip: every staged patch is awaited in front of its barrier
    assert "wgrad_reduce_batch_kernel   runs of" in r.stdout      # round 5: the batched split-K reduce keeps a quarter's loads in flight (no wait per load)
    assert "channel-major K" in r.stdout                 # round 5: the K-order instantiations of conv_mq are covered by the same loop checks


def test_store_data_scanner_sees_a_hazard():
    """The scanner itself: a register-soffset 16-B store whose data is rewritten by the next instruction is reported with 0
    wait states, an s_nop 1 in between makes it 2, a store whose data is not rewritten is not reported, and a permlane swap
    counts as a write of BOTH its operands."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_mp_isa as C
    asm = "\n".join([
        "_Zkernel_a:",
        "\tbuffer_store_dwordx4 v[138:141], v132, s[56:59], s85 offen",
        "\tv_pk_fma_f32 v[138:139], v[106:107], v[62:63], v[110:111]",
        "\ts_endpgm",
        ".Lfunc_end0:",
        "_Zkernel_b:",
        "\tbuffer_store_dwordx4 v[10:13], v1, s[4:7], s9 offen",
        "\ts_nop 1",
        "\tv_mov_b32_e32 v12, 0",
        "\ts_endpgm",
        ".Lfunc_end1:",
        "_Zkernel_c:",
        "\tbuffer_store_dwordx4 v[10:13], v1, s[4:7], 0 offen",
        "\tv_mov_b32_e32 v20, 0",
        "\tv_permlane16_swap_b32_e32 v30, v11",
        "\ts_endpgm",
        ".Lfunc_end2:",
        "_Zkernel_d:",
        "\tglobal_store_dwordx4 v[0:1], v[10:13], off",
        "\tv_mov_b32_e32 v20, 0",
        "\ts_endpgm",
        ".Lfunc_end3:",
    ])
    found = {k: w for k, w, _, _ in C.store_data_distances(asm)}
    assert found == {"_Zkernel_a": 0, "_Zkernel_b": 2, "_Zkernel_c": 1}
