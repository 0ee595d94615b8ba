"""Build-time safety net for the attention kernels' asm-issued staging loads (csrc/attention.hip, att_load): hipcc does not track
those loads, so between each of them and the kernel's own `s_waitcnt vmcnt(0)` nothing may touch -- or spill -- the destination
registers.  Compiles the file to gfx950 assembly (no GPU needed) and checks every kernel that uses them, plus: no scratch."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_asm_issued_loads_are_never_touched_in_flight(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from asm_inflight_check import check
    out = tmp_path / "attention.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", "-o", str(out), os.path.join(ROOT, "tqdne_amd", "csrc", "attention.hip")],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    txt = out.read_text()
    kernels = re.findall(r"^(_ZN\S*(?:attention_fwd2_kernel|attention_bwd2_dq_kernel|attention_bwd2_dkv_kernel)\S*):", txt, re.M)
    assert len(kernels) >= 6, kernels   # D = 32 and 64 of each
    for k in kernels:
        n, bad = check(str(out), re.escape(k))
        assert n > 0 and not bad, (k, n, bad[:5])
        meta = txt[txt.index(".amdhsa_kernel " + k):]
        assert int(re.search(r"private_segment_fixed_size (\d+)", meta).group(1)) == 0, k   # no spills at all


def test_the_checker_flags_a_touched_register(tmp_path):
    """the checker itself: a register written between an asm-issued load and the kernel's vmcnt(0) must be reported, one written
    after the wait must not; compiler-issued loads (outside ASMSTART / ASMEND) are the compiler's business"""
    from asm_inflight_check import check
    src = """
_ZN4testEv:
	global_load_dwordx4 v[20:23], v[0:1], off offset:16
	v_mov_b32_e32 v20, 0
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v[2:3], off
	;;#ASMEND
	v_add_f32_e32 v5, v6, v7
	v_mov_b32_e32 v11, 0
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_mov_b32_e32 v12, 0
	s_endpgm
"""
    f = tmp_path / "t.s"
    f.write_text(src)
    n, bad = check(str(f), "_ZN4testEv")
    assert n == 1 and bad == ["v_mov_b32_e32 v11, 0"], (n, bad)
