"""Static hazard screen of the hand-scheduled kernels (CPU: hipcc cross-compiles to gfx950 assembly, nothing runs).

The ping-pong kernels issue LDS reads from inline asm and wait for them with their own `s_waitcnt lgkmcnt(N)`.  The compiler does
not know that the destination registers are written LATER than the instruction: if such a read is still in flight when control
reaches code the compiler placed there (a copy of the value, or address arithmetic in registers it considered free on that path),
the data is lost or lands on top of something else - a launch that is wrong once in a few hundred runs (round 3: hoisted fragment
reads across a barrier; found by one flaky parity test, then by tools/check_async_lds.py).  The rule the checker enforces:
no instruction touches the destination registers of an asm LDS read before the wait that covers it."""
import importlib.util
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "semantic_pyramid_for_image_generation_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result", "-S", "--cuda-device-only"]


def _checker():
    spec = importlib.util.spec_from_file_location("check_async_lds", os.path.join(ROOT, "tools", "check_async_lds.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_checker_sees_a_premature_use():
    text = """_Zkernel:
	;;#ASMSTART
	ds_read_b128 v[4:7], v1 offset:0
	;;#ASMEND
	v_add_u32_e32 v5, s0, v2
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mov_b32_e32 v8, v4
	s_endpgm
"""
    assert _checker().check(text) == 1
    assert _checker().check(text.replace("v_add_u32_e32 v5, s0, v2", "v_add_u32_e32 v9, s0, v2")) == 0


@pytest.mark.parametrize("source", ["conv_pp.hip", "conv_ppw.hip", "conv_wgrad_rows.hip"])
def test_no_register_of_an_lds_read_in_flight_is_touched(source):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, source + ".s")
        subprocess.run([hipcc] + FLAGS + [os.path.join(CSRC, source), "-o", out], check=True, capture_output=True, timeout=900)
        assert _checker().check(open(out).read()) == 0
