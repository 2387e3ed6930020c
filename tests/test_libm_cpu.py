"""fw_libm.h (glibc's logf / log10f / sinf / asinf / acosf / atanf / atan2f / powf restated for the device) compiled for the
HOST with g++ and compared, bit for bit, with the libm of the machine the test runs on — which is what the reference's
f32::ln/log10/sin/asin/acos/atan2/powf call (Rust lowers them to the platform libm).  The quick sweep covers every 4099th bit
pattern of every one-argument function, every exponent the path uses over those x, every xi the counter RNG can draw for
log10f (2^24 values: exhaustive) and 2*10^7 random pairs for atan2f / powf; `tools/libm_sweep full` (all 2^32 inputs per
function, 10^9 pairs) is quoted in the header."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restated_libm_equals_the_host_libm_bit_for_bit(tmp_path):
    exe = str(tmp_path / "libm_sweep")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(ROOT, "tools", "libm_sweep.cpp"), "-lm"])
    cp = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=600)
    print(cp.stdout)
    assert cp.returncode == 0, cp.stdout + cp.stderr
    assert "TOTAL mismatches 0" in cp.stdout


def test_the_kernels_call_no_libm_function_directly():
    """every libm-class call in the kernels goes through fw_libm.h (ocml's versions differ from glibc's by an ulp here and there)"""
    import re
    src = open(os.path.join(ROOT, "firework_amd", "csrc", "fw_kernels.hip")).read()
    code = re.sub(r"//[^\n]*", "", src)
    for name in ("sinf", "cosf", "tanf", "atan2f", "atanf", "asinf", "acosf", "powf", "logf", "log10f", "log2f", "expf", "exp2f"):
        assert not re.search(r"(?<![\w:])" + name + r"\s*\(", code), name
