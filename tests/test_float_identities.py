"""Arithmetic identities the device code relies on where it departs from the letter of the reference's expressions."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_double_division_by_constant_equals_float_division():
    """brisk_device_detect.h divides floats by 6, 18 and 3072 in float where the reference divides in double and rounds the
    quotient to float (brisk-scale-space.cc:757-1364): the same float for every dividend.  tools/verify_float_division.c
    checks all 2^32 bit patterns in 50 s; here every 1021st (a prime stride: all exponents, varied mantissas)."""
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "vfd")
        subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "verify_float_division.c"), "-lm"])
        out = subprocess.run([exe, "1021"], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout
        assert "mismatches 0 / 0 / 0" in out.stdout
