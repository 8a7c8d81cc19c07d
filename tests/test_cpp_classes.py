"""The drop-in C++ host classes (include/brisk/*.h) over the C ABI, exercised by a C++ program that mirrors the
reference's own golden test (tests/cpp/test_binary_equal.cc)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_binary_equal")


def build_binary(name="test_binary_equal"):
    from ethzasl_brisk_amd import build
    build.build()
    src = os.path.join(ROOT, "tests", "cpp", name + ".cc")
    out = os.path.join(ROOT, "tests", "cpp", name)
    hdrs = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(ROOT, "include")) for f in fs]
    hdrs.append(os.path.join(ROOT, "tests", "cpp", "set_serialization.h"))
    if not os.path.exists(out) or any(os.path.getmtime(p) > os.path.getmtime(out) for p in [src] + hdrs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"), "-o", out,
                               src, "-L" + os.path.join(ROOT, "ethzasl_brisk_amd"), "-lbrisk_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "ethzasl_brisk_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_host_classes_compile_and_fail_loudly_without_gpu():
    import ethzasl_brisk_amd as B
    b = build_binary()
    if B.load_library().brisk_hip_device_count() > 0:
        pytest.skip("GPU present")
    r = subprocess.run([b, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True)
    assert r.returncode == 2 and "brisk_hip_create failed" in r.stdout   # no CPU fallback behind the classes


@pytest.mark.gpu
def test_reference_golden_through_cpp_classes():
    b = build_binary()
    r = subprocess.run([b, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "Verification success" in r.stdout
    assert r.stdout.count("OK") == 5


def test_thread_test_compiles():
    build_binary("test_threads")


@pytest.mark.gpu
def test_four_threads_bit_equal_to_serial():
    """per-thread workspaces (include/brisk/hip-context.h): 4 threads, different detector parameters incl. uniformity
    radii, one shared extractor - every result bit-equal to the serial run"""
    b = build_binary("test_threads")
    r = subprocess.run([b, os.path.join(ROOT, "tests", "golden"), "4", "12"], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "threads OK" in r.stdout


def test_set_container_cpp_round_trip(tmp_path):
    """the reference's golden .set files through the C++ reader / writer (tests/cpp/set_serialization.h): byte-identical"""
    b = build_binary("test_serialization")
    r = subprocess.run([b, os.path.join(ROOT, "tests", "golden"), str(tmp_path / "rt.set")], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and r.stdout.count("OK") == 3


def test_image16_header_compiles_and_fails_loudly_without_gpu():
    import ethzasl_brisk_amd as B
    b = build_binary("test_image16")
    if B.load_library().brisk_hip_device_count() > 0:
        pytest.skip("GPU present")
    r = subprocess.run([b], capture_output=True, text=True)
    assert r.returncode == 2 and "exception" in r.stdout   # no CPU fallback behind the functions


@pytest.mark.gpu
def test_image16_functions_through_the_cpp_header():
    """brisk::Halfsample16 / Twothirdsample16 / IntegralImage16 (include/brisk/internal/image-functions-16.h) against
    per-pixel readings of the reference arithmetic"""
    b = build_binary("test_image16")
    r = subprocess.run([b], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and r.stdout.count("OK") == 3
