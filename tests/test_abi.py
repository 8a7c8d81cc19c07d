"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports what the header declares."""
import os
import re

import pytest

import ethzasl_brisk_amd as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ethzasl_brisk_amd import build
    build.build()
    return B.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "brisk_hip.h")).read()
    declared = set(re.findall(r"\b(brisk_hip_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(B.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    dbg = open(os.path.join(ROOT, "include", "brisk_hip_debug.h")).read()
    declared_dbg = set(re.findall(r"\b(brisk_hip_debug_[a-z_0-9]+)\s*\(", dbg))
    assert declared_dbg == set(B.DEBUG_SYMBOLS)
    for s in declared_dbg:
        assert hasattr(lib, s), s          # (libbrisk_hip.so is the test / tuning build)


def test_release_library_has_no_scaffolding():
    """libbrisk_hip_release.so - what INTEGRATION.md links - exports the boundary and nothing of the test / tuning
    scaffolding: no brisk_hip_debug_* symbol, no tuning variable name in its strings (the three documented BRISK_HIP_*
    variables - RCCL library, image cache opt-in - remain)."""
    import ctypes
    import subprocess
    from ethzasl_brisk_amd import build
    rel = build.build_release()
    L = ctypes.CDLL(rel)
    for s in B.ABI_SYMBOLS:
        assert hasattr(L, s), s
    syms = subprocess.check_output(["nm", "-D", "--defined-only", rel], text=True)
    assert "brisk_hip_debug_" not in syms
    exported = set(re.findall(r" T (brisk_hip_[a-z_0-9]+)", syms))
    assert exported == set(B.ABI_SYMBOLS), exported ^ set(B.ABI_SYMBOLS)
    txt = subprocess.check_output(["strings", "-n", "6", rel], text=True)
    for frag in ("BRISK_TR_", "BRISK_II_", "BRISK_INTEGRAL_", "BRISK_SB_", "BRISK_CR_", "BRISK_CS_", "BRISK_DETECT_PROBE", "BRISK_SIDE_PRIO",
                 "BRISK_L0_INPLACE", "BRISK_HOST_SLICE"):
        assert frag not in txt, frag
    tun = subprocess.check_output(["strings", "-n", "6", build.build()], text=True)
    assert "BRISK_TR_BANDS" in tun        # (the check above can find what it looks for)


def test_no_cpu_fallback(lib):
    """Without a GPU every compute entry point must fail loudly."""
    if lib.brisk_hip_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(B.BriskHipError) as ei:
        B.Context()
    assert ei.value.code == 2   # BRISK_HIP_ERR_NO_DEVICE


def test_product_does_not_reference_oracle():
    for d, _, files in os.walk(os.path.join(ROOT, "ethzasl_brisk_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".inc")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "brisk_oracle" not in txt and "liboracle" not in txt and "oracle_lib" not in txt, f
    for d, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "oracle" not in open(os.path.join(d, f), errors="ignore").read(), f


def test_kernel_resources_hold_their_bounds():
    """Register bounds the launch code relies on (brisk_launch_detect: the large-batch tie kernel runs beside two integral
    workgroups per SIMD only at 96 VGPRs or fewer), and no kernel of the hot path may touch scratch."""
    from ethzasl_brisk_amd import build
    build.build()
    res = build.kernel_resources()
    if not res:
        pytest.skip("the objects were not compiled here (no resource remarks beside them)")
    by = lambda frag: {k: v for k, v in res.items() if frag in k}
    large = [v for k, v in by("k_tie_resolve").items() if "k_tie_resolve_small" not in k and "k_tie_resolve_pair" not in k]
    assert len(large) == 1 and large[0]["vgpr"] <= 96, large
    assert by("k_tie_resolve_pair") and all(v["vgpr"] <= 128 for v in by("k_tie_resolve_pair").values())
    for frag in ("k_detect", "k_describe", "k_tie_resolve", "k_pyramid", "k_score_blocks", "k_classify_refine", "k_finalize",
                 "k_integral_final", "k_desc_prepare"):
        ks = by(frag)
        assert ks, frag
        for k, v in ks.items():
            assert v["scratch"] == 0, (k, v)
