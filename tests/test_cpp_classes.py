"""The drop-in C++ host classes (include/brisk/*.h) over the C ABI, exercised by a C++ program that mirrors the
reference's own golden test (tests/cpp/test_binary_equal.cc)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_binary_equal")


def build_binary(name="test_binary_equal", hip_runtime=False, defines=(), extra_flags=()):
    from ethzasl_brisk_amd import build
    build.build()
    src = os.path.join(ROOT, "tests", "cpp", name + ".cc")
    out = os.path.join(ROOT, "tests", "cpp", name)
    hdrs = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(ROOT, "include")) for f in fs]
    hdrs.append(os.path.join(ROOT, "tests", "cpp", "set_serialization.h"))
    hdrs.append(os.path.join(ROOT, "tests", "cpp", "synthetic_frame.h"))
    if not os.path.exists(out) or any(os.path.getmtime(p) > os.path.getmtime(out) for p in [src] + hdrs):
        # hip_runtime: the test itself allocates device memory (plain g++ against the HIP runtime's C API)
        hip = ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lamdhip64"] if hip_runtime else []
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include")] +
                              ["-D" + d for d in defines] + ["-o", out, src] + list(extra_flags) +
                              ["-L" + os.path.join(ROOT, "ethzasl_brisk_amd"), "-lbrisk_hip"] + hip +
                              ["-Wl,-rpath," + os.path.join(ROOT, "ethzasl_brisk_amd"), "-Wl,-rpath,/opt/rocm/lib"])
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


def test_gather_test_compiles_and_fails_loudly_without_gpu():
    """tests/cpp/test_gather.cc: the C++ side of the multi-GPU batch path (brisk_hip_comm_* on RCCL, no torch)"""
    import ethzasl_brisk_amd as B
    b = build_binary("test_gather", hip_runtime=True)
    if B.load_library().brisk_hip_device_count() > 0:
        pytest.skip("GPU present")
    r = subprocess.run([b], capture_output=True, text=True)
    assert r.returncode == 2 and "brisk_hip_create failed" in r.stdout


@pytest.mark.gpu
def test_result_gather_through_the_c_abi_world_1():
    """three host-fed batches, gathered with brisk_hip_comm_gather_results on a one-rank RCCL communicator (both send
    slabs in use), every row compared with brisk_hip_batch_download"""
    b = build_binary("test_gather", hip_runtime=True)
    r = subprocess.run([b], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "gather OK" in r.stdout


def test_host_results_test_compiles_and_fails_loudly_without_gpu():
    """tests/cpp/test_host_results.cc: the batch path host memory -> host memory from C++ through the C ABI alone"""
    import ethzasl_brisk_amd as B
    b = build_binary("test_host_results")
    if B.load_library().brisk_hip_device_count() > 0:
        pytest.skip("GPU present")
    r = subprocess.run([b], capture_output=True, text=True)
    assert r.returncode == 2 and "brisk_hip_create failed" in r.stdout


@pytest.mark.gpu
def test_host_to_host_batches_through_the_c_abi():
    """seven batches over a page-locked frame ring with two destination sets alternating (brisk_hip_host_register,
    brisk_hip_detect_describe_batch_host_results, brisk_hip_batch_download_wait), every compared frame equal to its per-frame
    download, a rerun of a batch's frames equal to the batch, a destination one row short reports the cut frame"""
    b = build_binary("test_host_results")
    r = subprocess.run([b], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "host results OK" in r.stdout


def build_fake_rccl():
    """tests/cpp/fake_rccl.cc -> tests/cpp/libfake_rccl.so: the eight RCCL entry points brisk_comm.hip uses, over Unix sockets
    and staged hipMemcpy (a test double: several ranks on the one GPU of a test box)"""
    src = os.path.join(ROOT, "tests", "cpp", "fake_rccl.cc")
    out = os.path.join(ROOT, "tests", "cpp", "libfake_rccl.so")
    if not os.path.exists(out) or os.path.getmtime(src) > os.path.getmtime(out):
        subprocess.check_call(["g++", "-shared", "-fPIC", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               "-o", out, src, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_fake_rccl_builds_and_exports_the_rccl_subset():
    import ctypes
    lib = ctypes.CDLL(build_fake_rccl())
    for sym in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv",
                "ncclGetErrorString"):
        assert hasattr(lib, sym), sym


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_result_gather_through_the_c_abi_with_peers(tmp_path, world):
    """brisk_hip_comm_gather_results with PEERS: world 2 and 3, one fresh process per rank, all on the box's one GPU over the
    socket double of RCCL (BRISK_HIP_RCCL_LIB; RCCL itself refuses two ranks on one device).  Executes what world 1 never
    does - every ncclSend of a non-root rank, every ncclRecv into rank r's slab at d_counts + r * frames_max, d_kps + r *
    nb_k, d_desc + r * nb_d - with shards of unequal size (11 frames over 2 / 3 ranks) and three batches in a row (both
    send slabs reused); the root compares every peer's slab with its own recomputation of that peer's frames."""
    b = build_binary("test_gather", hip_runtime=True)
    env = dict(os.environ, BRISK_HIP_RCCL_LIB=build_fake_rccl())
    idf = str(tmp_path / "unique_id.bin")
    procs = [subprocess.Popen([b, "--rank", str(r), "--world", str(world), "--id-file", idf], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    print("\n".join(outs))
    assert [p.returncode for p in procs] == [0] * world, outs
    assert "gather OK: world %d" % world in outs[0]
    for r in range(1, world):
        assert "rank %d of %d done" % (r, world) in outs[r]


OPENCV_STUB = os.path.join(ROOT, "tests", "cpp", "opencv_stub")


def opencv_flags():
    """(flags, real): real OpenCV 4 where pkg-config finds it, else the header double under tests/cpp/opencv_stub (the public
    interface of cv::Mat / KeyPoint / DMatch / InputArray / OutputArray / Ptr / Feature2D / DescriptorMatcher as far as
    include/brisk/*.h use it; test infrastructure: it pins nothing about OpenCV, it makes the three #ifdef branches compile)"""
    import shutil
    if shutil.which("pkg-config") and subprocess.run(["pkg-config", "--exists", "opencv4"]).returncode == 0:
        return subprocess.check_output(["pkg-config", "--cflags", "--libs", "opencv4"], text=True).split(), True
    return ["-I" + OPENCV_STUB], False


def build_opencv_binary(name):
    """tests/cpp/<name>.cc against the BRISK_HAVE_OPENCV branch of the drop-in headers -> tests/cpp/<name>_cv"""
    from ethzasl_brisk_amd import build
    build.build()
    flags, _ = opencv_flags()
    src = os.path.join(ROOT, "tests", "cpp", name + ".cc")
    out = os.path.join(ROOT, "tests", "cpp", name + "_cv")
    hdrs = [os.path.join(d, f) for top in (os.path.join(ROOT, "include"), OPENCV_STUB) for d, _, fs in os.walk(top) for f in fs]
    hdrs.append(os.path.join(ROOT, "tests", "cpp", "set_serialization.h"))
    if not os.path.exists(out) or any(os.path.getmtime(p) > os.path.getmtime(out) for p in [src] + hdrs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"),
                               "-DBRISK_HAVE_OPENCV", "-o", out, src] + flags +
                              ["-L" + os.path.join(ROOT, "ethzasl_brisk_amd"), "-lbrisk_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "ethzasl_brisk_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return out


@pytest.mark.parametrize("name", ["test_binary_equal", "test_threads", "test_capacity", "test_image16", "test_multi_image"])
def test_opencv_branch_of_the_drop_in_headers_compiles(name):
    """include/brisk/*.h have a cv::Feature2D / cv::DescriptorMatcher branch (-DBRISK_HAVE_OPENCV: the classes derive from
    the OpenCV bases and take cv::InputArray / cv::OutputArray, brisk/include/brisk/brisk.h:56-59 of the reference) - the
    branch INTEGRATION.md tells a maintainer to use.  Compiled with -Wall -Werror against real OpenCV where the image has
    it, against the header double otherwise (never skipped); its first run found a cv::Ptr conversion that OpenCV 3
    rejects."""
    b = build_opencv_binary(name)
    if name == "test_binary_equal":
        import ethzasl_brisk_amd as B
        if B.load_library().brisk_hip_device_count() == 0:
            r = subprocess.run([b, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True)
            assert r.returncode == 2 and "brisk_hip_create failed" in r.stdout   # no CPU fallback behind this branch either


@pytest.mark.gpu
def test_reference_golden_through_the_opencv_branch():
    """the reference's golden and matching tests through cv::Feature2D::detect -> detectAndCompute, the extractor's compute
    overloads and cv::DescriptorMatcher::add / knnMatch -> knnMatchImpl of the OpenCV branch"""
    b = build_opencv_binary("test_binary_equal")
    r = subprocess.run([b, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "Verification success" in r.stdout
    assert r.stdout.count("OK") == 5


@pytest.mark.gpu
def test_four_threads_through_the_opencv_branch():
    b = build_opencv_binary("test_threads")
    r = subprocess.run([b, os.path.join(ROOT, "tests", "golden"), "4", "6"], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "threads OK" in r.stdout


def test_thread_test_compiles():
    build_binary("test_threads")
    build_binary("test_multi_image")


@pytest.mark.gpu
@pytest.mark.parametrize("opencv", [False, True])
def test_multi_image_overloads_equal_the_single_image_calls(opencv):
    """detect(vector<Mat>, vector<vector<KeyPoint>>) / compute(vector<Mat>, ..., vector<Mat>) - the overloads the reference's classes
    inherit from cv::FeatureDetector / cv::DescriptorExtractor - as one batch (brisk_hip_detect_images / _describe_images): seven
    images of one size (the goldens, mirrored copies, a blank one), a list with a smaller image and a list with masks (both image
    by image), every image bit-equal to its single-image call; plain headers and the cv::Feature2D branch (InputArrayOfArrays)"""
    b = build_opencv_binary("test_multi_image") if opencv else build_binary("test_multi_image")
    r = subprocess.run([b, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "multi-image OK" in r.stdout


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


def test_capacity_test_compiles():
    build_binary("test_capacity")


CAPACITY_CASES = [
    ("noise_vga_thr20", lambda np: np.random.default_rng(1).integers(0, 256, (480, 640), dtype=np.uint8), 20, 4),
    ("tie_blocks_thr21", lambda np: np.kron(np.random.default_rng(1487 * 8).integers(0, 4, (739 // 3 + 1, 525 // 3 + 1)) * 80 + 7,
                                            np.ones((3, 3)))[:739, :525].astype(np.uint8), 21, 2),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,mk,thr,octaves", CAPACITY_CASES, ids=[c[0] for c in CAPACITY_CASES])
def test_drop_in_classes_grow_the_workspace(tmp_path, name, mk, thr, octaves):
    """images with more candidates / ties / keypoints than the default workspace holds (every tenth pixel of a noise
    image is a keypoint; block images are all ties): the C ABI answers BRISK_HIP_ERR_CAPACITY, the drop-in classes grow
    the workspace and repeat the call - the caller sees the reference's result (which has no capacities)"""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import ethzasl_brisk_amd as B
    b = build_binary("test_capacity")
    img = np.ascontiguousarray(mk(np))
    raw, out = tmp_path / "img.raw", tmp_path / "out.bin"
    img.tofile(raw)
    r = subprocess.run([b, str(raw), str(img.shape[1]), str(img.shape[0]), str(thr), str(octaves), str(out)], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0
    blob = out.read_bytes()
    ndet, ndesc, drows, dcols = np.frombuffer(blob[:16], np.int32)
    KP = B.KEYPOINT
    kd = np.frombuffer(blob[16:16 + ndet * KP.itemsize], KP)
    kg = np.frombuffer(blob[16 + ndet * KP.itemsize:16 + (ndet + ndesc) * KP.itemsize], KP)
    dg = np.frombuffer(blob[16 + (ndet + ndesc) * KP.itemsize:], np.uint8).reshape(drows, dcols)
    ko = O.detect(img, thr, octaves)
    ko2, do = O.Extractor().compute(img, ko)
    assert len(ko) > 16384   # beyond the classes' first output capacity
    assert kd.tobytes() == ko.tobytes()
    assert kg.tobytes() == ko2.tobytes() and np.array_equal(dg, do)
