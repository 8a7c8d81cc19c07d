"""Builds libbrisk_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbrisk_hip.so")                  # tests / tools / bench.py: built with -DBRISK_HIP_TUNING
LIB_RELEASE = os.path.join(HERE, "libbrisk_hip_release.so")  # what a maintainer links: no env knobs, no debug bits, no brisk_hip_debug_*
TUNING = ["-DBRISK_HIP_TUNING"]
SOURCES = ["brisk_kernels.hip", "brisk_describe.hip", "brisk_export.hip", "brisk_image16.hip", "brisk_match.hip", "brisk_uniformity.hip", "brisk_comm.hip", "brisk_capi.hip", "brisk_pattern.cpp"]
# -ffp-contract=off: the reference binary has no FMA contraction (built with -mssse3 only); the
# sub-pixel / sub-scale float expressions must round after every operation to stay bit-exact.
# -simplifycfg-sink-common=false: sinking the common tails of the per-layer-class branches of the refinement code
# turns the three layer views into pointer PHIs, which forces them (and every score-block access) into scratch.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-mllvm", "-simplifycfg-sink-common=false",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


LINK_LIBS = ["-ldl"]


def needs_build(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", f) for f in ("brisk_hip.h", "brisk_hip_debug.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def kernel_revision():
    """Short hash of everything that defines the device code (committed PMC data names the revision it measured)."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h", ".inc")) and f not in ("brisk_capi.hip", "brisk_pool.inc"):  # (host code only)
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:12]


def build_variant(name, defines):
    """tuning experiments: the same sources with extra -D knobs into ethzasl_brisk_amd/<name>.so (load with BRISK_HIP_LIB);
    kernel_resources() does not see these builds (it describes the objects of build() / build_release())"""
    out = os.path.join(HERE, name + ".so")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + TUNING + ["-D" + d for d in defines] + ['-DBRISK_KERNEL_REV="%s"' % kernel_revision(), "-o", out] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    subprocess.check_call(cmd)
    return out


def _compile_objects(hipcc, flags, objdir, verbose, force):
    """one hipcc -c per source, all at once (independent translation units); an object is kept when neither its source,
    a header of csrc/ or include/, nor the command line changed (brisk_capi.hip alone carries the kernel revision)"""
    import hashlib
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs += [os.path.join(HERE, "..", "include", f) for f in ("brisk_hip.h", "brisk_hip_debug.h")]
    hdr_t = max(os.path.getmtime(h) for h in hdrs)
    procs, objs = [], []
    for s in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        src = os.path.join(CSRC, s)
        fl = list(flags)
        if s == "brisk_capi.hip":
            fl.append('-DBRISK_KERNEL_REV="%s"' % kernel_revision())
        cmd = [hipcc] + fl + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", obj, src]
        stamp = obj + ".cmd"
        sig = hashlib.sha1(" ".join(cmd).encode()).hexdigest()
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == sig
                and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t)):
            continue
        if verbose:
            print(" ".join(cmd))
        # (stderr = warnings + the resource remarks of every kernel: kept beside the object for kernel_resources())
        lf = open(obj + ".log", "w")
        procs.append((s, stamp, sig, obj + ".log", subprocess.Popen(cmd, stderr=lf), lf))
    for s, stamp, sig, log, p, lf in procs:
        rc = p.wait()
        lf.close()
        text = open(log).read()
        diag = _strip_remarks(text)
        if diag.strip():
            sys.stderr.write(diag)
        if rc != 0:
            raise subprocess.CalledProcessError(rc, "hipcc -c " + s)
        open(stamp, "w").write(sig)
    return objs


def _strip_remarks(text):
    """compiler output without the resource remarks (each a header line + the source lines it points at)"""
    import re
    out, held, skipping = [], [], False
    for ln in text.splitlines(True):
        if ln.startswith("In file included from "):
            held.append(ln)  # (belongs to the diagnostic that follows)
            continue
        if re.match(r"^(\S.*?: (remark|warning|error|note|fatal error): |\d+ (warning|error)s? generated)", ln):
            skipping = ": remark: " in ln
            if not skipping:
                out += held
            held = []
        if not skipping:
            out.append(ln)
    return "".join(out)


def kernel_resources(objdir=None):
    """{kernel name (mangled): {"vgpr", "agpr", "sgpr", "scratch", "occupancy", "lds"}} of the last build(), from the
    compiler's resource remarks kept beside the objects ({} if the objects were not built here)"""
    import re
    objdir = objdir or os.path.join(HERE, "..", "build", "obj")
    out = {}
    if not os.path.isdir(objdir):
        return out
    for f in sorted(os.listdir(objdir)):
        if not f.endswith(".o.log"):
            continue
        t = open(os.path.join(objdir, f)).read()
        for m in re.finditer(r"Function Name: (\S+).*?\n(.*?)LDS Size \[bytes/block\]: (\d+)", t, re.S):
            body = m.group(2)
            def g(k):  # (a field the compiler's remark format lacks is left out, not an error)
                mm = re.search(k + r": (\d+)", body)
                return int(mm.group(1)) if mm else None
            rec = {"vgpr": g("VGPRs"), "agpr": g("AGPRs"), "sgpr": g("SGPRs"), "scratch": g(r"ScratchSize \[bytes/lane\]"),
                   "occupancy": g(r"Occupancy \[waves/SIMD\]"), "lds": int(m.group(3))}
            out[m.group(1)] = {k: v for k, v in rec.items() if v is not None}
    return out


def _build(lib, objdir, extra_flags, force, verbose):
    if not force and not needs_build(lib):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cflags = [f for f in FLAGS if f != "-shared"] + extra_flags
    objs = _compile_objects(hipcc, cflags, objdir, verbose, force)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + LINK_LIBS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


def build(force=False, verbose=False):
    """libbrisk_hip.so: the engine with its test / tuning scaffolding (-DBRISK_HIP_TUNING: environment knobs, debug bits,
    the brisk_hip_debug_* entry points of include/brisk_hip_debug.h) - what the test suite, tools/ and bench.py load"""
    extra = os.environ.get("BRISK_HIPCC_EXTRA", "").split()  # tuning experiments only (e.g. -DBRISK_DETECT_ROWS_PER_THREAD=2)
    return _build(LIB, os.path.join(HERE, "..", "build", "obj"), TUNING + extra, force, verbose)


def build_release(force=False, verbose=False):
    """libbrisk_hip_release.so: the same sources without the scaffolding - the library INTEGRATION.md links"""
    return _build(LIB_RELEASE, os.path.join(HERE, "..", "build", "obj_release"), [], force, verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_release(force="--force" in sys.argv, verbose=True)
