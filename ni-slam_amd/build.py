"""Build libnislam_kcc_hip.so in-tree with hipcc for gfx950 (explicit hipcc -shared -fPIC; no JIT cache)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnislam_kcc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# -fno-slp-vectorize: the FFT engine issues its packed FP32 (v_pk_*_f32 with op_sel / neg modifiers) through inline asm
# (kcc_fft.h); what the SLP vectoriser packs on its own costs more in register shuffles than it saves (measured +9 %)
UNITS = [("kcc_kernels.hip", ["-fno-slp-vectorize"]), ("kcc_generic.hip", ["-fno-slp-vectorize"]), ("kcc_api.hip", ["-ffp-contract=off"]), ("kcc_tables.cpp", ["-ffp-contract=off"]), ("kcc_group.cpp", ["-ffp-contract=off"]), ("kcc_tracker.cpp", ["-ffp-contract=off"]),
         ("kcc_camera.cpp", ["-ffp-contract=off"]), ("kcc_map.cpp", ["-ffp-contract=off"]),
         ("kcc_posegraph.cpp", ["-ffp-contract=off"]), ("kcc_posegraph_dev.hip", ["-ffp-contract=off"]), ("kcc_pyramid.cpp", ["-ffp-contract=off"]),
         ("kcc_stitcher.hip", ["-ffp-contract=off"])]
HEADERS = ["kcc_tune.h", "kcc_posegraph_dev.h", "kcc_tables.h", "kcc_fft.h", "kcc_fft2.h", "kcc_consts.h", "kcc_kernels.h", "kcc_generic.h", "kcc_pointwise.h", os.path.join("..", "..", "include", "nislam_kcc.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# units whose code depends on the -DKCC_* tuning macros: a variant rebuilds only these and links the base objects of the rest
VARIANT_UNITS = ("kcc_kernels.hip", "kcc_generic.hip", "kcc_api.hip", "kcc_tracker.cpp", "kcc_posegraph.cpp")   # (the last three: kcc_tune.h switches only)


TUNE_SUFFIX = "_tune"
TUNE_LIB = LIB.replace(".so", TUNE_SUFFIX + ".so")


def _macros_used():
    """the KCC_* macro names the variant units (and the headers they include) test: a -D naming anything else is a typo"""
    import re
    names = set()
    for f in list(VARIANT_UNITS) + HEADERS:
        path = os.path.join(CSRC, f)
        if os.path.exists(path):
            names.update(re.findall(r"\bKCC_[A-Z0-9_]+\b", open(path).read()))
    return names


def build(force=False, verbose=False, defs=(), suffix=""):
    """defs/suffix: tuning variants, e.g. build(defs=["-DKCC_P360=15,24"], suffix="_p360b") -> lib..._p360b.so
    (variant objects live in csrc/_var/, git-ignored; only VARIANT_UNITS are recompiled for a variant -- `force` applies to
    those alone, and a -D that names a macro no variant unit tests is refused instead of being silently ignored)"""
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    lib = LIB.replace(".so", suffix + ".so")
    if suffix:
        build(force=False, verbose=verbose)          # the base objects the variant links
        os.makedirs(os.path.join(CSRC, "_var"), exist_ok=True)
        known = _macros_used()
        for d in defs:
            name = d[2:].split("=")[0] if d.startswith("-D") else None
            if name and name not in known:
                raise ValueError("variant %s: %s is not a macro of %s" % (suffix, name, ", ".join(VARIANT_UNITS)))
    for src, extra in UNITS:
        s = os.path.join(CSRC, src)
        var = bool(suffix) and (src in VARIANT_UNITS)
        o = os.path.join(CSRC, "_var", os.path.splitext(src)[0] + suffix + ".o") if var else os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        if (force and (var or not suffix)) or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + COMMON + extra + (list(defs) if var else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-lpthread", "-o", lib]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib


def build_tuning(force=False, verbose=False, defs=()):
    """libnislam_kcc_hip_tune.so: the same library with -DKCC_ABLATE -- every laboratory switch of kcc_tune.h alive ($NIK_ABLATE,
    $NIK_RING, $NIK_LDS_PAD_*, $NIK_FUSE_*, $NIK_LANE_CUS ...) and the ring-form B kernels compiled in.  tools/ and the tests that
    exercise a non-default form load it through $NIK_LIB; the release library contains none of it."""
    return build(force=force, verbose=verbose, defs=["-DKCC_ABLATE"] + list(defs), suffix=TUNE_SUFFIX)


if __name__ == "__main__":
    if "--tune" in sys.argv:
        print(build_tuning(force="--force" in sys.argv, verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
