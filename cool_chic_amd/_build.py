"""Builds libccd.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["ccd_format.cpp", "ccd_writer.cpp", "ccd_api.cpp", "ccd_entropy.hip", "ccd_entropy_pipe.hip", "ccd_float.hip", "ccd_synth_fused.hip", "ccd_fused.hip", "ccd_fused_pre.hip", "ccd_fused_cr.hip", "ccd_inter.hip", "ccd_png.hip", "ccd_rate.hip"]
HEADERS = ["ccd_format.hpp", "ccd_device.hpp", "ccd_laplace.hpp", "ccd_fused_kernel.inc", "ccd_exp_table.inc", "ccd_dec_block16.inc", "ccd_dec_block16p.inc", "ccd_dec_block16pm.inc", "ccd_dec_parts8.inc", "ccd_dec_parts4.inc", "ccd_dec_parts4_nc.inc", "ccd_dec_tramp16p.inc", "ccd_dec_block32.inc", "ccd_dec_tramp16.inc", "ccd_dec_tramp32.inc", "../../include/ccd.h", "../../include/ccd_scale_table.inc"]
LIB = os.path.join(_HERE, "libccd.so")


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X decoder cannot be built")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(_HERE, "csrc", s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def _flags():
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
             # every fused multiply-add of the float stages is an explicit __fmaf_rn (bit parity with the oracle)
             "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
    if os.environ.get("CCD_PIPE_PROFILE"):
        # cycle counters in the entropy kernel (ccd_batch_slot_stats): 1 = light (stalls, per-grid totals), 2 = every phase
        flags.append("-DCCD_PIPE_PROFILE=" + os.environ["CCD_PIPE_PROFILE"])
    flags += os.environ.get("CCD_EXTRA_FLAGS", "").split()  # e.g. -DCCD_T8=48 for tuning experiments
    return flags


LAST_BUILD = {"mode": "not run", "compiled": []}  # what the last build_lib() did (printed by __graft_entry__.build)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 ... -> cool_chic_amd/libccd.so (cross-compiles without a GPU).
    One object per source (compiled concurrently, rebuilt only when stale), then one link."""
    LAST_BUILD["compiled"] = []
    if not force and not is_stale():
        LAST_BUILD["mode"] = "up to date (prebuilt libccd.so newer than every source: nothing compiled)"
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    obj_dir = os.path.join(_HERE, "csrc", "_obj" + ("_prof" if os.environ.get("CCD_PIPE_PROFILE") else ""))
    os.makedirs(obj_dir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(_HERE, "csrc", h)) for h in HEADERS + ["../_build.py"])

    def compile_one(src):
        path = os.path.join(_HERE, "csrc", src)
        obj = os.path.join(obj_dir, src + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
            return obj
        cmd = [_hipcc()] + _flags() + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        LAST_BUILD["compiled"].append(src)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    LAST_BUILD["mode"] = "compiled %d of %d sources with hipcc --offload-arch=gfx950, linked" % (len(LAST_BUILD["compiled"]), len(SOURCES))
    return LIB


def build_variant(name: str, extra_flags: str) -> str:
    """A second copy of the library with extra compiler flags (e.g. -DCCD_FD_PROFILE), for tools: cool_chic_amd/libccd_<name>.so,
    selected at run time with CCD_LIB=<path>."""
    from concurrent.futures import ThreadPoolExecutor

    obj_dir = os.path.join(_HERE, "csrc", "_obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    out = os.path.join(_HERE, f"libccd_{name}.so")

    def compile_one(src):
        path = os.path.join(_HERE, "csrc", src)
        obj = os.path.join(obj_dir, src + ".o")
        if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), max(os.path.getmtime(os.path.join(_HERE, "csrc", h)) for h in HEADERS)):
            return obj
        subprocess.check_call([_hipcc()] + _flags() + extra_flags.split() + ["-c", path, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def list_variants(root=None) -> list:
    """Variant libraries (libccd_<name>.so) and object trees (csrc/_obj_<name>) that tools/ built with build_variant()."""
    import glob

    root = root or _HERE
    return sorted(glob.glob(os.path.join(root, "libccd_*.so")) + [p for p in glob.glob(os.path.join(root, "csrc", "_obj_*")) if os.path.isdir(p)])


def clean_variants(keep=(), root=None) -> list:
    """Removes the variant libraries (libccd_<name>.so) and object trees (csrc/_obj_<name>) that tools/ built with
    build_variant(), except the names in `keep`.  They are git-ignored but travel to the GPU box with every push (tens of MB),
    and a stale one selected through CCD_LIB would be timed in place of the product.  Returns what was removed.
    `root`: the package directory (tests point it at a scratch copy: the real one may hold variants somebody is about to use)."""
    import glob

    _HERE = root or globals()["_HERE"]
    gone = []
    for path in glob.glob(os.path.join(_HERE, "libccd_*.so")):
        if os.path.basename(path)[len("libccd_"):-len(".so")] not in keep:
            os.remove(path)
            gone.append(path)
    for path in glob.glob(os.path.join(_HERE, "csrc", "_obj_*")):
        if os.path.basename(path)[len("_obj_"):] not in keep and os.path.isdir(path):
            shutil.rmtree(path)
            gone.append(path)
    return gone


if __name__ == "__main__":
    import sys

    print(build_lib(force="--force" in sys.argv, verbose=True))
