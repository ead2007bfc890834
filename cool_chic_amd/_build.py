"""Builds libccd.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["ccd_format.cpp", "ccd_writer.cpp", "ccd_api.cpp", "ccd_entropy.hip", "ccd_entropy_pipe.hip", "ccd_float.hip", "ccd_synth_fused.hip", "ccd_inter.hip"]
HEADERS = ["ccd_format.hpp", "ccd_device.hpp", "../../include/ccd.h", "../../include/ccd_scale_table.inc"]
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


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 ... -> cool_chic_amd/libccd.so (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           # every fused multiply-add of the float stages is an explicit __fmaf_rn (bit parity with the oracle)
           "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-o", LIB]
    if os.environ.get("CCD_PIPE_PROFILE"):
        cmd.append("-DCCD_PIPE_PROFILE")  # cycle counters in the entropy kernel (ccd_batch_slot_stats)
    cmd += [os.path.join(_HERE, "csrc", s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
