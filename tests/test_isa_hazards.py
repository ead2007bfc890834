"""Build-time checks of the ISA around hand-written asm the compiler cannot see into (r05 advisor, ccd_entropy.hip).

The generic entropy kernel requests the next pixel's two table rows with a bare `ds_read_b32` pair in one asm statement and waits
for them (`s_waitcnt lgkmcnt(0)`) in another one at the end of the iteration: the compiler does not count LDS operations issued
from inline asm, so any copy / spill of the two destination registers it placed between the statements would read them before
the data has landed - silently wrong symbols, with this compiler version or the next.  The test compiles the file to assembly
(hipcc cross-compiles without a GPU, ~1 s) and requires that NOTHING between the reads and the first wait behind them names
the two registers; the same for the -DCCD_GEN_PROFILE variant tools/ build."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cool_chic_amd", "csrc", "ccd_entropy.hip")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


@pytest.mark.parametrize("extra", [[], ["-DCCD_GEN_PROFILE"]])
def test_generic_kernel_prefetched_rows_are_not_touched_before_the_wait(tmp_path, extra):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    out = tmp_path / "gen.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
                           "-S", "--cuda-device-only", SRC, "-o", str(out)] + extra, stderr=subprocess.DEVNULL)
    lines = out.read_text().splitlines()
    found = 0
    i = 0
    while i < len(lines) - 1:
        m0 = re.match(r"\s*ds_read_b32 (v\d+), (v\d+)\s*$", lines[i])
        m1 = re.match(r"\s*ds_read_b32 (v\d+), (v\d+) offset:256\s*$", lines[i + 1])
        if not (m0 and m1 and m0.group(2) == m1.group(2)):
            i += 1
            continue
        found += 1
        regs = {m0.group(1), m1.group(1)}
        j = i + 2
        while j < len(lines) and not re.match(r"\s*s_waitcnt (vmcnt\(\d+\) )?lgkmcnt\(0\)", lines[j]):
            code = lines[j].split(";")[0]
            if not code.strip().startswith("."):
                for r in regs:
                    # the register alone or inside a range v[a:b]
                    n = int(r[1:])
                    hit = re.search(r"\b%s\b" % r, code) or any(int(a) <= n <= int(b) for a, b in re.findall(r"v\[(\d+):(\d+)\]", code))
                    assert not hit, f"{os.path.basename(SRC)}{extra}: line {j + 1} touches {r} between the prefetch and its s_waitcnt: {lines[j].strip()}"
            j += 1
        assert j < len(lines), "no s_waitcnt lgkmcnt(0) behind the prefetch"
        i = j
    assert found >= 1, "the prefetch pair was not found in the assembly (did the kernel change? update this test)"
