"""ISA checks on the CPU build (hipcc cross-compiles gfx950 without a GPU).

DESIGN.md section 4.2: on gfx950 a 16-byte buffer store whose scalar offset is an SGPR, followed at once by a VALU write of its
data registers, corrupted the stored data (LLVM's createsVALUHazard skips the wait state exactly in that case).  The kernels
use soffset 0 + immediate offsets, for which the compiler inserts the s_nop.  Only the numeric convolution tests on the GPU
would notice if a toolchain change broke that; this test pins it in the assembly."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sprl_amd", "csrc")


def _device_asm(src, tmp_path, extra=()):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / (os.path.basename(src) + ".s"))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", *extra, "--cuda-device-only", "-S", "-o", out, src],
                          stderr=subprocess.DEVNULL)
    return [ln.strip() for ln in open(out) if ln.strip() and not ln.strip().startswith((";", ".", "//"))]


def _regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.parametrize("src,flags", [("cnn_wino.hip", ("-fno-slp-vectorize",)), ("cnn_epilogue.hip", ())])
def test_wide_buffer_stores_keep_their_wait_state(src, flags, tmp_path):
    lines = _device_asm(os.path.join(CSRC, src), tmp_path, flags)
    stores = 0
    for i, ln in enumerate(lines):
        if not ln.startswith(("buffer_store_dwordx4", "buffer_store_dwordx3")):
            continue
        stores += 1
        ops = [t.strip() for t in ln.split(None, 1)[1].split(",")]
        data = _regs(ops[0])
        soffset = ops[3].split()[0]
        assert not re.fullmatch(r"s\d+", soffset), f"{src}: scalar-register soffset on a wide store (gfx950 hazard): {ln}"
        nxt = lines[i + 1] if i + 1 < len(lines) else ""
        if nxt.startswith("v_") and not nxt.startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
            dest = _regs(nxt.split(None, 1)[1].split(",")[0].strip())
            assert not (dest & data), f"{src}: VALU write of a store's data registers with no wait state:\n  {ln}\n  {nxt}"
    if src == "cnn_wino.hip":
        assert stores >= 16          # the 8x8 kernel's output rows are 16-byte buffer stores
