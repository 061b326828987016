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


def _kernels(src, tmp_path, extra=()):
    """{mangled kernel name: (lines of its body, {'vgprs', 'scratch'})} of a HIP source compiled for gfx950"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / (os.path.basename(src) + ".full.s"))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", *extra, "--cuda-device-only", "-S", "-o", out, src],
                          stderr=subprocess.DEVNULL)
    kernels, name, body = {}, None, []
    for ln in open(out):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        body.append(ln.rstrip("\n"))
        m = re.match(r"^; ScratchSize: (\d+)", ln)
        if m:
            vg = [int(x.split(":")[1]) for x in body if x.startswith("; NumVgprs:")]
            kernels[name] = (body, {"vgprs": vg[-1] if vg else -1, "scratch": int(m.group(1))})
            name = None
    return kernels


def test_trunk_convolution_k_loop_shape(tmp_path):
    """DESIGN.md section 5 ("Where a wave of the 8x8 kernel spends its life"): what the last form of the K loop rests on, pinned in
    the assembly of the product build - a toolchain change that undid one of these would only show as a few per cent in the bench.
    (1) every 8x8 variant fits two waves per SIMD (<= 256 registers) with at most 12 bytes of scratch, none of it inside the phase
    loop; (2) the phase loop requests its activation chunk WITHOUT a branch around the two loads, so the K step's filter waits are
    counted exactly (vmcnt 10, 9, ... with the request in front of them) instead of two too small; (3) the B operands are read by
    the rolling asm reads (ds_read2st64 between #ASMSTART / #ASMEND, 18 per K step), not by the compiler's batches of eight."""
    ks = _kernels(os.path.join(CSRC, "cnn_wino.hip"), tmp_path, ("-fno-slp-vectorize",))
    eight = {n: v for n, v in ks.items() if "wino_conv64_kernelILi8ELi8E" in n}
    assert len(eight) == 5, sorted(eight)
    for n, (body, res) in eight.items():
        assert 0 < res["vgprs"] <= 256 and res["scratch"] <= 12, (n, res)
    plain = next(v for n, v in eight.items() if "ILi8ELi8ELi0ELi0ELi0E" in n)[0]
    heads = [i for i, ln in enumerate(plain) if "Inner Loop Header" in ln]
    assert heads, "the phase loop of the plain variant is expected to stay a rolled loop"
    header = plain[heads[-1]].split(":")[0].strip()         # (earlier inner loops: the LDS zero fill); e.g. ".LBB2_11"
    tag = "Header=" + header[2:]                            # blocks of the (rotated) loop carry "in Loop: Header=BB2_11"
    start = min([heads[-1]] + [i for i, ln in enumerate(plain) if tag in ln])
    labels = {ln.split(":")[0].strip(): i for i, ln in enumerate(plain) if re.match(r"^\.LBB\d+_\d+:", ln)}
    back = [i for i, ln in enumerate(plain) if i > start and re.search(r"s_cbranch_\w+\s+\.LBB\d+_\d+", ln) and
            start <= labels.get(ln.split()[-1], -1) <= i]
    assert back, "no backward branch found for the phase loop of the plain variant"
    end = max(back)
    loop = [ln.strip() for ln in plain[start:end + 1]]
    assert not any("scratch_" in ln for ln in loop), "spill inside the phase loop"
    mfma = [i for i, ln in enumerate(loop) if ln.startswith("v_mfma_f32_16x16x4")]
    assert len(mfma) == 72, len(mfma)                       # two K steps of 36
    nt_loads = [i for i, ln in enumerate(loop) if ln.startswith("buffer_load_dwordx4") and ln.endswith("nt")]
    assert len(nt_loads) == 2 and nt_loads[1] == nt_loads[0] + 1 or len(nt_loads) == 2 and nt_loads[1] - nt_loads[0] <= 2, nt_loads
    # no branch between the loop head and the request, and none between the request and the first MFMA behind it
    first_mfma_after = next(i for i in mfma if i > nt_loads[1])
    assert not any(ln.startswith("s_cbranch") for ln in loop[nt_loads[0] - 3:first_mfma_after]), loop[nt_loads[0] - 3:first_mfma_after]
    # the filter waits of the K step behind the request count the two loads of the request: the first one is vmcnt(10)
    waits = [int(m.group(1)) for ln in loop[nt_loads[1]:first_mfma_after + 1] for m in [re.search(r"vmcnt\((\d+)\)", ln)] if m]
    assert waits and waits[0] == 10, waits
    asm_reads = sum(1 for i, ln in enumerate(loop) if ln.startswith("ds_read2st64_b32") and "ASMSTART" in loop[i - 1])
    assert asm_reads == 36, asm_reads                       # 18 pairs per K step, two K steps
