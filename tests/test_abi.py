"""The C-ABI boundary without a GPU: the gfx950 library loads, exports every symbol include/sprl_amd.h
declares, and refuses loudly to create an engine when no MI355X is present (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from sprl_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(E.DEFAULT_LIB):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sprl_amd", "csrc")])
    return E.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sprl_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sprl_[a-z_]+)\s*\(", text)) - {"sprl_forward_fn"})


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sprl_amd.h but not exported"


def test_library_contains_gfx950_code():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", E.DEFAULT_LIB], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("llvm-readelf not available")
    assert ".hip_fatbin" in out.stdout
    blob = open(E.DEFAULT_LIB, "rb").read()
    assert b"gfx950" in blob


def test_torch_plugin_exports(lib):
    path = os.path.join(os.path.dirname(E.DEFAULT_LIB), "libsprl_amd_torch.so")
    assert os.path.exists(path)
    import torch  # noqa: F401  (makes libtorch resolvable)
    p = C.CDLL(path)
    for n in ("sprl_torch_load", "sprl_torch_forward", "sprl_torch_free"):
        assert hasattr(p, n)


def test_default_config_matches_reference_constants(lib):
    c = E.default_config("othello", lib)
    assert (c.max_batch, c.max_queue, c.num_traversals, c.concurrent_games) == (8, 4, 800, 4096)
    assert abs(c.dir_alpha - 0.3) < 1e-7 and abs(c.dir_eps - 0.25) < 1e-7 and abs(c.u_weight - 1.1) < 1e-7
    assert (c.early_cutoff, c.use_symmetry, c.add_noise, c.mask_frame) == (15, 1, 1, 0)
    c4 = E.default_config("c4", lib)
    assert abs(c4.dir_alpha - 0.5) < 1e-7


def test_no_cpu_fallback(lib):
    if lib.sprl_device_available():
        pytest.skip("a GPU is present")
    with pytest.raises(E.SprlError) as ei:
        E.Engine(E.default_config("othello", lib, concurrent_games=1), lib)
    assert ei.value.code == -4 and "gfx950" in str(ei.value)


def test_product_never_references_oracle():
    """Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may touch oracle/."""
    pkg = os.path.join(ROOT, "sprl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "pyoracle" not in text and "sprl_oracle" not in text, f
