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


def _struct_fields(text, name):
    """Field names of `typedef struct name { ... } name;` in the header, in declaration order."""
    body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(",")
        first = re.findall(r"[A-Za-z_][A-Za-z0-9_]*", names[0])[-1]
        fields.append(first)
        for extra in names[1:]:
            fields.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", extra)[-1])
    return fields


def test_ctypes_mirror_matches_the_header_layout(tmp_path):
    """the only consumer of include/sprl_amd.h used to be a hand-written ctypes mirror.  A C program
    compiled against the header prints sizeof / offsetof of every field of every struct; the ctypes mirror must agree."""
    text = open(os.path.join(ROOT, "include", "sprl_amd.h")).read()
    mirror = {"sprl_config": E.Config, "sprl_records": E.Records, "sprl_stats": E.Stats, "sprl_match_agent": E.MatchAgent}
    src = ['#include <stddef.h>', '#include <stdio.h>', '#include "sprl_amd.h"', "int main(void) {"]
    for name in mirror:
        src.append(f'    printf("{name} sizeof %zu\\n", sizeof({name}));')
        for f in _struct_fields(text, name):
            src.append(f'    printf("{name} {f} %zu\\n", offsetof({name}, {f}));')
    src += ["    return 0;", "}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(c)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    seen = 0
    for line in out.splitlines():
        name, field, value = line.split()
        cls = mirror[name]
        if field == "sizeof":
            assert C.sizeof(cls) == int(value), (name, C.sizeof(cls), value)
        else:
            assert getattr(cls, field).offset == int(value), (name, field, getattr(cls, field).offset, value)
        seen += 1
    assert seen > 80
    for name, cls in mirror.items():                       # and the mirror has no field the header lacks
        assert [f[0] for f in cls._fields_] == _struct_fields(text, name), name


def test_integration_md_snippets_compile(tmp_path):
    """The reference-side bindings shown in INTEGRATION.md are real code: every ```cpp block is compiled (syntax + types)
    against include/sprl_amd.h inside a function that declares the reference's local variables."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", md, flags=re.S)
    assert len(blocks) >= 2
    prologue = """
#include <cstdint>
#include <iostream>
#include <random>
#include <string>
#include <type_traits>
#include <vector>
struct OthelloNode {}; struct ConnectFourNode {};
"""
    for i, body in enumerate(blocks):
        includes = "\n".join(l for l in body.splitlines() if l.startswith(("extern", "#include", "}")) and "{" not in l.replace('extern "C" {', ""))
        code = "\n".join(l for l in body.splitlines() if not (l.startswith(("extern", "#include")) or l.strip() == "}"))
        src = prologue + 'extern "C" {\n#include "sprl_amd.h"\n}\n' + f"""
template <class ImplNode>
int snippet_{i}(int numGames, int numTraversals, int maxBatchSize, int maxQueueSize, float dirEps, float dirAlpha, int iter,
              std::string modelPath, std::string savePath, std::string modelPath0, std::string modelPath1,
              int model0UseSymmetrize, int model0UseParentQ, int model1UseSymmetrize, int model1UseParentQ) {{
    int numWins0 = 0, numWins1 = 0;
#define return return 0 +
{code}
#undef return
    return numWins0 + numWins1;
}}
template int snippet_{i}<OthelloNode>(int, int, int, int, float, float, int, std::string, std::string, std::string, std::string, int, int, int, int);
"""
        # `return;` inside the snippets (void context in the reference) becomes `return 0 + ;`: make that legal
        src = src.replace("return;", "return 0;").replace("return 1;", "return 1;")
        src = src.replace("#define return return 0 +\n", "").replace("#undef return\n", "")
        f = tmp_path / f"snippet_{i}.cpp"
        f.write_text(src)
        subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-variable",
                               "-I", os.path.join(ROOT, "include"), str(f)])


def test_busy_time_union_of_overlapping_launch_intervals(tmp_path):
    """sprl_amd/csrc/busy_log.h (bench.py's roofline accounting with several populations): the time with at least one launch
    executing is the length of the union of the launch intervals; with disjoint intervals it equals their sum."""
    src = tmp_path / "busy_test.cpp"
    src.write_text(r'''
#include "busy_log.h"
#include <cstdio>
#include <cmath>
int main() {
    busy::Log log;
    double sum = 0.0;
    // two streams: [0,2) [3,5) on one, [1,4) [10,11) on the other; a nested interval [3.5,3.6); an empty log first
    if (log.union_ms(&sum) != 0.0 || sum != 0.0) return 1;
    log.add({ 3.0, 5.0 }); log.add({ 0.0, 2.0 });
    if (std::fabs(log.union_ms(&sum) - 4.0) > 1e-12 || std::fabs(sum - 4.0) > 1e-12) return 2;       // disjoint: union == sum
    log.add(std::vector<std::pair<double, double>>{ { 1.0, 4.0 }, { 10.0, 11.0 }, { 3.5, 3.6 } });
    const double u = log.union_ms(&sum);
    if (std::fabs(u - 6.0) > 1e-12) return 3;              // [0,5) + [10,11)
    if (std::fabs(sum - 8.1) > 1e-12) return 4;
    log.reset();
    if (log.union_ms(nullptr) != 0.0) return 5;
    std::printf("ok\n");
    return 0;
}
''')
    exe = tmp_path / "busy_test"
    inc = os.path.join(ROOT, "sprl_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", inc, "-I", "/opt/rocm/include", "-o", str(exe), str(src),
                           "-lpthread"])
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip() == "ok"
