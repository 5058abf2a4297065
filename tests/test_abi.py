"""CPU tests: the C-ABI shared library loads and exports every symbol include/mcl3dl_hip.h declares, the ctypes binding
covers all of them, and nothing falls back to a CPU path when no GPU is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mcl3dl_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"static inline[^{;]*\{.*?\n\}", "", text, flags=re.S)   # header-only helpers are not exports
    return sorted(set(re.findall(r"\b(mcl3dl_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("mcl3dl_hip_create", "mcl3dl_hip_set_map", "mcl3dl_hip_measure_batch", "mcl3dl_hip_pf_measure",
                 "mcl3dl_hip_measure_update", "mcl3dl_hip_beam_status", "mcl3dl_hip_measure_device"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from mcl_3dl_amd import capi
    assert os.path.exists(capi.LIB_PATH), "libmcl3dl_hip.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), "missing export: " + name


def test_binding_covers_every_declared_symbol():
    from mcl_3dl_amd import capi
    assert sorted(capi.SIGNATURES) == declared_symbols()
    lib = capi.load_library()
    assert lib.mcl3dl_hip_abi_version() == 3


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU, creating an engine must raise — never silently compute on the CPU."""
    import torch
    from mcl_3dl_amd import capi
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the failure path is exercised on CPU-only boxes")
    with pytest.raises(capi.EngineError):
        capi.Engine(0)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "mcl_3dl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "pyoracle" not in text.replace("never imports anything from oracle/", ""), f
                assert "libmcl3dl_oracle" not in text and "libmcl3dl_ref" not in text, f


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 (-pedantic) and a C program must link against the library
    and get the documented failure from mcl3dl_hip_create on a box without a GPU (or success with one)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "mcl3dl_hip.h")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    src = tmp_path / "probe.c"
    src.write_text('#include "mcl3dl_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { mcl3dl_hip_ctx* c = NULL; int rc = mcl3dl_hip_create(&c, 0);\n'
                   '  printf("%d %d\\n", mcl3dl_hip_abi_version(), rc); if (rc == 0) mcl3dl_hip_destroy(c); return 0; }\n')
    exe = str(tmp_path / "probe")
    libdir = os.path.join(root, "mcl_3dl_amd")
    subprocess.run([gcc, "-std=c99", "-I", os.path.join(root, "include"), "-o", exe, str(src), "-L", libdir, "-lmcl3dl_hip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    abi, rc = (int(x) for x in out.stdout.split())
    assert abi == 3 and rc <= 0


def test_every_option_is_in_the_api_fuzz_pool():
    """no runtime switch without a test that draws it in combination with the others. Every key
    mcl3dl_hip_set_option accepts (parsed from api_support.inl) is in tests/test_gpu_api_fuzz.py's CHOICES — the fault-injection
    hook aside — and can be read back (the getter table)."""
    import ast
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "mcl_3dl_amd", "csrc", "api_support.inl")).read()
    a = src.index("int mcl3dl_hip_set_option(")
    b = src.index("kOptionGetters[]")
    settable = set(re.findall(r'key == "([a-z_0-9]+)"', src[a:b]))
    readable = set(re.findall(r'\{ "([a-z_0-9]+)", \[\]', src[b:]))
    fuzz = open(os.path.join(root, "tests", "test_gpu_api_fuzz.py")).read()
    tree = ast.parse(fuzz)
    pools = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") in ("DEFAULTS", "CHOICES"):
            pools[node.targets[0].id] = {kw.arg for kw in node.value.keywords}
    assert pools["DEFAULTS"] == pools["CHOICES"]
    assert settable - {"test_late_structures"} == pools["CHOICES"], (sorted(settable - pools["CHOICES"]), sorted(pools["CHOICES"] - settable))
    assert settable - {"test_late_structures"} <= readable, sorted(settable - readable)
    assert len(settable) <= 40
