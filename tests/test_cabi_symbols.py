"""CPU: the C-ABI library builds, loads, and exports every symbol include/nerfmeshes_hip.h declares."""
import os
import re

from nerfmeshes_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nerfmeshes_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nm_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    build.build(verbose=False)
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert lib.nm_abi_version() == 1


def test_no_cpu_fallback_in_product():
    """The product package (and the helper scripts outside tests/) must not import / execute anything under oracle/."""
    bad = re.compile(r"^\s*(from\s+\.*oracle\b|import\s+oracle\b)|import_module\([\"']oracle|oracle/_ref|oracle\.", re.M)
    for dirpath, _, files in list(os.walk(os.path.join(ROOT, "nerfmeshes_amd"))) + list(os.walk(os.path.join(ROOT, "scripts"))):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
                code = re.sub(r'"""[\s\S]*?"""', "", code)
                assert not bad.search(code), f"{f} references the oracle"
