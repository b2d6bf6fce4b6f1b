"""CPU: the C-ABI library builds, loads, and exports every symbol include/nerfmeshes_hip.h declares."""
import os
import re

import pytest

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
    assert lib.nm_abi_version() == 6


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


def test_argument_errors_are_reported_without_a_gpu():
    """Error convention of the boundary (int status, 0 = ok, + nm_last_error()): bad arguments are
    rejected BEFORE any HIP call, so the checks run on a host without a GPU -- null buffers, sizes the kernels do not
    support, and skimage's own message for a volume smaller than 2x2x2 (what the Python shim turns into its ValueError)."""
    import ctypes as C
    lib = _lib.load()
    null = C.c_void_p(None)
    one = C.c_void_p(8)                       # never dereferenced: validation fails first

    def err():
        return (lib.nm_last_error() or b"").decode()

    v, f = C.c_int64(), C.c_int64()
    assert lib.nm_mc_count(null, 4, 4, 4, 0.5, one, C.byref(v), C.byref(f), null) == 2 and "bad argument" in err()
    assert lib.nm_mc_count(one, 1, 4, 4, 0.5, one, C.byref(v), C.byref(f), null) == 2
    assert "Input array must be at least 2x2x2." in err()
    assert lib.nm_mc_workspace_bytes(1, 4, 4) == 0 and lib.nm_mc_workspace_bytes(480, 480, 480) > 480 ** 3
    assert lib.nm_buff_intersect_ex(null, 8, one, 0, one, 0.0, 1.0, one, 4, 16, 0, one, one, one, null) == 2
    assert lib.nm_buff_intersect_ex(one, 8, one, 0, one, 0.0, 1.0, one, 4, 513, 0, one, one, one, null) == 2
    assert "samples must be in [1, 512]" in err()
    assert lib.nm_buff_intersect_ex(one, 8, one, 0, one, 0.0, 1.0, one, 4, 16, 7, one, one, one, null) == 2
    assert "unknown tie order" in err()
    assert lib.nm_buff_intersect_random(one, 8, one, 0, one, 0.0, 1.0, null, one, 4, 16, one, one, one, null) == 2
    assert lib.nm_buff_intersect_random(one, 8, one, 0, one, 0.0, 1.0, one, one, 4, 0, one, one, one, null) == 2
    assert lib.nm_tree_integrate(one, one, one, 10, 0, 1, one, one, null) == 2 and "bad sizes" in err()
    assert lib.nm_export_obj(None, 3, None, 0, None, 0, None, 0, b"/nonexistent-dir/x.obj") == 2 and "null array" in err()
    ok = (C.c_float * 3)(1.0, 2.0, 3.0)
    assert lib.nm_export_obj(ok, 1, None, 0, None, 0, None, 0, b"/nonexistent-dir/x.obj") == 6 and "cannot open" in err()


def test_ctypes_structs_match_the_header(tmp_path):
    """The ctypes mirrors of the structs that cross the boundary by pointer (nerfmeshes_amd/_lib.py) against what a C compiler makes
    of include/nerfmeshes_hip.h: sizes and the offsets of the members added last (nm_mlp_tape.v_stride and nm_mlp_param_grads, ABI
    v6: the 64-wide networks' fused backward)."""
    import ctypes as C
    import shutil
    import subprocess
    from nerfmeshes_amd import _lib
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nerfmeshes_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(nm_mlp_tape), offsetof(nm_mlp_tape, v_stride), '
                   'sizeof(nm_mlp_param_grads), offsetof(nm_mlp_param_grads, xyz_bias), offsetof(nm_mlp_param_grads, rgb_bias), '
                   'sizeof(nm_mlp_deltas), sizeof(nm_weight_grad_job)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.run([cc, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(_lib.MlpTape), _lib.MlpTape.v_stride.offset, C.sizeof(_lib.MlpParamGrads), _lib.MlpParamGrads.xyz_bias.offset,
            _lib.MlpParamGrads.rgb_bias.offset, C.sizeof(_lib.MlpDeltas), C.sizeof(_lib.WeightGradJob)]
    assert got == want, (got, want)
