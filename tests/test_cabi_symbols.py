"""The C-ABI library loads (no GPU needed) and exports every function include/esmk.h declares; the ctypes
table in esm_amd/_lib.py covers exactly the same set.  No compute calls here."""
import ctypes
import os
import re

from esm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "esmk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct \{.*?\} \w+;", "", src, flags=re.S)
    return set(re.findall(r"\b(esmk_\w+)\s*\(", src))


def test_header_symbols_exported():
    names = declared_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in esmk.h but not exported by libesmk.so"


def test_only_header_symbols_exported_and_no_env_switches():
    """The product library exports exactly the header's entry points and reads no environment variable; the
    experiments library exports the same ABI."""
    import subprocess
    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
        return {l.split()[-1] for l in out.splitlines() if l.strip()}
    names = declared_functions()
    assert exported(_lib.LIB_PATH) == names
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "getenv" not in und
    assert exported(_lib.XLIB_PATH) == names


def test_ctypes_table_matches_header():
    assert set(_lib.SIGNATURES) == declared_functions()


def test_version_and_host_only_calls():
    lib = _lib.lib()
    assert lib.esmk_version() == 300
    assert re.fullmatch(rb"[0-9a-f]{12}", lib.esmk_build_id())
    assert lib.esmk_packed_weight_bytes(3456, 1152) == 3584 * 1152 * 2  # rows padded to 256
    assert lib.esmk_packed_weight_bytes(11, 24) == 256 * 64 * 2
    assert lib.esmk_head_workspace_bytes(100, 128) >= 3 * 100 * 128 * 2
    # host-side planner: pure C, no device
    import numpy as np
    cu = np.array([0, 129, 129 + 64], dtype=np.int32)
    assert lib.esmk_attn_plan(cu.ctypes.data, 2, None, 0) == 3
    bad = np.array([0, 5, 3], dtype=np.int32)
    assert lib.esmk_attn_plan(bad.ctypes.data, 2, None, 0) == -1
    assert b"non-decreasing" in lib.esmk_last_error()


def test_struct_layout_matches_header(tmp_path):
    """sizeof/offsetof of every struct as gcc sees include/esmk.h == the ctypes mirror."""
    import subprocess
    structs = {"esmk_layer": _lib.EsmkLayer, "esmk_esm3_encoder": _lib.EsmkEsm3Encoder, "esmk_head": _lib.EsmkHead,
               "esmk_trunk": _lib.EsmkTrunk, "esmk_batch": _lib.EsmkBatch}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "esmk.h"', 'int main(void){']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
