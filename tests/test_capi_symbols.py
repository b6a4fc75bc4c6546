"""CPU tier: libgsf.so builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports every symbol that
include/gsf.h declares; no compute call is made.  Also: the product fails loudly without a device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from gps_optimize_slam_amd import _lib
    if not os.path.exists(_lib.library_path()):
        _lib.build_library()
    return _lib


def declared_functions():
    src = open(os.path.join(ROOT, "include", "gsf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    L = C.CDLL(lib.library_path())
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/gsf.h but not exported by libgsf.so"
    assert set(names) == set(lib.SIGNATURES), set(names) ^ set(lib.SIGNATURES)


def test_version_and_abi(lib):
    L = lib.load()
    assert L.gsf_abi_version() == 1
    assert b"gfx950" in L.gsf_version()


def test_config_struct_layout(lib, tmp_path):
    # gsf_ekf_config: 7+7+3 doubles, 1 double, 2 int32
    assert C.sizeof(lib.EkfConfig) == 17 * 8 + 8 + 8
    # every POD of the boundary: size and the offset of every field as gcc lays the header's struct out == the ctypes mirror in _lib.py
    import subprocess
    structs = {"gsf_ekf_config": lib.EkfConfig, "gsf_prefilter_config": lib.PrefilterConfig, "gsf_run_config": lib.RunConfig}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gsf.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in subprocess.check_output([str(exe)], text=True).splitlines()}
    for cname, cls in structs.items():
        assert got[(cname, "size")] == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_fails_loudly_without_gpu(lib):
    L = lib.load()
    if L.gsf_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(lib.GsfError):
        lib.Context(0)
    from gps_optimize_slam_amd import ekfgpsslam as E
    import numpy as np
    with pytest.raises(lib.GsfError):
        E.compute_sim3_transform(np.random.rand(5, 3), np.random.rand(5, 3))
    h = C.c_void_p()
    assert L.gsf_create(0, C.byref(h)) == 3          # GSF_ERR_NO_DEVICE
    assert "no HIP device" in lib.last_error()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gps_optimize_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "gsf_oracle" not in txt, f
    # ... and nothing outside tests/ does either, except the two places the rules name: smoke() and bench.py's cpu_baseline leg / gates
    for top in ("tools", "examples", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".sh", ".c", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dirpath, f), errors="replace").read()
                    assert "import oracle" not in txt and "from oracle" not in txt and "libgsf_oracle" not in txt, os.path.join(top, f)
