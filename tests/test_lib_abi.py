"""CPU-side checks of the C-ABI library: builds for gfx950, loads, exports every declared symbol."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    syms = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        txt = open(os.path.join(ROOT, "include", fn)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        syms |= set(re.findall(r"\b((?:tw|ppo|mg)_[a-z0-9_]+)\s*\(", txt))
    return syms


def test_library_loads_and_exports_all_declared_symbols():
    import __graft_entry__ as ge
    ge.build()
    import twoarmy_amd
    lib = twoarmy_amd._lib.lib()
    declared = _declared()
    assert "tw_step" in declared and "tw_rollout" in declared
    for s in sorted(declared):
        assert hasattr(lib, s), "libtwoarmy_hip.so lacks %s declared in include/" % s
    assert set(twoarmy_amd._lib.exported_symbols()) == declared
    assert lib.tw_version().startswith(b"twoarmy-hip")


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import twoarmy_amd
    from twoarmy_amd.engine import TwoarmyEngine
    with pytest.raises(twoarmy_amd._lib.TwoarmyLibraryError):
        TwoarmyEngine(6, 4)


def test_record_layout_matches_header():
    import twoarmy_amd
    txt = open(os.path.join(ROOT, "include", "twoarmy.h")).read()
    body = re.search(r"enum tw_field \{(.*?)\};", txt, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    val, got = -1, {}
    for item in body.split(","):
        item = item.strip()
        if not item:
            continue
        if "=" in item:
            name, v = [s.strip() for s in item.split("=")]
            val = int(v)
        else:
            name, val = item, val + 1
        got[name[3:]] = val
    assert got == twoarmy_amd._lib.FIELDS


def test_ctypes_struct_mirrors_the_header():
    """struct tw_outputs as gcc lays it out from include/twoarmy.h == the ctypes mirror in _lib.py (size and offsets)."""
    import ctypes as C
    import subprocess
    import tempfile
    import twoarmy_amd
    fields = [f for f, _ in twoarmy_amd._lib.TwOutputs._fields_]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "twoarmy.h"\nint main(void){printf("%zu", sizeof(tw_outputs));' + \
          "".join('printf(" %%zu", offsetof(tw_outputs, %s));' % f for f in fields) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "a.c"), os.path.join(d, "a.out")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        vals = [int(v) for v in subprocess.check_output([exe]).split()]
    T = twoarmy_amd._lib.TwOutputs
    assert vals[0] == C.sizeof(T)
    assert vals[1:] == [getattr(T, f).offset for f in fields]
