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
