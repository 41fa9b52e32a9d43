"""CPU: the C-ABI library is built, loads, and exports every symbol include/alink_hip.h declares."""
import os
import re

import a_link_amd  # noqa: F401
from a_link_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(name="alink_hip.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(alink_[a-z0-9_]+)\s*\(", src)))


def _exported_symbols():
    """every defined dynamic symbol alink_* of the built library (nm -D; llvm-nm where binutils is absent)"""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.check_output([nm, "-D", "--defined-only", _abi.LIB_PATH]).decode()
    return sorted(set(ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("alink_")))


def test_library_exists_and_loads():
    assert os.path.exists(_abi.LIB_PATH), "run __graft_entry__.build() first"
    lib = _abi.load()
    assert lib.alink_version() >= 1


def test_every_header_symbol_is_exported_and_bound():
    lib = _abi.load()
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "library does not export %s" % s
        assert s in _abi.PROTOTYPES, "_abi.py has no prototype for %s" % s
    for s in _abi.PROTOTYPES:
        assert s in syms, "_abi.py binds %s which the header does not declare" % s


def test_no_export_without_a_declaration():
    """VERDICT r4 (weak #8): the library exported 23 `alink_debug_*` switches no header declared.  Every exported alink_*
    symbol must be declared — the product ABI in include/alink_hip.h, the A/B and diagnostic switches in
    include/alink_hip_debug.h (process-global, not thread-safe, stated there) — and neither header may declare a symbol the
    library lacks; the product header declares no debug switch and the ctypes table binds none."""
    exported = _exported_symbols()
    product, debug = _header_symbols(), _header_symbols("alink_hip_debug.h")
    assert not set(product) & set(debug)
    undeclared = [s for s in exported if s not in product and s not in debug]
    assert not undeclared, "exported but declared in no header: %s" % undeclared
    missing = [s for s in product + debug if s not in exported]
    assert not missing, "declared but not exported: %s" % missing
    assert all(s.startswith("alink_debug_") for s in debug) and not [s for s in product if s.startswith("alink_debug_")]
    assert not [s for s in _abi.PROTOTYPES if s.startswith("alink_debug_")]


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    with pytest.raises(_abi.AlinkError):
        _abi.init(0)
    from a_link_amd.head import DenseHead
    with pytest.raises(_abi.AlinkError):
        DenseHead(512)
    from a_link_amd import siamese
    with pytest.raises(_abi.AlinkError):
        siamese.ArcFace((112, 112), "synthetic:r18")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "a-link_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py") or f.endswith(".hip") or f.endswith(".h"):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_no_convolution_kernel_spills_to_scratch():
    """The build keeps hipcc's per-kernel resource report beside every object (a-link_amd/lib/obj/<file>.usage).  No
    instantiation of the convolution kernels may use scratch: a spill inside a K loop is silent and costly (round 3: a
    run-time branch added to the split-precision form of conv3x3_linear spilled 120 registers and cost its dominant
    kernel 8 %)."""
    obj = os.path.join(ROOT, "a-link_amd", "lib", "obj")
    seen = 0
    for name in ("conv3x3_linear", "conv3x3_lat", "conv3x3_direct", "conv3x3_c64", "conv3x3_s2c64", "front_c64", "conv_igemm", "stem_tail"):
        path = os.path.join(obj, name + ".usage")
        assert os.path.exists(path), "build with the Makefile (it writes %s)" % path
        fn = None
        for ln in open(path):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                fn = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", ln)
            if m and fn and "kernel" in fn:
                seen += 1
                assert int(m.group(1)) == 0, "%s uses %s bytes of scratch per lane" % (fn, m.group(1))
    assert seen >= 60, seen
