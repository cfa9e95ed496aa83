"""The C-ABI library loads and exports every symbol include/c2ray_hip.h declares; without a GPU it
refuses to create a context (there is no CPU path)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "c2ray_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(c2r_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(pkg):
    assert declared_symbols() == sorted(pkg._lib.SYMBOLS)


def test_library_exports_every_declared_symbol(pkg):
    lib = C.CDLL(str(pkg.build()))
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_no_device_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.C2RayHipError, match="no HIP device|no CPU path"):
        pkg.HipEngine((16, 16, 16))


def test_product_does_not_touch_the_oracle():
    """Nothing under the package may import, link or name the oracle."""
    for p in (ROOT / "c2-ray3dm1d_helium_amd").rglob("*"):
        if p.suffix in {".py", ".hip", ".hpp", ".h", ".F90", ".f90"}:
            t = p.read_text()
            assert "liboracle" not in t and "import oracle" not in t and "c2ray_oracle" not in t, p
