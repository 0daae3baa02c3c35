"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol the header declares."""
import ctypes
import re

import pytest

import helpers as H


def header_functions():
    text = (H.ROOT / "include" / "softmac_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smac_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(hip_lib):
    from softmac_amd import _ffi
    names = header_functions()
    assert len(names) >= 40
    assert set(names) == set(_ffi.SIGNATURES), set(names) ^ set(_ffi.SIGNATURES)
    for n in names:
        assert hasattr(hip_lib, n), n                     # exported symbol


def test_abi_version_and_config_layout(hip_lib):
    from softmac_amd import _ffi
    assert hip_lib.smac_abi_version() == _ffi.ABI_VERSION
    assert ctypes.sizeof(_ffi.SmacConfig) == 18 * 4 + 10 * 8     # 18 int32 + 10 doubles, no padding surprises


def test_no_cpu_fallback(hip_lib):
    """Without a GPU the product refuses to run instead of silently computing on the host."""
    if hip_lib.smac_device_count() > 0:
        pytest.skip("GPU present")
    from softmac_amd._ffi import SmacError
    from softmac_amd.engine.mpm_simulator import MPMSimulator
    with pytest.raises(SmacError, match="no HIP device"):
        MPMSimulator(H.sim_cfg(100, n_grid=32), (), 1e-3)


def test_product_does_not_reference_the_oracle():
    for p in (H.ROOT / "softmac_amd").rglob("*"):
        if p.suffix in (".py", ".hpp", ".hip", ".h", ".cpp"):
            text = p.read_text()
            assert "import oracle" not in text and "from oracle" not in text and "oracle/" not in text, p
