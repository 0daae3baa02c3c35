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


def test_rccl_not_found_is_an_error_message_not_a_crash(hip_lib):
    """the 'RCCL not found' path built its message from two dlerror() calls (the second returns NULL).  A forced library path
    that does not exist must come back as SMAC_ERR_INVALID with a message; run in a child so that this process keeps its own loader state."""
    import os
    import subprocess
    import sys
    code = (
        "import ctypes, sys\n"
        f"sys.path.insert(0, {str(H.ROOT)!r})\n"
        "from softmac_amd import _ffi\n"
        "lib = _ffi.load_library()\n"
        "buf = ctypes.create_string_buffer(128)\n"
        "rc = lib.smac_comm_unique_id(buf)\n"
        "msg = lib.smac_last_error(None)\n"
        "msg = msg.decode() if isinstance(msg, bytes) else str(msg)\n"
        "print(rc, '|', msg)\n"
    )
    env = dict(os.environ, SMAC_RCCL_LIB="/nonexistent/librccl-not-here.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rc, msg = out.stdout.strip().split("|", 1)
    assert int(rc) != 0 and "RCCL not found" in msg and "librccl-not-here" in msg, out.stdout
