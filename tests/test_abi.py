"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares."""
import ctypes
import os
import re

from conftest import REPO


def test_library_exports_every_declared_symbol():
    from alpha_omok_amd import _lib, build
    path = build.build()
    assert os.path.exists(path)
    lib = _lib.load(build_if_missing=False)
    hdr = open(os.path.join(REPO, "include", "omok_hip.h")).read()
    declared = set(re.findall(r"\b(ao_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ao_engine", "ao_net", "ao_config"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "libomok_hip.so does not export %s" % name
    assert set(_lib.SYMBOLS) == declared
    assert lib.ao_abi_version() == 1
    assert b"gfx950" in lib.ao_version()


def test_create_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from alpha_omok_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(9, 10, 5, games=1)
