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
    assert lib.ao_abi_version() == 2
    assert b"gfx950" in lib.ao_version()


def test_create_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from alpha_omok_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(9, 10, 5, games=1)


def test_host_thread_budget_follows_local_world_size():
    """ao_host_threads: the per-process pool for the per-move Dirichlet replay is hardware threads / LOCAL_WORLD_SIZE
    (torchrun's variable: the ranks of a node share the host), at most 32, at least 1; AO_HOST_THREADS overrides."""
    import subprocess
    import sys
    hw = os.cpu_count() or 1
    code = "from alpha_omok_amd import _lib; print(_lib.load(build_if_missing=False).ao_host_threads())"

    def ask(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("LOCAL_WORLD_SIZE", "AO_HOST_THREADS")}
        e.update(env)
        return int(subprocess.run([sys.executable, "-c", code], env=e, cwd=REPO, capture_output=True, text=True,
                                  check=True).stdout.strip().splitlines()[-1])

    assert ask() == max(1, min(32, hw))
    assert ask(LOCAL_WORLD_SIZE="8") == max(1, min(32, hw // 8))
    assert ask(LOCAL_WORLD_SIZE="8", AO_HOST_THREADS="3") == 3
