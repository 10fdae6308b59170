"""Select which build of the C-ABI library a test drives.

* ``emu``  -- tests/hipemu: the SAME kernel sources compiled for the CPU fiber emulator
               (runs in the GPU-less build container; marker `emu`)
* ``gpu``  -- virtex_amd/lib/libvirtex_amd.so on a real MI355X (marker `gpu`)
"""
import os

import pytest
import torch

from virtex_amd import _lib, build

BACKENDS = [pytest.param("emu", marks=pytest.mark.emu), pytest.param("gpu", marks=pytest.mark.gpu)]


def select(backend: str) -> torch.device:
    if backend == "emu":
        path = build.EMU_LIB_PATH
        if os.environ.get("VTX_EMU_LIB"):
            # an instrumented build of the emulator library (AddressSanitizer: tools/build_emu_asan.sh), used as it is
            path = os.environ["VTX_EMU_LIB"]
        elif not os.path.exists(path) or os.environ.get("VTX_REBUILD_EMU", "1") == "1":
            path = build.build_emu()
        _lib.use_library(path)
        assert _lib.is_emulator()
        return torch.device("cpu")
    _lib.use_library(_lib.DEFAULT_LIB)
    assert _lib.backend() == "hip:gfx950"
    return torch.device("cuda:0")


def tol(dtype):
    """fp32 kernels: the north-star bound 1e-3 relative; bf16 storage: 8-bit mantissa."""
    return dict(rtol=1e-3, atol=1e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)


def rel_err(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
