"""CPU: the C-ABI library loads, exports every symbol include/mmsurv.h declares, and its struct layouts
match the host's parsed view of the header (no compute calls)."""
import ctypes
import os

import pytest

from multimodal_survival_prediction_amd import _build, _lib


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.lib_path()):
        _build.build()
    return _lib.load_library()


def test_exports_every_declared_symbol(lib):
    protos = _lib.protos()
    assert len(protos) >= 10
    for name in protos:
        assert hasattr(lib, name), name


def test_struct_layouts_match(lib):
    for name, st in _lib.structs().items():
        assert lib.mms_abi_sizeof(name.encode()) == ctypes.sizeof(st), name
    assert lib.mms_abi_sizeof(b"NoSuchStruct") == -1
    assert lib.mms_abi_version() >= 1
