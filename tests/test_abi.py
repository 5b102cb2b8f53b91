"""CPU: the C-ABI library loads without a GPU and exports exactly what include/occ4d.h declares;
the ctypes table (occlusions-4d_amd/_lib.py) names the same symbols.  No compute calls here."""
import ctypes
import os
import re

import pytest

import occlusions4d_amd as pk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'occ4d.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(occ4d_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_entry_points():
    syms = header_symbols()
    assert 'occ4d_knn_f32' in syms and 'occ4d_linear_f32' in syms and 'occ4d_fps_f32' in syms
    assert len(syms) >= 15


def test_ctypes_table_matches_header():
    assert sorted(pk._lib.SIGNATURES) == header_symbols()


def test_library_loads_and_exports_every_symbol():
    lib = pk._lib.lib()           # raises NativeLibraryError if absent/stale
    raw = ctypes.CDLL(pk._lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(raw, name), name
    assert lib.occ4d_abi_version() == pk._lib.ABI_VERSION


def test_linear_args_struct_layout():
    a = pk._lib.LinearArgs
    assert ctypes.sizeof(a) == 144
    assert a.M.offset == 72 and a.add_rows.offset == 96 and a.sub_idx.offset == 136


def test_cpu_tensors_are_rejected_loudly():
    import torch
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pk.ops.linear(torch.zeros(4, 8), torch.zeros(4, 8))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pk.point_transformer_layer.kNN_torch(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3), 2)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(pk._lib, '_lib', None)
    monkeypatch.setattr(pk._lib, 'LIB_PATH', str(tmp_path / 'libocc4d.so'))
    with pytest.raises(pk._lib.NativeLibraryError, match='no CPU/PyTorch fallback'):
        pk._lib.lib()
