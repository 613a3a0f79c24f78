"""Host-side logic that needs no GPU: checkpoint layout, key tables, ABI surface, sharding maths."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_layout_matches_spec():
    import importlib
    from e2fgvi_amd.synth import state_spec, synth_state_dict
    for name, n in (("e2fgvi", 243), ("e2fgvi_hq", 244)):
        net = importlib.import_module("model." + name).InpaintGenerator()
        sd = net.state_dict()
        spec = state_spec(name)
        assert len(sd) == n
        assert list(sd.keys()) == list(spec.keys())
        for k, (shape, dt) in spec.items():
            assert tuple(sd[k].shape) == tuple(shape) and sd[k].dtype == dt, k
        net.load_state_dict(synth_state_dict(name, "stress", 0), strict=True)
        assert sum(v.numel() for v in sd.values() if v.is_floating_point()) > 41e6


def test_default_init_distribution():
    """reference random init (e2fgvi.py:29-68,203-208): N(0,0.02) weights, zero biases, zero conv_offset[-1]"""
    import importlib
    torch.manual_seed(0)
    net = importlib.import_module("model.e2fgvi").InpaintGenerator()
    sd = net.state_dict()
    assert abs(sd["encoder.layers.8.weight"].std().item() - 0.02) < 1e-3
    assert sd["encoder.layers.8.bias"].abs().max() == 0
    assert sd["feat_prop_module.deform_align.forward_.conv_offset.6.weight"].abs().max() == 0
    w = sd["feat_prop_module.deform_align.forward_.weight"]
    assert abs(w.abs().max().item() - 1 / np.sqrt(256 * 9)) < 1e-4          # uniform(+-1/sqrt(C*9)) untouched
    assert abs(sd["transformer.3.pool_layers.0.weight"].std().item() - 0.02) < 8e-3   # overridden by init_weights
    assert sd["transformer.0.norm1.weight"].eq(1).all() and sd["sc.bias"].abs().max() == 0


def test_synth_is_deterministic_and_cpu_forward_refuses():
    import importlib
    from e2fgvi_amd.synth import synth_state_dict
    a, b = synth_state_dict("e2fgvi", "default", 0), synth_state_dict("e2fgvi", "default", 0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = synth_state_dict("e2fgvi", "default", 1)
    assert not torch.equal(a["decoder.2.weight"], c["decoder.2.weight"])
    net = importlib.import_module("model.e2fgvi").InpaintGenerator()
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 2, 3, 240, 432), 2)       # no CPU path: must fail loudly


def _reference_key_ids(fh, fw, T=1):
    """Enumerate, with torch ops on an id tensor, the keys each window sees -- the reference's roll /
    window_partition / valid_ind_rolled / unfold chain (tfocal_transformer.py:235-333)."""
    from oracle import e2fgvi_oracle as O
    ids = torch.arange(fh * fw, dtype=torch.float32).view(1, 1, fh, fw, 1)
    own = O._win_part(ids, O.WIN).view(-1, 45)
    valid = O.rolled_valid_index()
    parts = []
    for sy, sx in ((-2, -4), (-2, 4), (2, -4), (2, 4)):
        parts.append(O._win_part(torch.roll(ids, shifts=(sy, sx), dims=(2, 3)), O.WIN).view(-1, 45))
    rolled = torch.cat(parts, 1)[:, valid]
    nwh, nww = fh // 5, fw // 9
    pid = torch.arange(1, nwh * nww + 1, dtype=torch.float32).view(1, 1, nwh, nww)
    un = torch.nn.functional.unfold(pid, kernel_size=(5, 9), padding=(2, 4)).view(45, nwh * nww).t()   # 0 = padded slot
    return own.long(), rolled.long(), un.long()


@pytest.mark.parametrize("fh,fw", [(5, 9), (10, 18), (20, 36), (15, 45), (60, 108)])
def test_key_table_matches_reference_enumeration(fh, fw):
    from e2fgvi_amd.engine import build_key_table
    from e2fgvi_amd.synth import rolled_valid_index
    from oracle import e2fgvi_oracle as O
    assert torch.equal(rolled_valid_index(), O.rolled_valid_index())
    tab, nk = build_key_table(fh, fw, rolled_valid_index().tolist())
    own, rolled, un = _reference_key_ids(fh, fw)
    for w in range(tab.shape[0]):
        refs = tab[w, :nk[w]].tolist()
        toks = sorted(r for r in refs if r >= 0)
        pooled = sorted(-(r + 1) for r in refs if r < 0)
        assert toks == sorted(own[w].tolist() + rolled[w].tolist())          # multiset: duplicates must match
        assert pooled == sorted((un[w][un[w] > 0] - 1).tolist())
        assert nk[w] == 165 + int((un[w] > 0).sum())
    if fh >= 10 and fw >= 18:
        # the famous 12 duplicates among the 120 ring keys (SURVEY.md 8a trap 6)
        assert len(set(rolled[0].tolist())) == 108


def test_abi_exports_every_declared_symbol():
    """the C-ABI library loads without a GPU and exports exactly what include/e2fgvi_hip.h declares"""
    from e2fgvi_amd import lib
    hdr = open(os.path.join(ROOT, "include", "e2fgvi_hip.h")).read()
    declared = set(re.findall(r"\b(e2fgvi_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    if not os.path.exists(lib.LIB_PATH):
        pytest.skip("library not built yet (python -m e2fgvi_amd.build)")
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), name
    assert lib.load().e2fgvi_abi_version() == 1


def test_desc_struct_sizes_are_plain_c():
    from e2fgvi_amd import lib
    # pointers 8 bytes, int32 fields: sizes must be multiples of 8 and stable
    assert ctypes.sizeof(lib.ConvDesc) % 8 == 0 and ctypes.sizeof(lib.MdcnDesc) % 8 == 0


def test_missing_library_fails_loudly(monkeypatch):
    from e2fgvi_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libe2fgvi_hip.so")
    with pytest.raises(lib.HipLibraryMissing):
        lib.load()


def test_shard_range_partitions():
    from e2fgvi_amd.runner import shard_range
    for n in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flops_formula_matches_survey():
    import bench
    saved, bench.GFLOP_PER_CLIP = bench.GFLOP_PER_CLIP, {}
    try:
        assert abs(bench.flops_per_clip(10, 10) - 2039.1) < 0.5
        assert abs(bench.flops_per_clip(5, 5) - 932.5) < 0.5
        assert abs(bench.flops_per_clip(10, 5) - 2 * 859.4) < 0.5
    finally:
        bench.GFLOP_PER_CLIP = saved


def test_token_grid():
    from e2fgvi_amd.engine import token_grid
    assert token_grid(60, 108) == (20, 36) and token_grid(180, 324) == (60, 108) and token_grid(270, 486) == (90, 162)
