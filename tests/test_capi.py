"""CPU checks of the C-ABI boundary: the library builds, loads, and exports every
symbol include/egt_amd.h declares; argument validation runs without a GPU."""
import ctypes as C
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(REPO, "include", "egt_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(egt_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ("egt_attn_fwd", "egt_attn_bwd", "egt_edge_proj_fwd", "egt_edge_proj_bwd", "egt_edge_proj_bwd_acc",
              "egt_edge_update_fwd", "egt_edge_update_bwd", "egt_block_fwd", "egt_block_bwd",
              "egt_last_error_string"):
        assert s in syms


def test_library_exports_every_declared_symbol(egt_lib):
    raw = C.CDLL(egt_lib._name)
    missing = [s for s in declared_symbols() if not hasattr(raw, s)]
    assert not missing, f"declared in include/egt_amd.h but not exported: {missing}"


def test_ctypes_table_matches_header(egt_lib):
    from egt_amd import _lib
    table = set(_lib._PROTOS) | set(_lib._OPTIONAL_PROTOS)
    assert set(declared_symbols()) == table


def test_argument_validation_without_gpu(egt_lib):
    from egt_amd import _lib as L
    d = L.AttnDesc(B=1, N=4, H=3, d=8, dtype=L.EGT_F32, flags=0)
    rc = egt_lib.egt_attn_fwd(C.byref(d), *([None] * 12))
    assert rc == L.EGT_E_SHAPE and b"power of two" in egt_lib.egt_last_error_string()
    d = L.AttnDesc(B=1, N=4, H=8, d=8, dtype=L.EGT_F32, flags=L.F_SCALE_DEGREE)
    assert egt_lib.egt_attn_fwd(C.byref(d), *([None] * 12)) == L.EGT_E_FLAGS
    with pytest.raises(ValueError):
        L.check(L.EGT_E_FLAGS)
    d = L.AttnDesc(B=1, N=4, H=8, d=8, dtype=7, flags=0)
    assert egt_lib.egt_attn_fwd(C.byref(d), *([None] * 12)) == L.EGT_E_DTYPE
    d = L.AttnDesc(B=1, N=4, H=8, d=8, dtype=L.EGT_F32, flags=0)
    assert egt_lib.egt_attn_fwd(C.byref(d), *([None] * 12)) == L.EGT_E_NULL
    e = L.EdgeDesc(rows=10, De=24, H=8, dtype=L.EGT_F32, flags=0, act=0, act_alpha=0, ln_eps=1e-3)
    assert egt_lib.egt_edge_update_fwd(C.byref(e), *([None] * 6)) == L.EGT_E_SHAPE


def test_pair_entry_points_validate_without_gpu(egt_lib):
    """egt_pair_* (the fused pair operator of the large-head geometry): coverage query, workspace size and argument errors -- no launch"""
    from egt_amd import _lib as L
    d = L.BlockDesc(B=8, N=512, H=8, d=64, De=32, dtype=L.EGT_F32, flags=L.BF_GATE | L.BF_CLIP, clip_lo=-5, clip_hi=5,
                    random_mask_prob=0.0, ln_eps=1e-3, reserved=0, seed=0, seed_device=None)
    assert egt_lib.egt_pair_supported(C.byref(d)) == 1
    arr = 8 * 8 * 512 * 64
    nwg = 8 * 32
    floats = 6 * arr + 8 * 8 * 512 * 4 + 8 * 8 * 512 * 512 + (nwg + 1) * (32 * 16 + 16) + (nwg + 1) * (8 * 32 + 32) + 256 + 7 * 64 * 4
    assert egt_lib.egt_pair_workspace_bytes(C.byref(d)) == 4 * floats
    prm = L.BlockParams()
    assert egt_lib.egt_pair_fwd(C.byref(d), C.byref(prm), *([None] * 8)) == L.EGT_E_NULL      # qkv / e / workspace NULL
    d.d = 8                                                                                  # the d <= 8 geometry belongs to egt_block_*
    assert egt_lib.egt_pair_supported(C.byref(d)) == 0 and egt_lib.egt_pair_workspace_bytes(C.byref(d)) == 0
    assert egt_lib.egt_pair_fwd(C.byref(d), C.byref(prm), *([None] * 8)) == L.EGT_E_SHAPE
    assert b"fused pair operator" in egt_lib.egt_last_error_string()
    assert egt_lib.egt_pair_bwd(C.byref(d), C.byref(prm), *([None] * 9), C.byref(prm), None, None) == L.EGT_E_SHAPE


def test_dp_entry_points_validate_without_a_communicator(egt_lib):
    """egt_dp_* (SURVEY 8(b)): state queries and argument errors before any communicator exists (no RCCL call)."""
    from egt_amd import _lib as L
    assert egt_lib.egt_dp_world() == 0 and egt_lib.egt_dp_rank() == -1
    assert egt_lib.egt_dp_allreduce(None, 16, 1, None) == L.EGT_E_FLAGS
    assert b"before egt_dp_init" in egt_lib.egt_last_error_string()
    assert egt_lib.egt_dp_init(None, 1, 0) == L.EGT_E_NULL
    ident = C.create_string_buffer(128)
    assert egt_lib.egt_dp_init(ident, 2, 2) == L.EGT_E_SHAPE and egt_lib.egt_dp_init(ident, 0, 0) == L.EGT_E_SHAPE
    assert egt_lib.egt_dp_unique_id(None) == L.EGT_E_NULL
    assert egt_lib.egt_dp_finalize() == L.EGT_OK          # nothing to destroy
    with pytest.raises(RuntimeError):
        L.check(L.EGT_E_RCCL)


def test_layer_constructor_errors_match_reference():
    from egt_amd import EGT, EGTBlock
    with pytest.raises(ValueError):           # egt_layers.py:20-21
        EGT(scale_degree=True, gate_input=False)
    with pytest.raises(ValueError):           # egt_layers.py:23-24
        EGT(scaler_type="sqrt")
    with pytest.raises(KeyError):             # graph_xformer_model_base.py:328-334
        EGTBlock(edge_channel_type="bogus")
    cfg = EGT(name="mha_00").get_config()
    assert "attn_dropout" not in cfg and cfg["num_heads"] == 8


def test_no_cpu_fallback():
    import torch
    from egt_amd import EGT
    layer = EGT()
    qkv = torch.zeros(1, 4, 3 * 8 * 8)
    x = torch.zeros(1, 4, 4, 8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        layer([qkv, x, x])


def test_keras_parameter_names():
    from egt_amd import EGTBlock
    blk = EGTBlock(model_width=64, edge_width=64)
    names = set(blk.keras_named_parameters("03"))
    want = {f"{n}_03/{p}" for n, ps in dict(
        norm_edge=("gamma", "beta"), attention_gates=("kernel", "bias"),
        dense_edge_b=("kernel", "bias"), norm_mha=("gamma", "beta"),
        dense_qkv=("kernel", "bias"), dense_mha=("kernel", "bias"),
        dense_edge_r=("kernel", "bias")).items() for p in ps}
    assert names == want


def test_ffn_and_mfma_argument_validation_without_gpu(egt_lib):
    """the newer entry points validate before they launch: no GPU needed"""
    from egt_amd import _lib as L
    f = L.FfnDesc(rows=100, width=40, dtype=L.EGT_F32, activation=L.ACT_ELU, ln_eps=1e-3)
    assert egt_lib.egt_ffn_supported(C.byref(f)) == 0 and egt_lib.egt_ffn_workspace_bytes(C.byref(f)) == 0
    p = L.FfnParams()
    assert egt_lib.egt_ffn_fwd(C.byref(f), C.byref(p), None, None, None, None) == L.EGT_E_NULL   # workspace NULL first
    f = L.FfnDesc(rows=100, width=64, dtype=L.EGT_F32, activation=L.ACT_LRELU, ln_eps=1e-3)
    assert egt_lib.egt_ffn_supported(C.byref(f)) == 0
    f = L.FfnDesc(rows=100, width=48, dtype=L.EGT_F32, activation=L.ACT_RELU, ln_eps=1e-3)
    assert egt_lib.egt_ffn_supported(C.byref(f)) == 1
    W = 48
    assert egt_lib.egt_ffn_workspace_bytes(C.byref(f)) >= 4 * (4 * 2 * W * W)
    a = L.AttnDesc(B=1, N=40, H=8, d=24, dtype=L.EGT_F32, flags=0)
    assert egt_lib.egt_attn_mfma_supported(C.byref(a), 0) == 0          # d not in {16,32,64}
    a = L.AttnDesc(B=1, N=40, H=8, d=32, dtype=L.EGT_F32, flags=0)
    assert egt_lib.egt_attn_mfma_supported(C.byref(a), 0) == 1 and egt_lib.egt_attn_mfma_supported(C.byref(a), 1) == 0
    assert egt_lib.egt_attn_mfma_workspace_bytes(C.byref(a)) > egt_lib.egt_attn_mfma_fwd_workspace_bytes(C.byref(a)) > 0
    b = L.BlockDesc(B=2, N=16, H=8, d=8, De=64, dtype=L.EGT_BF16, flags=0, clip_lo=0, clip_hi=0,
                    random_mask_prob=0, ln_eps=1e-3, reserved=0, seed=0)
    assert egt_lib.egt_block_supported(C.byref(b)) == 1                 # bf16 edge tensors are covered
    b.dtype = 5
    assert egt_lib.egt_block_supported(C.byref(b)) == 0


def test_header_compiles_as_c_and_struct_layouts_match_the_ctypes_mirrors(tmp_path):
    """include/egt_amd.h is the boundary: plain C (gcc -std=c99, no HIP / torch types), and every struct a caller fills has
    the size and field offsets of its ctypes mirror in egt_amd/_lib.py (a silent mismatch would shift every later field)."""
    import ctypes as C
    import subprocess
    from egt_amd import _lib as L
    pairs = [("egt_attn_desc", L.AttnDesc), ("egt_edge_desc", L.EdgeDesc), ("egt_block_desc", L.BlockDesc),
             ("egt_block_params", L.BlockParams), ("egt_ffn_desc", L.FfnDesc), ("egt_ffn_params", L.FfnParams),
             ("egt_embed_desc", L.EmbedDesc)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "egt_amd.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  printf("abi %d\\n", EGT_ABI_VERSION);', '  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = dict(ln.rsplit(" ", 1) for ln in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
    assert int(got["abi"]) == 4
