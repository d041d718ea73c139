"""-m "not gpu": the C-ABI library loads, exports every symbol include/brats_hip.h declares, and the
host-side logic (argument validation, size queries, state-dict contract) works without a GPU."""
import argparse
import ctypes

import pytest
import torch

from brats21_amd import _lib


def test_library_exports_every_declared_symbol():
    l = ctypes.CDLL(_lib.LIB_PATH)
    names = _lib.declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(l, n), f"{n} declared in include/brats_hip.h but not exported"
    # one source of truth: the header's define, returned by the library, parsed by the binding
    assert _lib.lib().brats_abi_version() == _lib._header_abi_version() >= 4


def test_host_side_queries_and_argument_errors():
    l = _lib.lib()
    # chunk selection: width-48 layers use 48-channel chunks in bf16, 16 in f32; first layer 8 / 4
    assert l.brats_conv3d_chunk(_lib.BF16, 3, 1, 48, 0, 0) == 48
    assert l.brats_conv3d_chunk(_lib.BF16, 3, 1, 96, 0, 96) == 48
    assert l.brats_conv3d_chunk(_lib.BF16, 3, 1, 8, 0, 48) == 8
    assert l.brats_conv3d_chunk(_lib.BF16, 3, 1, 64, 0, 64) == 32
    assert l.brats_conv3d_chunk(_lib.F32, 3, 1, 48, 0, 48) == 16
    assert l.brats_conv3d_chunk(_lib.F32, 3, 1, 4, 0, 48) == 4
    assert l.brats_conv3d_chunk(_lib.BF16, 3, 1, 12, 0, 0) == 0
    # layers with 48 (mod 96) output channels: 24-channel chunks for the 4x8x16-tile kernel, unless switched off
    assert l.brats_conv3d_chunk(_lib.BF16, 3, 1, 48, 48, 48) == 24
    assert l.brats_conv3d_chunk(_lib.BF16, 3, 2, 48, 0, 48) == 48
    old = l.brats_conv3d_set_vs8(0)
    assert l.brats_conv3d_chunk(_lib.BF16, 3, 1, 48, 48, 48) == 48
    l.brats_conv3d_set_vs8(old)
    assert l.brats_conv3d_tiles_per_sample(128, 128, 128) == 32 * 16 * 16
    assert l.brats_conv3d_tiles_per_sample(4, 4, 4) == 1
    # packed size: chunks * macro-steps * cout16 * 64 lanes * 16 B
    assert l.brats_conv3d_packed_bytes(_lib.BF16, 3, 48, 48, 48) == 1 * 41 * 3 * 64 * 16
    assert l.brats_conv3d_packed_bytes(_lib.F32, 3, 16, 16, 16) == 1 * 27 * 1 * 64 * 16
    # NULL pointers are rejected with an error string, never dereferenced
    rc = l.brats_conv3d_fwd(None, 8, 8, None, 0, 0, None, None, None, 8, None, 0, 0, None, _lib.BF16, 3, 1, 1, 8, 8, 8, 8, None)
    assert rc == -1 and b"conv3d_fwd" in l.brats_last_error()
    rc = l.brats_maxpool2_fwd(None, 8, None, 8, None, _lib.BF16, 1, 8, 8, 8, 8, 0, None)
    assert rc == -1
    # the round-3 fold entry points: NULL tensors are argument errors with their own message, unsupported shapes say so
    P = ctypes.c_void_p
    one = (ctypes.c_float * 16)()
    ptr = ctypes.cast(one, P)
    rc = l.brats_gn_act_bwd_head(None, None, 3, None, 8, None, None, None, None, 8, None, None, None, None, None, None, _lib.BF16,
                                 1, 0.01, 1, 64, 8, 8, None, None)
    assert rc == -1 and b"gn_act_bwd_head" in l.brats_last_error()
    rc = l.brats_gn_act_bwd_head(ptr, ptr, 4, ptr, 8, ptr, ptr, ptr, ptr, 8, ptr, ptr, ptr, ptr, ptr, ptr, _lib.BF16, 1, 0.01, 1, 64, 8,
                                 8, None, None)
    assert rc == -2 and b"K = 3" in l.brats_last_error()  # (four logit planes: the two-call path is the one to use)
    rc = l.brats_gn_head_fwd(None, 8, None, 1, 0.01, None, None, None, _lib.BF16, 1, 8, 3, 64, None)
    assert rc == -1 and b"gn_head_fwd" in l.brats_last_error()
    rc = l.brats_gn_head_fwd(ptr, 8, ptr, 4, 0.01, ptr, None, ptr, _lib.BF16, 1, 8, 3, 64, None)
    assert rc == -2 and b"relu" in l.brats_last_error()  # (swish: not an activation this fold is built for)
    rc = l.brats_gn_act_bwd_pool(None, 8, None, 8, None, None, 8, None, None, None, None, 8, None, None, None, _lib.BF16, 1, 0.01, 1, 8,
                                 8, 8, 8, 8, None, None)
    assert rc == -1 and b"gn_act_bwd_pool" in l.brats_last_error()
    rc = l.brats_affine_act_pool_fwd(None, 8, None, None, 8, None, 8, None, _lib.BF16, 1, 0.01, None, 1, 8, 8, 8, 8, 0, None, None)
    assert rc == -1 and b"affine_act_pool_fwd" in l.brats_last_error()
    rc = l.brats_maxpool2_bwd_idx(None, None, 8, None, 8, None, 8, _lib.BF16, 1, 8, 8, 8, 8, 0, None)
    assert rc == -1 and b"maxpool2_bwd_idx" in l.brats_last_error()
    rc = l.brats_evonorm_se_fwd(None, 8, None, None, None, None, None, None, None, None, 8, None, None, None, None, 4, _lib.BF16, 1,
                                64, 8, 8, None, None)
    assert rc == -1 and b"evonorm_se_fwd" in l.brats_last_error()
    rc = l.brats_se_fwd(None, 1.0, None, None, None, None, None, None, 1, 8, 4, None)
    assert rc == -1 and b"se_fwd" in l.brats_last_error()
    assert l.brats_gn_bwd_head_ws_floats(2, 48, 3) == 2 * 2048 * (3 * 48 + 3)


def test_state_dict_contract_and_factory_errors():
    from brats21_amd import get_model
    from oracle import unet
    ns = dict(width=8, norm="group", act="relu", num_classes=3, dropout=0)
    m = get_model(argparse.Namespace(model="equiunet", **ns))
    shapes = unet.equiunet_state_shapes(8)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
    with pytest.raises(NameError):
        get_model(argparse.Namespace(model="nnunet", **ns))
    with pytest.raises(NotImplementedError):
        get_model(argparse.Namespace(model="equiunet", **{**ns, "norm": "layer"}))
    # --norm bcn (round 5): BCNorm + EstBN parameter / buffer names and order (networks/factory.py:125-160)
    mc = get_model(argparse.Namespace(model="equiunet", **{**ns, "norm": "bcn"}))
    shapes_c = unet.equiunet_state_shapes(8, norm="bcn")
    assert list(mc.state_dict().keys()) == list(shapes_c.keys())
    assert all(tuple(mc.state_dict()[k].shape) == tuple(v) for k, v in shapes_c.items())
    # --dropout (round 5): constructs for p in [0, 1), no extra state-dict entry, bad values raise like nn.Dropout
    md = get_model(argparse.Namespace(model="equiunet", **{**ns, "dropout": 0.1}))
    assert md.dropout_p == 0.1 and list(md.state_dict().keys()) == list(shapes.keys())
    with pytest.raises(ValueError):
        get_model(argparse.Namespace(model="equiunet", **{**ns, "dropout": 1.0}))
    # --norm batch (round 4): nn.BatchNorm3d's parameter / buffer names and order
    mb = get_model(argparse.Namespace(model="equiunet", **{**ns, "norm": "batch"}))
    shapes_b = unet.equiunet_state_shapes(8, norm="batch")
    assert list(mb.state_dict().keys()) == list(shapes_b.keys())
    assert all(tuple(mb.state_dict()[k].shape) == tuple(v) for k, v in shapes_b.items())
    # no CPU fallback: a CPU forward must fail loudly
    from brats21_amd import BratsHipError
    with pytest.raises(BratsHipError):
        m(torch.zeros(1, 4, 16, 16, 16))
    import copy
    copy.deepcopy(m)  # AveragedModel / SWA path of the reference (src/main_train.py:113)
