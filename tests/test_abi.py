"""CPU: the C-ABI shared library loads without a GPU and exports every symbol include/osi.h declares; the ctypes table in
openset_imagenet/_native.py lists exactly those symbols; host-only executor queries agree with the oracle's layout."""
import ctypes
import os
import re

import pytest

from openset_imagenet import _native as N
from oracle import resnet50_oracle as R

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "osi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(osi_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = N.lib()
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f"{s} is declared in include/osi.h but not exported by libosi_hip.so"
    assert sorted(N.declared_symbols()) == syms, "ctypes signature table and header drifted apart"
    assert lib.osi_build_arch() == b"gfx950" and lib.osi_abi_version() >= 1
    assert lib.osi_strerror(-1).startswith(b"invalid")


def test_argument_validation_without_gpu():
    """Precondition failures return OSI_ERR_ARG before anything is launched (safe to call on a CPU-only host)."""
    lib = N.lib()
    d = N.ConvDesc.make(2, 8, 8, 48, 64, 3, 1, 1)   # Cin not a multiple of 32
    assert lib.osi_conv_fwd(ctypes.byref(d), 16, 16, 16, 0, None) == -1
    bad = N.ConvDesc(2, 8, 8, 64, 9, 9, 64, 3, 3, 1, 1)  # inconsistent output size
    assert lib.osi_conv_fwd(ctypes.byref(bad), 16, 16, 16, 0, None) == -1
    assert lib.osi_conv_fwd(ctypes.byref(d), None, None, None, 0, None) == -1
    # weight-gradient shapes: input channels in 64s (32s only for the all-taps 3x3 stride-1 form); unsupported -> 0 bytes / OSI_ERR_ARG
    lib.osi_conv_wgrad_workspace.restype = ctypes.c_size_t
    assert lib.osi_conv_wgrad_workspace(ctypes.byref(N.ConvDesc.make(2, 8, 8, 32, 64, 1, 1, 0))) == 0
    assert lib.osi_conv_wgrad(ctypes.byref(N.ConvDesc.make(2, 8, 8, 32, 64, 1, 1, 0)), 16, 16, 16, 16, 0, None) == -1
    assert lib.osi_conv_wgrad_workspace(ctypes.byref(N.ConvDesc.make(64, 14, 14, 32, 64, 3, 1, 1))) > 0
    assert lib.osi_bn_apply(None, None, None, None, None, 4, 64, 1, None) == -1
    assert lib.osi_loss_fwd_bwd(7, 16, 16, 4, 4, 1.0, -1, None, None, 0, 0.0, 0.0, 16, 16, None, None) == -1
    assert lib.osi_adam_step(16, 16, 16, 16, 6, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None) == -1   # n % 4 != 0
    h = ctypes.c_void_p()
    assert lib.osi_resnet50_create(ctypes.byref(h), 0, 224, 224, 10, 10, 0) == -1
    with pytest.raises(RuntimeError):
        N.check(-1, "demo")


def test_executor_layout_matches_reference_state_dict():
    lib = N.lib()
    h = ctypes.c_void_p()
    assert lib.osi_resnet50_create(ctypes.byref(h), 128, 224, 224, 116, 116, 0) == 0
    try:
        assert lib.osi_resnet50_num_tensors(h) == 162 and lib.osi_resnet50_num_bn(h) == 53
        assert lib.osi_resnet50_param_floats(h) == 23759172
        name = ctypes.create_string_buffer(160)
        nd, shp, off, ne = ctypes.c_int(), (ctypes.c_int * 4)(), ctypes.c_size_t(), ctypes.c_size_t()
        names, end = [], 0
        sd = R.init_state(116, 116)
        for i in range(162):
            assert lib.osi_resnet50_tensor_info(h, i, name, 160, ctypes.byref(nd), shp, ctypes.byref(off), ctypes.byref(ne)) == 0
            k = name.value.decode()
            names.append(k)
            assert tuple(shp[j] for j in range(nd.value)) == tuple(sd[k].shape), k
            assert off.value % 4 == 0 and off.value >= end
            end = off.value + ne.value
        assert names == R.param_keys(sd)
        # backward stages tile the gradient arena: head+layer4, layer3, layer2, layer1+stem
        lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
        ranges = []
        for s in range(lib.osi_resnet50_num_stages(h)):
            assert lib.osi_resnet50_stage_grad_range(h, s, ctypes.byref(lo), ctypes.byref(hi)) == 0
            ranges.append((lo.value, hi.value))
        assert ranges[0][1] == 23759172 and ranges[-1][0] == 0
        assert all(ranges[i][0] == ranges[i + 1][1] for i in range(len(ranges) - 1))
        assert lib.osi_resnet50_workspace_bytes(h) > 10 * 2 ** 30
        # out-of-order use is refused, not launched
        assert lib.osi_resnet50_backward(h, 16, 16, 16, 16, None, 0, 1, None) == -3
    finally:
        lib.osi_resnet50_destroy(h)


def test_backward_winograd_plans_leave_the_channel_count_of_cus_free():
    """Host logic only (no launch): under data parallelism the backward-pass Winograd kernels — one 512-register workgroup per CU —
    plan for hw_cus - 8 * dp_reserved_cus CUs (a CU holding one resident RCCL channel workgroup cannot host them at all), capped at a
    quarter of the chip; the weight gradient's split count, hence its workspace, shows the plan. Without a device the library assumes
    the MI355X's 256 CUs."""
    from openset_imagenet import _native as N
    L = N.lib()
    per_split = 9 * 4096 * 4                                   # one [9][64][64] fp32 partial
    d1 = N.ConvDesc.make(128, 56, 56, 64, 64, 3, 1, 1)         # 1 block  -> splits = CUs
    d16 = N.ConvDesc.make(128, 14, 14, 256, 256, 3, 1, 1)      # 16 blocks -> splits = CUs / 16
    prev = ctypes.c_int()
    N.check(L.osi_get_tuning(b"dp_reserved_cus", ctypes.byref(prev)))
    try:
        seen = {}
        for reserved in (0, 4, 8, 32):
            N.check(L.osi_set_tuning(b"dp_reserved_cus", reserved))
            seen[reserved] = (L.osi_conv_wgrad_wino_workspace(ctypes.byref(d1)) // per_split,
                              L.osi_conv_wgrad_wino_workspace(ctypes.byref(d16)) // per_split)
    finally:
        N.check(L.osi_set_tuning(b"dp_reserved_cus", prev.value))
    import torch
    if not torch.cuda.is_available():                          # 256 CUs assumed
        assert seen == {0: (256, 256), 4: (224, 224), 8: (192, 192), 32: (192, 192)}, seen
    assert seen[0][0] > seen[4][0] > seen[8][0] == seen[32][0] and seen[0][0] - seen[4][0] == 32
