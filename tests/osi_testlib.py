"""Helpers for the GPU parity tests: every call goes through the C ABI of libosi_hip.so (ctypes), torch is only the allocator."""
import ctypes

import torch

from openset_imagenet import _native as N


def S():
    return torch.cuda.current_stream().cuda_stream


def nhwc(x):          # NCHW tensor -> contiguous NHWC copy
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):          # NHWC tensor -> NCHW view
    return x.permute(0, 3, 1, 2)


def krsc(w):          # OIHW -> [O][R][S][I]
    return w.permute(0, 2, 3, 1).contiguous()


def oihw(w):
    return w.permute(0, 3, 1, 2)


def conv_fwd(x_nhwc, w_krsc, k, stride, pad, tile=0):
    B, H, W, Cin = x_nhwc.shape
    Cout = w_krsc.shape[0]
    d = N.ConvDesc.make(B, H, W, Cin, Cout, k, stride, pad)
    y = torch.empty(B, d.Ho, d.Wo, Cout, device=x_nhwc.device)
    N.check(N.lib().osi_conv_fwd(ctypes.byref(d), N.ptr(x_nhwc), N.ptr(w_krsc), N.ptr(y), tile, S()), "conv_fwd")
    return y


def conv_dgrad(dy_nhwc, w_krsc, H, W, k, stride, pad, accumulate_into=None, tile=0):
    B, Ho, Wo, Cout = dy_nhwc.shape
    Cin = w_krsc.shape[3]
    d = N.ConvDesc.make(B, H, W, Cin, Cout, k, stride, pad)
    assert (d.Ho, d.Wo) == (Ho, Wo)
    dx = accumulate_into if accumulate_into is not None else torch.full((B, H, W, Cin), float("nan"), device=dy_nhwc.device)
    N.check(N.lib().osi_conv_dgrad(ctypes.byref(d), N.ptr(dy_nhwc), N.ptr(w_krsc), N.ptr(dx), int(accumulate_into is not None), tile, S()), "conv_dgrad")
    return dx


def conv_wgrad(dy_nhwc, x_nhwc, k, stride, pad):
    B, H, W, Cin = x_nhwc.shape
    Cout = dy_nhwc.shape[3]
    d = N.ConvDesc.make(B, H, W, Cin, Cout, k, stride, pad)
    stem = Cin == 4 and k == 7
    ktot = 224 if stem else k * k * Cin
    dw = torch.full((Cout, ktot), float("nan"), device=x_nhwc.device)
    nbytes = N.lib().osi_conv_wgrad_workspace(ctypes.byref(d))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x_nhwc.device)
    N.check(N.lib().osi_conv_wgrad(ctypes.byref(d), N.ptr(dy_nhwc), N.ptr(x_nhwc), N.ptr(dw), N.ptr(ws), nbytes, S()), "conv_wgrad")
    return dw if stem else dw.view(Cout, k, k, Cin)


def hip_gates(model):
    """The ReLU / arg-max decisions of the model's latest forward, in the structure oracle.resnet50_oracle.forward(gates=...)
    takes: {"relu": [49 bool tensors, NCHW], "pool_idx": int64 [B,64,Hp,Wp]} (CPU tensors). Read through the debug entry points of
    the C ABI (include/osi.h, "debug" section), i.e. from the very buffers the backward kernels consume."""
    net, _ = model._last
    lib, dev = N.lib(), model._flat_params.device
    B = next(b for (b, h, w), n in model._nets.items() if n is net)
    relu, pool_idx = [], None
    C, H, W = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    for i in range(lib.osi_resnet50_debug_num_gates(net.h)):
        N.check(lib.osi_resnet50_debug_gate_shape(net.h, i, ctypes.byref(C), ctypes.byref(H), ctypes.byref(W)))
        g = torch.empty(B, C.value, H.value, W.value, dtype=torch.uint8, device=dev)
        am = torch.empty(B, C.value, H.value, W.value, dtype=torch.int32, device=dev) if i == 0 else None
        N.check(lib.osi_resnet50_debug_gate(net.h, N.ptr(model._ws), i, N.ptr(g), N.ptr(am), S()), "debug_gate")
        relu.append(g.cpu().bool())
        if am is not None:
            pool_idx = am.cpu().long()
    return {"relu": relu, "pool_idx": pool_idx}


class Fusion(ctypes.Structure):
    """osi_dgrad_fusion of include/osi.h (ABI 4)."""
    _fields_ = [("relu_mask", ctypes.c_void_p), ("y0", ctypes.c_void_p), ("mean0", ctypes.c_void_p), ("invstd0", ctypes.c_void_p),
                ("y1", ctypes.c_void_p), ("mean1", ctypes.c_void_p), ("invstd1", ctypes.c_void_p), ("partials", ctypes.c_void_p),
                ("partials_bytes", ctypes.c_size_t), ("scale0", ctypes.c_void_p), ("shift0", ctypes.c_void_p),
                ("pool_idx", ctypes.c_void_p), ("pool_H", ctypes.c_int), ("pool_W", ctypes.c_int), ("addend_stride", ctypes.c_int)]
