"""ResNet50 with the deep-feature and logits layers of the reference, executed by libosi_hip on an MI355X.

Drop-in for `openset_imagenet.model.ResNet50` (reference openset_imagenet/model.py:5-39):
same constructor `ResNet50(fc_layer_dim, out_features, logit_bias)`, `forward(image) -> (logits, features)`,
`model.logits.in_features / .out_features` (read at reference train.py:210-211) and the same 321 `state_dict()`
keys (`resnet_base.conv1.weight` ... `resnet_base.fc.bias`, `logits.weight`), so reference checkpoints load.

What is different underneath (nothing of torchvision / ATen / MIOpen runs):
  * all 162 parameter tensors are views into ONE flat fp32 arena (`_flat_params`), gradients into a second arena of the
    same layout (`_flat_grads`), BN running statistics into a third; conv weights keep the logical OIHW shape but are
    stored KRSC (= channels_last strides), which is what the implicit-GEMM kernels read directly;
  * `forward` is one C call (`osi_resnet50_forward`) that enqueues the whole network on the current HIP stream;
    `backward` is `osi_resnet50_backward`, run stage by stage so that a gradient all-reduce (dp.py) can start on the
    finished part of the arena while earlier layers are still being differentiated.
There is no CPU path: calling the module with a CPU tensor raises.
"""
import ctypes
import math
import os
import weakref

import torch
from torch import nn

from . import _native as N


class _Node(nn.Module):
    """Pure container mirroring one torchvision sub-module in the state_dict hierarchy (no forward of its own)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("sub-modules of the MI355X ResNet50 are parameter containers; call the model itself")


def debug_options_from_env():
    """Development A/B switches of the executor, read from the environment in ONE place and handed over explicitly through
    osi_resnet50_set_option (the library itself never reads the environment). Every default is the measured optimum; nothing here
    is needed to run the product. {option name: value} for the variables that are set:
        OSI_NO_OVERLAP=1      overlap 0           weight gradients stay on the main stream (serialised backward)
        OSI_FWD_RECOMPUTE=1   fwd_recompute 1     conv1 recomputes the previous identity-shortcut block output in its loader
        OSI_FWD_FORK=0        fwd_fork 0          projection shortcut of the forward pass on the main stream
        OSI_STAGGER=1         stagger 1           weight gradients only beside BatchNorm-backward kernels, never beside an input gradient
        OSI_SIDE_PRIO=n       side_priority_normal 1
        OSI_STEM_FUSED=0      stem_fused 0        conv1's weight gradient from a materialised 112x112x64 gradient (BatchNorm apply pass)
        OSI_STEM_POOL_STATS=0 stem_pool_stats 0   bn1's backward reductions by their own pass instead of layer1.0.conv1's dgrad epilogue
        OSI_DS_SPARSE=0       ds_sparse 0         stride-2 shortcut gradients written / read as dense tensors (zero fill included)
        OSI_STEM_WGRAD_MAIN=0 stem_wgrad_main 0   the fused stem weight gradient queued on the side stream behind layer1's weight gradients
        OSI_WINO_WEIGHTS_ASIDE=0 wino_weights_aside 0   the Winograd weight transforms of a training forward on the main stream (not beside the stem)
        OSI_EVAL_FUSED=0      eval_fused 0        eval-mode forwards through the training topology on running statistics (no inference forms)
    (There is no switch for round 2's fused in-block activations: the unfused executor path no longer exists; its price on one box is the
    three-way A/B of the committed round-1 / round-2 / current trees, profiles/r03_ab_rounds.txt.)"""
    env = os.environ
    out = {}
    if env.get("OSI_NO_OVERLAP"):
        out["overlap"] = 0
    if env.get("OSI_DBG_SKIP") and env.get("OSI_DEV") == "1" and N.DIAGNOSTIC_LIB:
        # timing experiments only (wrong results): the switch exists in the diagnostic build alone (`make -C csrc diag`, -DOSI_DIAG)
        out["dbg_skip"] = int(env["OSI_DBG_SKIP"])
    if env.get("OSI_FWD_RECOMPUTE") == "1":
        out["fwd_recompute"] = 1
    if env.get("OSI_FWD_FORK") == "0":
        out["fwd_fork"] = 0
    if env.get("OSI_STAGGER") == "1":
        out["stagger"] = 1
    if env.get("OSI_SIDE_PRIO", "")[:1] == "n":
        out["side_priority_normal"] = 1
    if env.get("OSI_STEM_FUSED") == "0":
        out["stem_fused"] = 0
    if env.get("OSI_STEM_POOL_STATS") == "0":
        out["stem_pool_stats"] = 0
    if env.get("OSI_DS_SPARSE") == "0":
        out["ds_sparse"] = 0
    if env.get("OSI_WINO_WEIGHTS_ASIDE") == "0":
        out["wino_weights_aside"] = 0
    if env.get("OSI_STEM_WGRAD_MAIN") == "0":
        out["stem_wgrad_main"] = 0
    if env.get("OSI_EVAL_FUSED") == "0":
        out["eval_fused"] = 0
    return out


class _Net:
    """Owner of one executor handle (fixed batch / image size)."""

    def __init__(self, B, H, W, F, O, logit_bias):
        self.h = ctypes.c_void_p()
        N.check(N.lib().osi_resnet50_create(ctypes.byref(self.h), B, H, W, F, O, int(bool(logit_bias))), "osi_resnet50_create")
        self.ws_bytes = N.lib().osi_resnet50_workspace_bytes(self.h)
        self.staged = False      # executor option "stage_join" = 0 has been set (data-parallel backward)
        for name, value in debug_options_from_env().items():
            N.check(N.lib().osi_resnet50_set_option(self.h, name.encode(), value), f"osi_resnet50_set_option({name})")

    def __del__(self):
        try:
            if self.h:
                N.lib().osi_resnet50_destroy(self.h)
                self.h = None
        except Exception:
            pass


class _BackboneFn(torch.autograd.Function):
    """Autograd node standing for the whole network: parameter gradients are written straight into the gradient arena."""

    @staticmethod
    def forward(ctx, image, anchor, model, flip):
        ctx.model = model
        ctx.set_materialize_grads(False)
        out = model._run_forward(image, True, flip)
        ctx.serial = model._fwd_serial
        return out

    @staticmethod
    def backward(ctx, dlogits, dfeatures):
        if ctx.serial != ctx.model._fwd_serial:
            raise RuntimeError("backward() of a forward pass that is no longer the model's latest one: the executor keeps the "
                               "activations of ONE forward (the reference loop is forward, loss, backward, step — train.py:132-139)")
        ctx.model._run_backward(dlogits, dfeatures)
        return None, None, None, None


class ResNet50(nn.Module):
    """Represents a ResNet50 model (reference model.py:5)."""

    def __init__(self, fc_layer_dim=1000, out_features=1000, logit_bias=True):
        super().__init__()
        self._F, self._O, self._logit_bias = int(fc_layer_dim), int(out_features), bool(logit_bias)
        lib = N.lib()
        probe = _Net(1, 32, 32, self._F, self._O, self._logit_bias)  # layout does not depend on batch / image size
        self._n_stages = lib.osi_resnet50_num_stages(probe.h)
        nparam = lib.osi_resnet50_param_floats(probe.h)
        nbuf = lib.osi_resnet50_buffer_floats(probe.h)
        nbn = lib.osi_resnet50_num_bn(probe.h)
        object.__setattr__(self, "_flat_params", torch.zeros(nparam))
        object.__setattr__(self, "_flat_grads", torch.zeros(nparam))
        object.__setattr__(self, "_flat_buffers", torch.zeros(nbuf))
        object.__setattr__(self, "_nbt", torch.zeros(nbn, dtype=torch.int64))
        object.__setattr__(self, "_anchor", torch.zeros(1, requires_grad=True))
        self._stage_ranges = []
        lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
        for s in range(self._n_stages):
            N.check(lib.osi_resnet50_stage_grad_range(probe.h, s, ctypes.byref(lo), ctypes.byref(hi)))
            self._stage_ranges.append((lo.value, hi.value))

        # ---- build the module tree + parameter views -------------------------------------------------------
        self._pinfo = []   # (name, offset, numel, shape)
        self._binfo = []   # (node, buffer name, arena name, offset, numel)
        name = ctypes.create_string_buffer(160)
        nd, shp, off, ne = ctypes.c_int(), (ctypes.c_int * 4)(), ctypes.c_size_t(), ctypes.c_size_t()
        bn_prefix = {}
        C, rm, rv = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_size_t()
        for j in range(nbn):
            N.check(lib.osi_resnet50_bn_info(probe.h, j, name, 160, ctypes.byref(C), ctypes.byref(rm), ctypes.byref(rv)))
            bn_prefix[name.value.decode()] = (j, C.value, rm.value, rv.value)
        for i in range(lib.osi_resnet50_num_tensors(probe.h)):
            N.check(lib.osi_resnet50_tensor_info(probe.h, i, name, 160, ctypes.byref(nd), shp, ctypes.byref(off), ctypes.byref(ne)))
            full = name.value.decode()
            shape = tuple(shp[k] for k in range(nd.value))
            self._pinfo.append((full, off.value, ne.value, shape))
            *path, leaf = full.split(".")
            node = self._node(path)
            node.register_parameter(leaf, nn.Parameter(self._view(self._flat_params, off.value, ne.value, shape)))
            prefix = ".".join(path)
            if leaf == "bias" and prefix in bn_prefix:
                j, c, rmo, rvo = bn_prefix[prefix]
                node.register_buffer("running_mean", self._flat_buffers[rmo:rmo + c])
                node.register_buffer("running_var", self._flat_buffers[rvo:rvo + c])
                node.register_buffer("num_batches_tracked", self._nbt[j])
                self._binfo += [(node, "running_mean", "_flat_buffers", rmo, c), (node, "running_var", "_flat_buffers", rvo, c),
                                (node, "num_batches_tracked", "_nbt", j, 0)]
        self.logits.in_features, self.logits.out_features = self._F, self._O
        self.resnet_base.fc.in_features, self.resnet_base.fc.out_features = 2048, self._F
        self._plist = [dict(self.named_parameters())[n] for (n, _, _, _) in self._pinfo]
        for p in self._plist:
            p._osi_owner = weakref.ref(self)
        self._nets = {}
        self._ws = None
        self._grad_sync = None   # set by dp.DistributedDataParallel
        self._fwd_serial = 0     # number of forward passes run; a backward must belong to the latest one
        self._grads_fresh = False  # a backward has filled the gradient arena since the last optimizer.zero_grad()
        self.reset_parameters()

    # ------------------------------------------------------------------------------------------------------
    def _node(self, path):
        node = self
        for comp in path:
            if comp not in node._modules:
                node.add_module(comp, _Node())
            node = node._modules[comp]
        return node

    @staticmethod
    def _view(arena, off, numel, shape):
        flat = arena[off:off + numel]
        if len(shape) == 4:  # logical OIHW over physical [O][H][W][I]
            o, i, h, w = shape
            return flat.view(o, h, w, i).permute(0, 3, 1, 2)
        return flat.view(shape)

    def reset_parameters(self):
        """torchvision's initialisation: kaiming-normal(fan_out, relu) convs, BN weight 1 / bias 0, default nn.Linear."""
        with torch.no_grad():
            for (name, _, _, shape), p in zip(self._pinfo, self._plist):
                if len(shape) == 4:
                    nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")
                elif len(shape) == 2:
                    nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                elif name.endswith("fc.bias") or name == "logits.bias":
                    fan_in = 2048 if name.endswith("fc.bias") else self._F
                    bound = 1 / math.sqrt(fan_in)
                    nn.init.uniform_(p, -bound, bound)
                elif name.endswith(".weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()
            for node, bname, _, _, _ in self._binfo:
                buf = node._buffers[bname]
                buf.fill_(1.0) if bname == "running_var" else buf.zero_()

    # ---- device / dtype movement keeps the arenas whole ---------------------------------------------------
    def _apply(self, fn, recurse=True):
        new = {}
        for arena in ("_flat_params", "_flat_grads", "_flat_buffers"):
            t = fn(getattr(self, arena))
            if t.dtype != torch.float32:
                raise RuntimeError("the MI355X ResNet50 is fp32 only (parity dtype of the reference path)")
            new[arena] = t.contiguous()
        new["_nbt"] = fn(self._nbt).to(torch.int64)
        anchor = fn(self._anchor.detach()).requires_grad_(True)
        for k, v in new.items():
            object.__setattr__(self, k, v)
        object.__setattr__(self, "_anchor", anchor)
        with torch.no_grad():
            for (name, off, numel, shape), p in zip(self._pinfo, self._plist):
                had_grad = p.grad is not None
                p.data = self._view(self._flat_params, off, numel, shape)
                if had_grad:
                    p.grad = self._view(self._flat_grads, off, numel, shape)
            for node, bname, arena, off, c in self._binfo:
                src = getattr(self, arena)
                node._buffers[bname] = src[off] if arena == "_nbt" else src[off:off + c]
        self._ws = None
        return self

    # ---- arena access for the optimizer / DP layers -------------------------------------------------------
    def flat_parameters(self):
        return self._flat_params

    def flat_gradients(self):
        return self._flat_grads

    def gradient_buckets(self):
        """[(lo, hi)] float ranges of the gradient arena in the order backward finishes them (head first)."""
        return list(self._stage_ranges)

    def bind_gradients(self):
        """Point every parameter's .grad at its slice of the gradient arena."""
        for (name, off, numel, shape), p in zip(self._pinfo, self._plist):
            if p.grad is None or p.grad.data_ptr() != self._flat_grads.data_ptr() + 4 * off:
                p.grad = self._view(self._flat_grads, off, numel, shape)

    # ---- execution -------------------------------------------------------------------------------------------
    def _net(self, B, H, W):
        key = (B, H, W)
        if key not in self._nets:
            self._nets[key] = _Net(B, H, W, self._F, self._O, self._logit_bias)
        net = self._nets[key]
        dev = self._flat_params.device
        if self._ws is None or self._ws.numel() < net.ws_bytes or self._ws.device != dev:
            self._ws = None
            self._ws = torch.empty(net.ws_bytes, dtype=torch.uint8, device=dev)
        return net

    def mark_gradients_ready(self):
        """Tell the fused optimizers that the gradient arena was filled by hand (tests, custom loops) rather than by backward()."""
        self._grads_fresh = True

    @staticmethod
    def _is_nhwc4(image):
        """fp32 [B, H, W, 4] batch already in the executor's input layout (written by pipeline.DevicePrefetcher)."""
        return image.dtype == torch.float32 and image.dim() == 4 and image.shape[3] == 4 and image.shape[1] != 3

    def _check_image(self, image):
        if isinstance(image, torch.Tensor) and image.dtype == torch.uint8:   # decoded RGB batch, staged on the device
            if image.dim() != 4 or image.shape[3] != 3:
                raise ValueError("a uint8 image batch must be [B, H, W, 3] (decoded RGB rows)")
        elif isinstance(image, torch.Tensor) and self._is_nhwc4(image):
            pass
        elif not isinstance(image, torch.Tensor) or image.dim() != 4 or image.shape[1] != 3:
            raise ValueError("expected an image batch [B, 3, H, W]")
        if not image.is_cuda or not self._flat_params.is_cuda:
            raise RuntimeError("openset_imagenet (MI355X build) has no CPU path: move the model and the batch to the GPU "
                               "(set_device_gpu(index); device(model); device(images)).")
        if image.device != self._flat_params.device:
            raise RuntimeError("image batch and model live on different devices")
        if image.dtype not in (torch.float32, torch.uint8):
            raise TypeError("image batch must be float32 [B,3,H,W] (reference: ToTensor(), train.py:259-263) or uint8 [B,H,W,3]")

    def _run_forward(self, image, want_grad, flip=None):
        image = image.contiguous()
        staged = image.dtype == torch.uint8
        bound = not staged and self._is_nhwc4(image)
        if staged or bound:
            B, H, W, _ = image.shape
        else:
            B, _, H, W = image.shape
        net = self._net(B, H, W)
        if staged and flip is not None:   # ToTensor + horizontal flip + NHWC4 staging in one pass on the device (osi_u8hwc3_to_nhwc4)
            flip = torch.as_tensor(flip).to(device=image.device, dtype=torch.uint8).contiguous()
            if flip.numel() != B:
                raise ValueError("flip must hold one flag per image")
        elif flip is not None:
            raise ValueError("flip flags are only meaningful with a uint8 [B,H,W,3] batch")
        # one custom op = the whole network on the current HIP stream: uint8 batches are staged, NHWC4 batches bound in place
        # (conv1 forward now, conv1 weight gradient at the end of this step's backward), NCHW batches converted on the way in
        logits, features = N.ops().resnet50_forward(net.h.value, self._flat_params, self._flat_buffers, self._nbt, image, flip,
                                                    self._ws, self._F, self._O, bool(self.training))
        self._fwd_serial += 1
        self._last = (net, image if (want_grad or bound) else None)   # keeps a bound batch alive until the next forward
        return logits, features

    def _run_backward(self, dlogits, dfeatures):
        net, _ = self._last
        if dlogits is None:
            dlogits = torch.zeros(net_shape(self, net)[0], self._O, device=self._flat_params.device)
        dlogits = dlogits.contiguous().float()
        dfeatures = None if dfeatures is None else dfeatures.contiguous().float()
        sync = self._grad_sync
        bwd = N.ops().resnet50_backward
        if sync is None:  # single GPU: all stages in one call (one side-stream join at the end)
            bwd(net.h.value, self._flat_params, self._flat_grads, self._ws, dlogits, dfeatures, 0, self._n_stages)
        else:             # data parallel: stage by stage, each finished slice of the gradient arena goes to the all-reduce
            if not net.staged:   # staged calls no longer join the weight-gradient side stream into the compute stream (only the last does)
                N.check(N.lib().osi_resnet50_set_option(net.h, b"stage_join", 0), "osi_resnet50_set_option(stage_join)")
                net.staged = True
            grads = self._flat_grads
            handoff = lambda comm: N.ops().resnet50_grads_ready(net.h.value, grads, comm.cuda_stream)
            for s in range(self._n_stages):
                bwd(net.h.value, self._flat_params, grads, self._ws, dlogits, dfeatures, s, s + 1)
                lo, hi = self._stage_ranges[s]
                sync.bucket_ready(grads, lo, hi, handoff)
            sync.finish()
        self._grads_fresh = True
        self.bind_gradients()

    def forward(self, image, flip=None):
        """Forward pass: returns (logits, deep features) like the reference (model.py:28-39).

        `image` is the reference's fp32 [B,3,H,W] batch in [0,1], or — the device-side input pipeline — a uint8 [B,H,W,3] batch
        of decoded, cropped RGB rows with optional per-image horizontal-flip flags (ToTensor(), RandomHorizontalFlip and the
        layout staging then happen in one pass on the GPU and the host link carries a quarter of the bytes), or an fp32
        [B,H,W,4] batch already staged in the executor's layout by pipeline.DevicePrefetcher (read in place)."""
        self._check_image(image)
        if torch.is_grad_enabled() and self.training and self._plist[0].requires_grad:
            return _BackboneFn.apply(image, self._anchor, self, flip)
        return self._run_forward(image, False, flip)


def net_shape(model, net):
    for (B, H, W), n in model._nets.items():
        if n is net:
            return B, H, W
    raise RuntimeError("unknown executor")
