"""Fused optimizers over the flat parameter arena: one HIP launch per step instead of 162 x several.

Stand in for `torch.optim.Adam(params, lr)` / `torch.optim.SGD(params, lr, momentum=0.9)` built at reference
openset_imagenet/train.py:356-359 and stepped at train.py:139. They subclass torch.optim.Optimizer, keep per-parameter
state entries (`step`, `exp_avg`, `exp_avg_sq` / `momentum_buffer`) as VIEWS into flat state arenas, so `state_dict()` /
`load_state_dict()` round-trip with the stock torch optimizers and with the reference checkpoint dict
(`opt_state_dict`, train.py:54-60). `lr` is read from `param_groups[0]` each step, so `lr_scheduler.StepLR` works.
"""
import torch

from . import _native as N


def _arena_of(params):
    """The model that owns `params` in one flat arena (all parameters must come from one MI355X ResNet50)."""
    owner = None
    for p in params:
        o = getattr(p, "_osi_owner", None)
        o = o() if o is not None else None
        if o is None or (owner is not None and o is not owner):
            return None
        owner = o
    return owner


class _FlatOptimizer(torch.optim.Optimizer):
    _state_names = ()
    _has_step = True    # per-parameter "step" entry in the state dict (torch.optim.Adam has one, torch.optim.SGD does not)

    def __init__(self, model_or_params, defaults):
        if isinstance(model_or_params, torch.nn.Module):
            model = model_or_params
        else:  # reference spelling: Adam(params=model.parameters(), lr=...)
            given = list(model_or_params)
            model = _arena_of(given)
            if model is None or len(given) != len(model._plist) or any(a is not b for a, b in zip(given, model._plist)):
                raise ValueError("the fused optimizers step the whole flat arena: pass the MI355X ResNet50 (or exactly its "
                                 "model.parameters()); use torch.optim.* for anything else")
        params = list(model.parameters())
        super().__init__(params, defaults)
        self._model = model
        self._flat_state = {}
        self._steps = 0
        self._loaded_state = False   # load_state_dict brought moment / momentum buffers along

    def _ensure_state(self):
        m = self._model
        flat = m.flat_parameters()
        ok = all(k in self._flat_state and self._flat_state[k].device == flat.device for k in self._state_names)
        if ok:
            return
        old = dict(self._flat_state)
        for k in self._state_names:
            t = torch.zeros_like(flat)
            if k in old:
                t.copy_(old[k])
            self._flat_state[k] = t
        self._bind_views()

    def _bind_views(self):
        m = self._model
        for (name, off, numel, shape), p in zip(m._pinfo, m._plist):
            st = self.state[p]
            if self._has_step:
                st["step"] = torch.tensor(float(self._steps))
            for k in self._state_names:
                st[k] = m._view(self._flat_state[k], off, numel, shape)

    def zero_grad(self, set_to_none=True):
        # the executor overwrites the whole gradient arena each backward; dropping the references is enough
        super().zero_grad(set_to_none=set_to_none)
        self._model._grads_fresh = False

    def _skip_step(self):
        """torch skips parameters whose .grad is None; here the arena is all-or-nothing: without a backward since the last
        zero_grad() there is nothing to apply (stepping would re-apply stale gradients)."""
        m = self._model
        if any(not p.requires_grad for p in m._plist):
            raise RuntimeError("the fused optimizers step the whole arena: frozen (requires_grad=False) parameters are not supported")
        return not getattr(m, "_grads_fresh", False)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        m = self._model
        flat = m.flat_parameters()
        steps = 0
        for k in self._state_names:
            self._flat_state[k] = torch.zeros_like(flat)
        for (name, off, numel, shape), p in zip(m._pinfo, m._plist):
            st = self.state.get(p, {})
            if "step" in st:
                steps = max(steps, int(float(st["step"])))
            for k in self._state_names:
                if k in st and st[k] is not None:
                    m._view(self._flat_state[k], off, numel, shape).copy_(st[k])
        self._steps = steps
        self._loaded_state = any(k in self.state.get(p, {}) and self.state[p][k] is not None
                                 for p in m._plist for k in self._state_names)
        self._bind_views()


class Adam(_FlatOptimizer):
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False) as one fused launch."""
    _state_names = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                     foreach=None, capturable=False, differentiable=False, fused=None))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        m = self._model
        if self._skip_step():
            return
        self._ensure_state()
        g = self.param_groups[0]
        self._steps += 1
        p, gr = m.flat_parameters(), m.flat_gradients()
        N.ops().adam_step(p, gr, self._flat_state["exp_avg"], self._flat_state["exp_avg_sq"], float(g["lr"]), float(g["betas"][0]),
                          float(g["betas"][1]), float(g["eps"]), self._steps, float(grad_scale))

    def state_dict(self):
        for q in self._model._plist:  # materialise torch's per-parameter step counters only when somebody looks
            if q in self.state:
                self.state[q]["step"] = torch.tensor(float(self._steps))
        return super().state_dict()


class SGD(_FlatOptimizer):
    """torch.optim.SGD(lr, momentum) (dampening 0, no nesterov, no weight decay) as one fused launch."""
    _state_names = ("momentum_buffer",)
    _has_step = False

    def __init__(self, params, lr=1e-3, momentum=0.9):
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=0, nesterov=False, maximize=False,
                                     foreach=None, differentiable=False, fused=None))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        m = self._model
        if self._skip_step():
            return
        self._ensure_state()
        g = self.param_groups[0]
        # torch initialises momentum_buffer = grad on the first step of a FRESH optimizer; a buffer that came in through
        # load_state_dict (own or stock torch.optim.SGD checkpoint, which carries no step counter) continues as mu*buf + g
        first = 1 if (self._steps == 0 and not self._loaded_state) else 0
        self._steps += 1
        p, gr = m.flat_parameters(), m.flat_gradients()
        N.ops().sgd_step(p, gr, self._flat_state["momentum_buffer"], float(g["lr"]), float(g["momentum"]), bool(first), float(grad_scale))
