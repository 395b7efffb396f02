"""
AdamUniform (reference: largesteps/optimize.py:3-41): Adam whose per-element second-moment scaling is replaced
by one global max. Same constructor, state keys ("step", "g1", "g2") and update rule; the step itself runs as
two fused HIP kernels (csrc/adam.hip) instead of ~10 torch kernels and stays on the device.
"""
import torch

from . import _native

_scratch = {}


def _scratch_for(dev):
    s = _scratch.get(dev)
    if s is None:
        s = torch.empty(4096, dtype=torch.uint8, device=dev)
        _scratch[dev] = s
    return s


class AdamUniform(torch.optim.Optimizer):
    """
    Adam with one global step scale (reference: optimize.py:3-41): the first moment is divided by the LARGEST root second
    moment of the whole parameter instead of element by element, so every coordinate of a vertex (and every vertex) moves
    on the same scale. Constructor arguments and state keys ("step", "g1", "g2") are the reference's.

    capturable=True (as in torch.optim.Adam) keeps the step count on the device -- state["step"] is then an int32 tensor whose
    first element counts the steps -- so that `step()` can be recorded in a `torch.cuda.graph` together with the forward and
    backward pass and replayed; the update is the same.
    """
    def __init__(self, params, lr=0.1, betas=(0.9, 0.999), capturable=False):
        defaults = dict(lr=lr, betas=betas, capturable=bool(capturable))
        super(AdamUniform, self).__init__(params, defaults)

    def __setstate__(self, state):
        super(AdamUniform, self).__setstate__(state)
        self._restore_step_counters()

    def load_state_dict(self, state_dict):
        """torch.optim.Optimizer.load_state_dict casts a tensor-valued "step" to float32 when the group is capturable (and
        every other tensor state to the parameter's dtype); the kernel reads the counter as int32 -- convert it back BY VALUE."""
        super(AdamUniform, self).load_state_dict(state_dict)
        self._restore_step_counters()

    def _restore_step_counters(self):
        for st in self.state.values():
            step = st.get("step") if isinstance(st, dict) else None
            if torch.is_tensor(step) and step.dtype != torch.int32:
                st["step"] = step.round().to(torch.int32).contiguous()

    @torch.no_grad()
    def step(self):
        lib = _native.lib()
        for group in self.param_groups:
            lr = group['lr']
            b1, b2 = group['betas']
            for p in group["params"]:
                state = self.state[p]
                # Lazy initialization
                capturable = group.get("capturable", False)
                if len(state) == 0:
                    state["step"] = torch.zeros(2, dtype=torch.int32, device=p.device) if capturable else 0
                    state["g1"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                    state["g2"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                if not capturable:
                    state["step"] += 1
                _native.require_device(p.data, "AdamUniform parameter")
                if p.dtype != torch.float32:
                    raise TypeError(f"AdamUniform parameters must be float32, got {p.dtype}")
                if not p.data.is_contiguous():
                    raise ValueError("AdamUniform parameters must be contiguous")
                grad = p.grad.data.contiguous()
                dev = p.device
                if capturable:
                    step = state["step"]
                    if not (torch.is_tensor(step) and step.dtype == torch.int32 and step.numel() >= 2 and step.device == dev):
                        raise TypeError("AdamUniform(capturable=True): state['step'] must be an int32 tensor of two elements on the parameter's device")
                    # (no torch.cuda.device context: the native call selects `dev` itself)
                    _native.check(lib.ls_adam_uniform_step_device(_native.ptr(p.data), _native.ptr(grad), _native.ptr(state["g1"]),
                                                                  _native.ptr(state["g2"]), p.numel(), float(lr), float(b1), float(b2),
                                                                  _native.ptr(state["step"]), _native.ptr(_scratch_for(dev)), dev.index,
                                                                  _native.stream_of(dev)))
                    continue
                _native.check(lib.ls_adam_uniform_step(_native.ptr(p.data), _native.ptr(grad), _native.ptr(state["g1"]),
                                                       _native.ptr(state["g2"]), p.numel(), float(lr), float(b1), float(b2),
                                                       int(state["step"]), _native.ptr(_scratch_for(dev)), dev.index,
                                                       _native.stream_of(dev)))
