"""Fused actor / critic forward pass on the matrix cores (include/openroborl_policy.h, csrc/orr_policy.hip).

One launch per env step instead of the reference's one TF session call per robot (agents/imitation_runners.py:88-92)
and instead of ~15 torch kernels for the two 160 -> 512 -> 256 -> {12, 1} MLPs.  f32 in, f32 accumulate.
There is no fallback here: without the HIP library the constructor raises.
"""
import ctypes as C
import math

from . import _abi, _lib

KEYS = (("w0_pi", "model/pi_fc0/w:0"), ("b0_pi", "model/pi_fc0/b:0"), ("w1_pi", "model/pi_fc1/w:0"), ("b1_pi", "model/pi_fc1/b:0"),
        ("w2_pi", "model/pi/w:0"), ("b2_pi", "model/pi/b:0"), ("w0_vf", "model/vf_fc0/w:0"), ("b0_vf", "model/vf_fc0/b:0"),
        ("w1_vf", "model/vf_fc1/w:0"), ("b1_vf", "model/vf_fc1/b:0"), ("w2_vf", "model/vf/w:0"), ("b2_vf", "model/vf/b:0"))
SHAPES = {"w0_pi": (160, 512), "w0_vf": (160, 512), "w1_pi": (512, 256), "w1_vf": (512, 256), "w2_pi": (256, 12), "w2_vf": (256, 1)}


class FusedActorCritic(object):
    """Device-resident packed copy of an actor-critic parameter dict (stable-baselines names, torch tensors on the GPU).
    Call `refresh()` after the parameters changed (e.g. after an optimiser step) and `forward()` once per env step."""

    def __init__(self, params, device, std, clip=2.0 * math.pi):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FusedActorCritic needs a GPU device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.L = _lib.load()
        self.params = params
        self.std = float(std)
        self.clip = float(clip)
        self.packed = {}
        self.biases = {}
        for field, key in KEYS:
            w = params[key]
            if field.startswith("w"):
                k, n = SHAPES[field]
                if tuple(w.shape) != (k, n):
                    raise ValueError("%s has shape %s, the fused kernel is built for %s" % (key, tuple(w.shape), (k, n)))
                size = int(self.L.orr_policy_packed_size(k, n))
                self.packed[field] = torch.empty(size, dtype=torch.float32, device=self.device)
        self.net = _abi.OrrPolicyNet()
        self.refresh()

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def refresh(self):
        """Re-pack the weights (six small launches) and re-read the biases from the parameter tensors."""
        t = self.torch
        for field, key in KEYS:
            w = self.params[key].detach()
            if w.dtype != t.float32 or not w.is_contiguous() or w.device != self.device:
                w = w.to(device=self.device, dtype=t.float32).contiguous()
            if field.startswith("w"):
                k, n = w.shape
                _lib.check(self.L.orr_policy_pack(w.data_ptr(), int(k), int(n), self.packed[field].data_ptr(), self._stream()), self.L)
                setattr(self.net, field, self.packed[field].data_ptr())
            else:
                if field not in self.biases:          # own copy (stays valid while the optimiser updates the original) at a fixed address,
                    self.biases[field] = t.empty_like(w)   # so that a captured forward pass can be replayed after a refresh
                self.biases[field].copy_(w)
                setattr(self.net, field, self.biases[field].data_ptr())

    def forward(self, obs, noise=None, want_mean=False, out_action=None, out_raw=None, out_value=None):
        """obs [N,160] float32 on the device; noise [N,12] standard normal or None (deterministic).
        Returns (clipped action [N,12], raw action [N,12], value [N], mean [N,12] or None); the out_* tensors
        (contiguous float32 of those shapes, e.g. rows of a rollout buffer) are written in place when given."""
        t = self.torch
        if obs.dtype != t.float32 or not obs.is_contiguous() or obs.device != self.device or obs.dim() != 2 or obs.shape[1] != 160:
            raise ValueError("obs must be a contiguous float32 [N,160] tensor on %s" % (self.device,))
        n = obs.shape[0]
        if noise is not None and (noise.dtype != t.float32 or not noise.is_contiguous() or tuple(noise.shape) != (n, 12)
                                  or noise.device != self.device):
            raise ValueError("noise must be a contiguous float32 [N,12] tensor on the same device")
        def out(x, shape):
            if x is None:
                return t.empty(shape, dtype=t.float32, device=self.device)
            if x.dtype != t.float32 or not x.is_contiguous() or tuple(x.shape) != shape or x.device != self.device:
                raise ValueError("output tensor must be contiguous float32 %s on %s" % (shape, self.device))
            return x
        act, raw, val = out(out_action, (n, 12)), out(out_raw, (n, 12)), out(out_value, (n,))
        mean = t.empty(n, 12, dtype=t.float32, device=self.device) if want_mean else None
        _lib.check(self.L.orr_policy_forward(C.byref(self.net), obs.data_ptr(), int(n), noise.data_ptr() if noise is not None else None,
                                             self.std, self.clip, act.data_ptr(), raw.data_ptr(), val.data_ptr(),
                                             mean.data_ptr() if mean is not None else None, self._stream()), self.L)
        return act, raw, val, mean
