"""PPO learner in PyTorch-ROCm -- the consumer of rollout.py's [T, N] buffers.

Restates the update of agents/ppo_imitation.py:156-258,352-382 (stable-baselines PPO1 graph) without TF1/MPI:
  * actor 160 -> 512 -> 256 -> 12 and critic 160 -> 512 -> 256 -> 1, ReLU (run.py:101-105), fixed-variance diagonal
    Gaussian, std 0.125 (agents/imitation_policies.py:44-51,96-107);
  * clipped surrogate with clip_param 0.2, value loss mean((v - tdlamret)^2), Adam lr 1e-5, eps 1e-5, one epoch over
    the segment (run.py:111-125; ppo1/pposgd_simple.py loss terms);
  * data-parallel: one process per GPU, gradients averaged with ONE all-reduce of a flat 1.7 MB buffer per minibatch
    (RCCL over xGMI; replaces MpiAdam's Allreduce, stable_baselines/common/mpi_adam.py:40-62).
The env never leaves the device, so a full iteration is: collect_rollout -> gae -> normalize_per_robot -> update.
"""
import math

import numpy as np

from . import policy as pol


_LINEAR = {}


def _wgrad(torch, x, g):
    """x^T g for a tall batch: hipBLASLt picks a 32x32 macro-tile for these (K = batch) shapes and fills a third of the
    chip; a 16-way split over the batch (one bmm + a sum) is 2.5-3x faster on MI355X (100 -> 35 us at 16384 x 160 x 512)."""
    B = x.shape[0]
    if B >= 4096 and B % 16 == 0:
        return torch.bmm(x.view(16, B // 16, x.shape[1]).transpose(1, 2), g.view(16, B // 16, g.shape[1])).sum(0)
    return x.t() @ g


def _linear(torch, relu):
    """x W + b (optionally ReLU) as one GEMM with a bias (+ ReLU) epilogue; the backward is spelled out because
    torch._addmm_activation has no autograd formula and to route the weight gradient through _wgrad."""
    if relu not in _LINEAR:
        class Linear(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w, b):
                h = torch._addmm_activation(b, x, w) if relu else torch.addmm(b, x, w)
                ctx.save_for_backward(x, w, h if relu else b)
                return h

            @staticmethod
            def backward(ctx, gh):
                x, w, h = ctx.saved_tensors
                gz = gh * (h > 0).to(gh.dtype) if relu else gh
                gx = gz @ w.t() if ctx.needs_input_grad[0] else None
                return gx, _wgrad(torch, x, gz.contiguous()), gz.sum(dim=0)
        _LINEAR[relu] = Linear.apply
    return _LINEAR[relu]


class ActorCritic(object):
    """Trainable twin of policy.MLPPolicy (same parameter names as the stable-baselines zips)."""

    def __init__(self, device, obs_dim=160, act_dim=12, hidden=(512, 256), std=pol.PI_STD, params=None, seed=0):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self._std = float(std)
        g = torch.Generator().manual_seed(seed)
        self.p = {}
        dims = [obs_dim] + list(hidden)
        for net, out in (("pi", act_dim), ("vf", 1)):
            for i in range(len(hidden)):
                self._init("model/%s_fc%d" % (net, i), dims[i], dims[i + 1], g, math.sqrt(2.0))
            self._init("model/%s" % net, dims[-1], out, g, 0.01 if net == "pi" else 1.0)
        self.extra = {}            # variables of a loaded zip that this learner does not train (model/q/*): re-emitted by state_dict()
        if params is not None:
            for k, v in params.items():
                if k in self.p:
                    self.p[k].data.copy_(torch.as_tensor(np.asarray(v), dtype=torch.float32))
                else:
                    self.extra[k] = np.asarray(v, dtype=np.float32).copy()
        for v in self.p.values():
            v.requires_grad_(True)
        self.fused = None          # policy_hip.FusedActorCritic once enable_fused() was called
        self._fused_dirty = True

    @property
    def std(self):
        return self._std

    @std.setter
    def std(self, value):
        """One number for the sampler AND the log-probabilities: the fused forward pass takes std by value at every launch, so a policy
        whose std is changed after enable_fused() must hand it on (a re-captured rollout graph sampled with the old std while
        the log-probabilities used the new one)."""
        self._std = float(value)
        if getattr(self, "fused", None) is not None:
            self.fused.std = self._std

    def enable_fused(self):
        """Route act() through the fused matrix-core forward pass (csrc/orr_policy.hip).  The packed weight copy is
        refreshed lazily after mark_updated() (the learner calls it after every optimiser step)."""
        from . import policy_hip
        self.fused = policy_hip.FusedActorCritic(self.p, self.device, std=self.std)
        self._fused_dirty = False
        return self

    def mark_updated(self):
        self._fused_dirty = True

    def _init(self, name, fan_in, fan_out, g, gain):
        t = self.torch
        w = t.randn(fan_in, fan_out, generator=g) * (gain / math.sqrt(fan_in))
        self.p[name + "/w:0"] = w.to(self.device)
        self.p[name + "/b:0"] = t.zeros(fan_out, device=self.device)

    def parameters(self):
        return [self.p[k] for k in sorted(self.p)]

    def _mlp(self, net, x):
        t = self.torch   # bias and ReLU ride in the GEMM epilogue (hipBLASLt): one kernel per layer instead of three
        f, out = _linear(t, True), _linear(t, False)
        h = f(x, self.p["model/%s_fc0/w:0" % net], self.p["model/%s_fc0/b:0" % net])
        h = f(h, self.p["model/%s_fc1/w:0" % net], self.p["model/%s_fc1/b:0" % net])
        return out(h, self.p["model/%s/w:0" % net], self.p["model/%s/b:0" % net])

    def mean(self, obs):
        return self._mlp("pi", obs)

    def value(self, obs):
        return self._mlp("vf", obs)[:, 0]

    def act(self, obs, deterministic=False, generator=None, noise=None, out_raw=None, out_value=None):
        """-> (clipped action, raw action, value).  `noise` ([N,12] standard normal) overrides the generator."""
        t = self.torch
        if self.fused is not None:
            if self._fused_dirty:
                self.fused.refresh()
                self._fused_dirty = False
            if noise is None and not deterministic:
                noise = t.randn((obs.shape[0], 12), device=obs.device, generator=generator)
            a, raw, v, _ = self.fused.forward(obs, None if deterministic else noise, out_raw=out_raw, out_value=out_value)
            return a, raw, v
        with t.no_grad():
            mu = self.mean(obs)
            if noise is None and not deterministic:
                noise = t.randn(mu.shape, device=mu.device, generator=generator)
            a = mu if deterministic else mu + self.std * noise
            return t.clamp(a, -2.0 * math.pi, 2.0 * math.pi), a, self.value(obs)

    def log_prob(self, obs, actions):
        mu = self.mean(obs)
        var = self.std * self.std
        return (-0.5 * ((actions - mu) ** 2) / var - 0.5 * math.log(2.0 * math.pi * var)).sum(dim=1)

    def state_dict(self):
        d = {k: v.detach().cpu().numpy() for k, v in self.p.items()}
        d.update(self.extra)
        return d


class PPO(object):
    def __init__(self, model, clip_param=0.2, lr=1e-5, adam_eps=1e-5, minibatch=4096, vf_coef=1.0, group=None):
        import torch
        self.torch = torch
        self.model = model
        self.clip = clip_param
        self.minibatch = int(minibatch)
        self.vf_coef = vf_coef
        self.group = group
        params = model.parameters()
        # one fused multi-tensor kernel per step on the GPU (the default foreach path is ~7 launches, 0.2 ms per step)
        self.opt = torch.optim.Adam(params, lr=lr, eps=adam_eps, fused=bool(params and params[0].is_cuda))

    def _allreduce_grads(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        t = self.torch
        params = [p for p in self.model.parameters() if p.grad is not None]
        flat = t.cat([p.grad.reshape(-1) for p in params])
        dist.all_reduce(flat, group=self.group)
        flat /= dist.get_world_size(self.group)
        off = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n

    def update(self, obs, actions, adv, ret, old_logp=None, epochs=1, generator=None):
        """obs [B,160], actions [B,12] (unclipped samples), adv [B] (already normalised), ret [B] (TD(lambda) targets)."""
        t = self.torch
        B = obs.shape[0]
        if old_logp is None:
            with t.no_grad():
                old_logp = self.model.log_prob(obs, actions)
        stats = []
        for _ in range(epochs):
            perm = t.randperm(B, device=obs.device, generator=generator)
            # one gather per epoch; the minibatches are then contiguous views
            obs_p, act_p, adv_p, ret_p, old_p = obs[perm], actions[perm], adv[perm], ret[perm], old_logp[perm]
            for s in range(0, B, self.minibatch):
                e = s + self.minibatch
                logp = self.model.log_prob(obs_p[s:e], act_p[s:e])
                ratio = t.exp(logp - old_p[s:e])
                a = adv_p[s:e]
                surr = -t.min(ratio * a, t.clamp(ratio, 1.0 - self.clip, 1.0 + self.clip) * a).mean()
                vf = ((self.model.value(obs_p[s:e]) - ret_p[s:e]) ** 2).mean()
                loss = surr + self.vf_coef * vf
                self.opt.zero_grad(set_to_none=True)
                loss.backward()
                self._allreduce_grads()
                self.opt.step()
                if hasattr(self.model, "mark_updated"):
                    self.model.mark_updated()
                stats.append(t.stack((surr.detach(), vf.detach())))   # stays on the device: no sync per minibatch
        return t.stack(stats).mean(dim=0).cpu().numpy()
