"""Device-resident rollout buffers + per-robot GAE.

Replaces the host-side trajectory generator (agents/imitation_runners.py:22-208) and
`add_vtarg_and_adv` (agents/ppo_imitation.py:68-93) plus the per-robot advantage normalisation
(ppo_imitation.py:329-338) with [T, N] tensors that never leave the GPU:

  obs [T,N,160], actions [T,N,12], rewards [T,N], dones [T,N], vpred [T,N], next_vpred [T,N]

Semantics kept from the reference: gamma = lam = 0.95 (run.py:113,120); the value after a finished episode is
0 and -- the reference's quirk -- so is the bootstrap value at the end of a segment
(imitation_runners.py:98-100 `last_vpred = 0.0`); advantages are standardised per robot over the segment.
Deliberate divergences: robots reset individually (auto-reset inside env.step) instead of the whole env
resetting when any robot is done, and the GAE recursion uses each robot's own done flags by default (the reference
indexes `episode_starts[(step*num_robot+i) + (1+i)]`, ppo_imitation.py:88, which reads a neighbouring
robot's flag; `legacy_gae_index=True` reproduces exactly that, for comparisons with the reference at num_robot > 1).
"""


def collect_rollout(env, policy, horizon, obs=None, deterministic=False, generator=None, noise=None):
    """Run `horizon` env steps of all robots.  Returns a dict of [T, N, ...] tensors + the final observation.
    noise: optional [T, N, 12] standard-normal samples for the fused policy (default: drawn from `generator`)."""
    t = env.torch
    n = env.num_robot
    dev = env.device
    if obs is None:
        obs = env.reset()
    buf = {
        "obs": t.empty((horizon, n, obs.shape[1]), device=dev), "actions": t.empty((horizon, n, 12), device=dev),
        "rewards": t.empty((horizon, n), device=dev), "dones": t.empty((horizon, n), dtype=t.bool, device=dev),
        "vpred": t.empty((horizon, n), device=dev),
    }
    fused = getattr(policy, "fused", None) is not None
    if fused and not deterministic and noise is None:   # one launch for the whole segment's exploration noise
        noise = t.randn((horizon, n, 12), device=dev, generator=generator)
    if not fused or deterministic:
        noise = None
    for k in range(horizon):
        if fused:   # one fused launch writes the raw action and the value straight into the buffers
            clipped, _, _ = policy.act(obs, deterministic=deterministic, noise=None if noise is None else noise[k],
                                       out_raw=buf["actions"][k], out_value=buf["vpred"][k])
            buf["obs"][k].copy_(obs)
        else:
            clipped, raw, v = policy.act(obs, deterministic=deterministic, generator=generator)
            buf["obs"][k].copy_(obs)
            buf["actions"][k].copy_(raw)
            buf["vpred"][k].copy_(v)
        obs, rew, done, _ = env.step(clipped.contiguous())
        buf["rewards"][k].copy_(rew)
        buf["dones"][k].copy_(done.bool())
    buf["last_obs"] = obs
    if noise is not None:
        # log-probability of the sampled actions under the sampling policy: a - mean = std * noise
        import math
        std = float(policy.std)
        buf["logp"] = -0.5 * (noise * noise).sum(dim=-1) - 12.0 * math.log(std * math.sqrt(2.0 * math.pi))
    return buf


class GraphRollout(object):
    """collect_rollout() for the fused policy with static [T, N] buffers: the policy forward pass writes actions and values, and
    env.step_into writes observations, rewards and done flags, straight into their rows, and the segment's launches (weight
    re-pack, T x (policy, env step), log-probabilities) are captured once and replayed as ONE hipGraph.  The first segment runs
    eagerly (first launches load code objects, which a capture must not do); the capture happens on the second call.

    Kernel arguments that are passed by value - the env's configuration incl. the seed (env.seed()), the policy's std - are frozen
    into a captured graph; collect() notices when they change (env.launch_params_generation, policy.std) and captures again.

    The returned tensors are views of the static buffers: they are overwritten by the next collect() (clone what must survive it,
    e.g. the last row of `dones` that the next segment's GAE takes as first_starts)."""

    def __init__(self, env, policy, horizon):
        t = env.torch
        if getattr(policy, "fused", None) is None:
            raise RuntimeError("GraphRollout needs the fused policy (ActorCritic.enable_fused()); the plain path is collect_rollout()")
        self.env, self.policy, self.T = env, policy, int(horizon)
        n, dev, T = env.num_robot, env.device, self.T
        f = lambda *shape: t.empty(shape, dtype=t.float32, device=dev)   # noqa: E731
        self.obs, self.actions, self.clipped = f(T + 1, n, 160), f(T, n, 12), f(n, 12)
        self.rewards, self.vpred, self.noise, self.logp = f(T, n), f(T, n), f(T, n, 12), f(T, n)
        self.dones = t.zeros((T, n), dtype=t.uint8, device=dev)
        self.graph, self.calls = None, 0
        self._captured_for = None          # (env.launch_params_generation, policy.std) the graph was captured with

    def _launch_params(self):
        return (getattr(self.env, "launch_params_generation", 0), float(self.policy.std))

    def _segment(self):
        import math
        t, env, fused = self.env.torch, self.env, self.policy.fused
        fused.refresh()                   # inside the graph: every replay re-packs the weights the learner has just updated
        for k in range(self.T):
            fused.forward(self.obs[k], self.noise[k], out_action=self.clipped, out_raw=self.actions[k], out_value=self.vpred[k])
            env.step_into(self.clipped, self.obs[k + 1], self.rewards[k], self.dones[k])
        # log-probability of the sampled actions under the sampling policy: a - mean = std * noise
        t.sum(self.noise * self.noise, dim=-1, out=self.logp)
        self.logp.mul_(-0.5).sub_(12.0 * math.log(float(self.policy.std) * math.sqrt(2.0 * math.pi)))

    def collect(self, obs, generator=None, noise=None):
        t, env = self.env.torch, self.env
        with t.no_grad():
            if obs.data_ptr() != self.obs[0].data_ptr():
                self.obs[0].copy_(obs)
            if noise is not None:
                self.noise.copy_(noise)
            else:
                self.noise.normal_(generator=generator)
            self.calls += 1
            if self.calls == 1:
                self._segment()
            else:
                # the captured launches carry the env's configuration (seed) and the policy's std as BY-VALUE kernel arguments: a graph
                # captured before env.seed(new) / a new std would silently replay the old values - drop it and capture again
                if self.graph is not None and self._captured_for != self._launch_params():
                    self.graph = None
                if self.graph is None:
                    self._captured_for = self._launch_params()
                    counter = env._env_step_counter
                    self.graph = t.cuda.CUDAGraph()
                    with t.cuda.graph(self.graph):
                        self._segment()
                    env._env_step_counter = counter          # the capture launched nothing
                self.graph.replay()
                env._env_step_counter += self.T
            self.policy._fused_dirty = False
        return {"obs": self.obs[:self.T], "actions": self.actions, "rewards": self.rewards, "dones": self.dones.view(t.bool),
                "vpred": self.vpred, "logp": self.logp, "last_obs": self.obs[self.T]}


def legacy_nonterminal(dones, first_starts=None):
    """The flags the reference's recursion actually reads for num_robot > 1: `1 - episode_starts[(step*N+i) + (1+i)]` of the flat
    [T*N] (+ N x False) array (agents/ppo_imitation.py:75,88), with episode_starts[t] = done[t-1] (imitation_runners.py:178) and
    `first_starts` ([N] bool, default all True) for the first step of the segment."""
    import torch
    T, n = dones.shape
    starts = torch.empty((T + 1, n), dtype=torch.bool, device=dones.device)
    starts[0] = True if first_starts is None else first_starts.to(torch.bool)
    starts[1:T] = dones[:-1]
    flat = torch.cat([starts[:T].reshape(-1), torch.zeros(2 * n, dtype=torch.bool, device=dones.device)])
    idx = (torch.arange(T, device=dones.device)[:, None] * n + 2 * torch.arange(n, device=dones.device)[None, :] + 1)
    return ~flat[idx]


def gae(rewards, vpred, dones, gamma=0.95, lam=0.95, bootstrap=None, legacy_gae_index=False, first_starts=None):
    """Per-robot GAE(lambda) on [T, N] tensors.  next value = vpred[t+1] unless the episode ended at t (then 0);
    at the end of the segment `bootstrap` ([N], default 0 like the reference).  legacy_gae_index=True reproduces the reference's
    neighbouring-robot flag index in the recursion (see legacy_nonterminal; identical for N = 1)."""
    import torch
    T, n = rewards.shape
    nonterminal = (~dones).to(rewards.dtype)
    nxt = torch.empty_like(vpred)
    nxt[:-1] = vpred[1:]
    nxt[-1] = 0.0 if bootstrap is None else bootstrap
    nxt = nxt * nonterminal
    delta = rewards + gamma * nxt - vpred
    rec = legacy_nonterminal(dones, first_starts).to(rewards.dtype) if legacy_gae_index else nonterminal
    adv = torch.empty_like(rewards)
    last = torch.zeros(n, dtype=rewards.dtype, device=rewards.device)
    for k in range(T - 1, -1, -1):
        last = delta[k] + gamma * lam * rec[k] * last
        adv[k] = last
    return adv, adv + vpred


def normalize_per_robot(adv, eps=0.0):
    """ppo_imitation.py:329-338: (a - mean) / std over each robot's own samples (population std, like numpy)."""
    return (adv - adv.mean(dim=0, keepdim=True)) / (adv.std(dim=0, unbiased=False, keepdim=True) + eps)


def gae_fused(rewards, vpred, dones, gamma=0.95, lam=0.95, bootstrap=None, normalize=True, eps=0.0, legacy_gae_index=False, first_starts=None):
    """gae() + normalize_per_robot() in one HIP launch (include/openroborl_policy.h: orr_gae_flags).  GPU tensors only;
    returns (advantages [T,N] - standardised per robot when `normalize` -, TD(lambda) targets [T,N])."""
    import ctypes as C
    import torch
    from . import _lib
    L = _lib.load()
    T, n = rewards.shape
    rewards, vpred = rewards.contiguous(), vpred.contiguous()
    d8 = dones.to(torch.uint8).contiguous()
    adv, ret = torch.empty_like(rewards), torch.empty_like(rewards)
    boot = None if bootstrap is None else bootstrap.to(rewards.dtype).contiguous()
    stream = C.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream)
    fs = None if first_starts is None else first_starts.to(torch.uint8).contiguous()
    flags = (1 if normalize else 0) | (2 if legacy_gae_index else 0)          # ORR_GAE_NORMALIZE | ORR_GAE_LEGACY_INDEX
    _lib.check(L.orr_gae_flags(rewards.data_ptr(), vpred.data_ptr(), d8.data_ptr(), None if fs is None else fs.data_ptr(),
                               None if boot is None else boot.data_ptr(), int(T), int(n), float(gamma), float(lam), flags, float(eps),
                               adv.data_ptr(), ret.data_ptr(), stream), L)
    return adv, ret
