"""Batched policy inference on the device (SURVEY.md section 8f item 1 -- the caller side of the hot path).

The reference evaluates the policy once per robot with batch size 1 through a TF1 session
(agents/imitation_runners.py:88-92).  Here one batched MLP serves all robots of a GPU:
  pi: 160 -> 512 -> 256 -> 12, ReLU (run.py:101-105), fixed-variance diagonal Gaussian with std 0.125
  (agents/imitation_policies.py:44-51,96-107), optional value head vf: 160 -> 512 -> 256 -> 1.
Weights load from a stable-baselines zip (`parameters` npz inside the zip, keys `model/pi_fc0/w:0` ...;
stable_baselines/common/base_class.py:552-590) or from an npz with the same keys.
This is plain torch (hipBLASLt GEMMs): a dense contraction, but nowhere near the env step in cost.
"""
import io
import math
import zipfile

import numpy as np

PI_STD = 0.125  # imitation_policies.py:106 pi_init_std


def _norm_key(k):
    return k.replace("__", "/").replace("_0", ":0") if "__" in k else k


def load_parameters(path):
    """dict name -> float32 array from a stable-baselines zip or an npz."""
    if zipfile.is_zipfile(path):
        with zipfile.ZipFile(path) as z:
            names = z.namelist()
            if "parameters" in names:            # stable-baselines model zip
                params = np.load(io.BytesIO(z.read("parameters")))
                return {k: params[k].astype(np.float32) for k in params.files}
    params = np.load(path)
    return {_norm_key(k): params[k].astype(np.float32) for k in params.files}


# variables of the reference's ImitationPolicy graph in the order stable-baselines saves them (parameter_list of the shipped
# zips; tests/golden/policy_parameter_list.json).  `model/q/*` is the action-value head stable-baselines attaches to every
# feed-forward policy (common/policies.py proba_distribution_from_latent); PPO1 never trains or reads it, but
# BaseRLModel.load_parameters(exact_match=True) (common/base_class.py:437-500) refuses a zip without it.
SB_PARAMETER_LIST = (("model/pi_fc0/w:0", (160, 512)), ("model/pi_fc0/b:0", (512,)), ("model/vf_fc0/w:0", (160, 512)),
                     ("model/vf_fc0/b:0", (512,)), ("model/pi_fc1/w:0", (512, 256)), ("model/pi_fc1/b:0", (256,)),
                     ("model/vf_fc1/w:0", (512, 256)), ("model/vf_fc1/b:0", (256,)), ("model/vf/w:0", (256, 1)),
                     ("model/vf/b:0", (1,)), ("model/pi/w:0", (256, 12)), ("model/pi/b:0", (12,)), ("model/q/w:0", (256, 12)),
                     ("model/q/b:0", (12,)))


def save_parameters_zip(path, params, data=None):
    """Write weights in the stable-baselines zip layout (`data` JSON, `parameter_list` JSON, `parameters` npz;
    stable_baselines/common/base_class.py:552-590) so that the reference's `agent.load_parameters(model_file)`
    (run.py:220-221, exact_match=True) can read a policy trained here: the FULL variable set of the reference graph in its
    saved order.  The unused q head is carried over when `params` holds one (a policy warm-started from a reference zip) and
    zero-filled otherwise.  `data` carries no pickled objects (load_parameters ignores it)."""
    import json
    out = {}
    for name, shape in SB_PARAMETER_LIST:
        if name in params:
            v = np.asarray(params[name], dtype=np.float32)
            if v.shape != shape:
                raise ValueError("%s has shape %s, the reference graph expects %s" % (name, v.shape, shape))
        elif name.startswith("model/q/"):
            v = np.zeros(shape, dtype=np.float32)
        else:
            raise ValueError("missing variable %s" % name)
        out[name] = v
    names = [n for n, _ in SB_PARAMETER_LIST]
    buf = io.BytesIO()
    np.savez(buf, **out)
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("data", json.dumps(data or {}))
        z.writestr("parameter_list", json.dumps(names))
        z.writestr("parameters", buf.getvalue())


class MLPPolicy(object):
    def __init__(self, params, device, std=PI_STD):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.std = float(std)
        g = lambda k: torch.tensor(params[k], dtype=torch.float32, device=self.device)
        self.pi = [(g("model/pi_fc0/w:0"), g("model/pi_fc0/b:0")), (g("model/pi_fc1/w:0"), g("model/pi_fc1/b:0")),
                   (g("model/pi/w:0"), g("model/pi/b:0"))]
        self.vf = None
        if "model/vf_fc0/w:0" in params:
            self.vf = [(g("model/vf_fc0/w:0"), g("model/vf_fc0/b:0")), (g("model/vf_fc1/w:0"), g("model/vf_fc1/b:0")),
                       (g("model/vf/w:0"), g("model/vf/b:0"))]

    @classmethod
    def from_file(cls, path, device, std=PI_STD):
        return cls(load_parameters(path), device, std)

    def _mlp(self, layers, x):
        t = self.torch
        h = t.relu(t.addmm(layers[0][1], x, layers[0][0]))
        h = t.relu(t.addmm(layers[1][1], h, layers[1][0]))
        return t.addmm(layers[2][1], h, layers[2][0])

    def mean(self, obs):
        return self._mlp(self.pi, obs)

    def value(self, obs):
        if self.vf is None:
            return self.torch.zeros(obs.shape[0], device=obs.device)
        return self._mlp(self.vf, obs)[:, 0]

    def act(self, obs, deterministic=False, generator=None):
        """-> (clipped action [N,12] for env.step, unclipped action, value [N]).  Clip = action space +-2 pi
        (imitation_runners.py:140-143)."""
        t = self.torch
        mu = self.mean(obs)
        a = mu if deterministic else mu + self.std * t.randn(mu.shape, device=mu.device, generator=generator)
        return t.clamp(a, -2.0 * math.pi, 2.0 * math.pi), a, self.value(obs)

    def log_prob(self, obs, actions):
        t = self.torch
        mu = self.mean(obs)
        var = self.std * self.std
        return (-0.5 * ((actions - mu) ** 2) / var - 0.5 * math.log(2.0 * math.pi * var)).sum(dim=1)
