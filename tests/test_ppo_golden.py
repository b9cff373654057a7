"""Per-robot GAE + advantage standardisation against vectors produced by executing the reference's own
`add_vtarg_and_adv` / standardisation source (tests/golden/make_golden_ppo.py, fixture ppo_gae.npz).

num_robot = 1: the reference's index arithmetic is exact and both the torch restatement (rollout.gae, CPU) and the HIP launch
(orr_gae, GPU) must reproduce it.  num_robot = 3: ppo_imitation.py:88 reads `episode_starts[(step*N+i) + (1+i)]`, i.e. a
NEIGHBOURING robot's flag; the fixture pins that quirk (restated below), and this implementation deliberately uses each
robot's own flags instead."""
import os

import numpy as np
import pytest

from tests import oracle_lib as ol

G = np.load(os.path.join(ol.GOLDEN, "ppo_gae.npz"))
N1 = [c for c in G["cases"] if c.startswith("n1_")]


def _case(name):
    return {k: G["%s/%s" % (name, k)] for k in ("rewards", "vpred", "dones", "adv", "tdlamret", "adv_normalized")}


@pytest.mark.parametrize("name", N1)
def test_torch_gae_matches_reference_for_one_robot(name):
    import torch
    from openroborl_amd import rollout
    c = _case(name)
    adv, ret = rollout.gae(torch.tensor(c["rewards"]), torch.tensor(c["vpred"]), torch.tensor(c["dones"]), 0.95, 0.95)
    np.testing.assert_allclose(adv.numpy(), c["adv"], atol=3e-6)
    np.testing.assert_allclose(ret.numpy(), c["tdlamret"], atol=3e-6)
    nrm = rollout.normalize_per_robot(torch.tensor(c["adv"]))
    np.testing.assert_allclose(nrm.numpy(), c["adv_normalized"], atol=2e-5)


def test_reference_quirk_for_three_robots_is_understood():
    """The fixture for num_robot = 3 equals the recursion with the neighbouring-flag index, and differs from per-robot GAE."""
    import torch
    from openroborl_amd import rollout
    c = _case("n3_T40")
    T, n = c["rewards"].shape
    starts = np.zeros((T, n), dtype=bool)
    starts[0] = True
    starts[1:] = c["dones"][:-1]
    flat = np.append(starts.reshape(-1), [False] * n)
    nextv = np.zeros((T, n), dtype=np.float32)
    nextv[:-1] = np.where(c["dones"][:-1], 0.0, c["vpred"][1:])
    adv = np.zeros((T, n), dtype=np.float32)
    last = np.zeros(n)
    for t in reversed(range(T)):
        for i in range(n):
            nonterminal = 1.0 - float(flat[(t * n + i) + (1 + i)])            # ppo_imitation.py:88
            delta = c["rewards"][t, i] + 0.95 * nextv[t, i] - c["vpred"][t, i]
            last[i] = delta + 0.95 * 0.95 * nonterminal * last[i]
            adv[t, i] = last[i]
    np.testing.assert_allclose(adv, c["adv"], atol=3e-6)
    mine, _ = rollout.gae(torch.tensor(c["rewards"]), torch.tensor(c["vpred"]), torch.tensor(c["dones"]), 0.95, 0.95)
    assert np.abs(mine.numpy() - c["adv"]).max() > 1e-2                          # the deliberate divergence is real
    # the standardisation itself has no quirk: per-robot (population) mean / std for any num_robot
    nrm = rollout.normalize_per_robot(torch.tensor(c["adv"]))
    np.testing.assert_allclose(nrm.numpy(), c["adv_normalized"], atol=2e-5)


def test_legacy_gae_index_reproduces_the_reference_for_three_robots():
    """legacy_gae_index=True: the reference's own output for num_robot = 3 (fixture n3_T40, produced by executing the source text of
    add_vtarg_and_adv), not the per-robot recursion."""
    import torch
    from openroborl_amd import rollout
    c = _case("n3_T40")
    adv, ret = rollout.gae(torch.tensor(c["rewards"]), torch.tensor(c["vpred"]), torch.tensor(c["dones"]), 0.95, 0.95, legacy_gae_index=True)
    np.testing.assert_allclose(adv.numpy(), c["adv"], atol=3e-6)
    np.testing.assert_allclose(ret.numpy(), c["tdlamret"], atol=3e-6)
    for name in N1:         # for one robot the switch changes nothing
        c1 = _case(name)
        a1, _ = rollout.gae(torch.tensor(c1["rewards"]), torch.tensor(c1["vpred"]), torch.tensor(c1["dones"]), 0.95, 0.95, legacy_gae_index=True)
        np.testing.assert_allclose(a1.numpy(), c1["adv"], atol=3e-6)


@pytest.mark.gpu
def test_hip_legacy_gae_index_matches_reference_for_three_robots():
    import torch
    from openroborl_amd import rollout
    c = _case("n3_T40")
    dev = torch.device("cuda:0")
    r, v, d = (torch.tensor(c[k], device=dev) for k in ("rewards", "vpred", "dones"))
    adv, ret = rollout.gae_fused(r, v, d, 0.95, 0.95, normalize=False, legacy_gae_index=True)
    np.testing.assert_allclose(adv.cpu().numpy(), c["adv"], atol=3e-6)
    np.testing.assert_allclose(ret.cpu().numpy(), c["tdlamret"], atol=3e-6)
    nrm, _ = rollout.gae_fused(r, v, d, 0.95, 0.95, normalize=True, legacy_gae_index=True)
    np.testing.assert_allclose(nrm.cpu().numpy(), c["adv_normalized"], atol=5e-5)
    # first_starts is honoured: with "no episode starts at the segment's first step" only robots whose index lands in step 0 change
    fs = torch.zeros(3, dtype=torch.bool, device=dev)
    a2, _ = rollout.gae_fused(r, v, d, 0.95, 0.95, normalize=False, legacy_gae_index=True, first_starts=fs)
    a2t, _ = rollout.gae(r.cpu(), v.cpu(), d.cpu(), 0.95, 0.95, legacy_gae_index=True, first_starts=fs.cpu())
    np.testing.assert_allclose(a2.cpu().numpy(), a2t.numpy(), atol=3e-6)
    # and the default path is untouched
    mine, _ = rollout.gae_fused(r, v, d, 0.95, 0.95, normalize=False)
    ref, _ = rollout.gae(r.cpu(), v.cpu(), d.cpu(), 0.95, 0.95)
    np.testing.assert_allclose(mine.cpu().numpy(), ref.numpy(), atol=3e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", N1)
def test_hip_gae_matches_reference_for_one_robot(name):
    import torch
    from openroborl_amd import rollout
    c = _case(name)
    dev = torch.device("cuda:0")
    r, v, d = (torch.tensor(c[k], device=dev) for k in ("rewards", "vpred", "dones"))
    adv, ret = rollout.gae_fused(r, v, d, 0.95, 0.95, normalize=False)
    np.testing.assert_allclose(adv.cpu().numpy(), c["adv"], atol=3e-6)
    np.testing.assert_allclose(ret.cpu().numpy(), c["tdlamret"], atol=3e-6)
    nrm, _ = rollout.gae_fused(r, v, d, 0.95, 0.95, normalize=True)
    np.testing.assert_allclose(nrm.cpu().numpy(), c["adv_normalized"], atol=5e-5)
