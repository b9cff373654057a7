"""GPU tests at BASELINE.json's full sizes and over long horizons (run with -m gpu on an MI355X).

  * size-independent properties for configs[1] (4096 Laikago), configs[2] (4096 mini-cheetah) and configs[4] (8192
    interleaved Laikago / mini-cheetah): finite, reward in [0, 1], counters exact, bitwise reproducible;
  * row J: the device-side episode log against the per-step outputs and against the oracle's episodes,
    and the rollout-boundary gather on top of it;
  * a 2000-step soak at 4096 robots with invariants accumulated on the device;
  * which robots may disagree with the oracle on `done`: only those the oracle itself puts within a small margin of a
    termination threshold.
"""
import numpy as np
import pytest

from openroborl_amd import _abi, state as statemod
from tests import oracle_lib as ol
from tests.test_gpu_parity import make_pair, gpu_state64

pytestmark = pytest.mark.gpu

FULL = {
    "laikago4096": dict(robot="laikago", n=4096, mixed=None),
    "minicheetah4096": dict(robot="mini_cheetah", n=4096, mixed=None),
    "mixed8192": dict(robot=None, n=8192, mixed=["laikago", "mini_cheetah"]),
    # four dispatch rounds of the two-waves-per-SIMD kernel (its priority alternation keys on the round), and a batch that does not
    # fill the last wave
    "laikago16386": dict(robot="laikago", n=16386, mixed=None),
}


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_size_properties(name):
    import torch
    spec = FULL[name]
    n = spec["n"]
    outs = []
    for rep in range(2):
        env, orc = make_pair(spec["robot"] or "laikago", n=n, randomizer=True, auto_reset=True, mode="train", seed=21, mixed=spec["mixed"])
        orc.close()
        obs = env.reset()
        g = torch.Generator(device="cpu"); g.manual_seed(0)
        total_done = torch.zeros((), dtype=torch.int64, device=env.device)
        bad = torch.zeros((), dtype=torch.int64, device=env.device)
        for k in range(45):
            a = (torch.randn(n, 12, generator=g) * 0.125).to(env.device)
            obs, rew, done, _ = env.step(a)
            bad += (~torch.isfinite(obs)).sum() + (~torch.isfinite(rew)).sum() + (rew < 0).sum() + (rew > 1.0 + 1e-6).sum()
            total_done += done.sum()
        torch.cuda.synchronize()
        assert int(bad) == 0
        total_done = int(total_done)
        cnt = env.counters.cpu().numpy()
        assert cnt[_abi.CNT_TOTAL_STEP_COUNT] == total_done          # wrapper_env.py:82-83 per reset robot
        assert cnt[_abi.CNT_TOTAL_TIMESTEPS] == 45 * n
        assert cnt[_abi.CNT_TICKET] == 0 and cnt[_abi.CNT_DONE_ACCUM] == 0
        assert cnt[_abi.CNT_EPISODES] == total_done and cnt[_abi.CNT_EPLOG_DROPPED] == 0
        assert total_done >= 2 * n                                     # curriculum start: 20-step episodes, 45 steps
        ep = env.field_int("EP_STEP")[:, 0].cpu().numpy()
        assert ep.max() < 20 and ep.min() >= 0
        if spec["mixed"]:
            rt = env.field_int("ROBOT_TYPE")[:, 0].cpu().numpy()
            assert (rt == np.arange(n) % 2).all()                      # both models in every wavefront
            # mini-cheetah joints sit at different default angles: the two halves really ran different models
            q = env.field("Q").cpu().numpy()
            assert abs(q[0::2, 1].mean() - q[1::2, 1].mean()) > 0.3
        low = torch.tensor(env.observation_space.low[48:84], device=env.device)
        assert (obs[:, 48:84] >= low - 1e-5).all()
        outs.append((obs.cpu().numpy().copy(), env.state.cpu().numpy().copy()))
        env.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])              # bitwise reproducible for a fixed seed
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


def test_episode_log_row_j():
    """ppo_imitation.py:405-423 / imitation_runners.py:185-197: every finished episode is logged once as (return, length)."""
    import torch
    from openroborl_amd import dist as odist
    n, steps = 256, 25
    env, orc0 = make_pair("laikago", n=n, randomizer=True, auto_reset=True, mode="train", seed=31)
    orc0.close()
    orc = ol.OracleEnv(env.cfg, env.models, env.clips, n, robot_type=env.robot_type, clip_id=env.clip_id, threads=8, ep_log_capacity=4 * n)
    env.reset(); orc.reset()
    orc.state[:] = gpu_state64(env)
    rng = np.random.RandomState(2)
    ret_g, len_g, ret_o, len_o = np.zeros(n), np.zeros(n, dtype=int), np.zeros(n), np.zeros(n, dtype=int)
    exp_g, exp_o = [], []               # episodes reconstructed from the per-step (reward, done) outputs
    agree = np.ones(n, dtype=bool)      # done history identical on both sides so far
    pairs = []                          # (robot, device return, oracle return, length) of episodes with identical done history
    for k in range(steps):
        a = rng.uniform(-0.1, 0.1, (n, 12)).astype(np.float32)
        og, rg, dg, _ = env.step(torch.from_numpy(a).to(env.device))
        oo, ro, do = orc.step(a.astype(np.float64))
        rg, dg = rg.cpu().numpy().astype(np.float64), dg.cpu().numpy().astype(bool)
        ret_g += rg; len_g += 1; ret_o += ro; len_o += 1
        for i in np.nonzero(dg)[0]:
            exp_g.append((len_g[i], ret_g[i]))
        for i in np.nonzero(do)[0]:
            exp_o.append((len_o[i], ret_o[i]))
        for i in np.nonzero(dg & do & agree)[0]:
            pairs.append((i, ret_g[i], ret_o[i], len_g[i]))
        agree &= (dg == do)
        ret_g[dg] = 0; len_g[dg] = 0          # each side's episodes end on its own done flags
        ret_o[do] = 0; len_o[do] = 0
        # after a disagreement the two sides are in different episodes: re-synchronise the oracle to the device (state and episode
        # bookkeeping: the state record carries the episode step counter and the running return)
        if (dg != do).any():
            orc.state[:] = gpu_state64(env)
            dis = dg != do
            ret_o[dis] = ret_g[dis]; len_o[dis] = len_g[dis]
    torch.cuda.synchronize()
    n_eps = int(env.counters[_abi.CNT_EPISODES].item())
    assert n_eps == len(exp_g) and n_eps >= n          # 20-step limit: every robot finishes at least once in 25 steps
    # the one-launch pack (orr_episode_stats) against the torch restatement of the payload on the same log
    cnt_dev = torch.clamp(env.counters[_abi.CNT_EPISODES].clone(), max=env.ep_log.shape[0])
    ref_buf = odist.pack_episode_stats(env.ep_log[:, 0].clone(), env.ep_log[:, 1].clone(), steps * n, env.counters[_abi.CNT_EPLOG_DROPPED].clone(),
                                       64, count=cnt_dev)
    state = (env.counters.clone(), env.ep_log.clone())
    buf = env.episode_stats_packed(steps * n, 64)          # capacity 64 < number of episodes: the list is truncated, the sums are not
    np.testing.assert_allclose(buf.cpu().numpy(), ref_buf.cpu().numpy(), rtol=1e-12, atol=1e-9)
    assert int(buf[0]) == 64 and int(buf[3]) == n_eps and int(buf[2]) == n_eps - 64
    env.counters.copy_(state[0]); env.ep_log.copy_(state[1])       # put the log back for the checks below
    stats = odist.gather_env_episodes(env, steps)       # drains the device log through the rollout-boundary collective path
    rets, lens, ts, dropped = stats
    assert dropped == 0 and ts == steps * n and stats.sums[0] == n_eps and rets.numel() == n_eps
    dev = sorted(zip(lens.tolist(), rets.tolist()))
    exp = sorted(exp_g)
    np.testing.assert_array_equal([d[0] for d in dev], [e[0] for e in exp])                     # lengths: exact multiset
    np.testing.assert_allclose(sorted(r for _, r in dev), sorted(r for _, r in exp), atol=2e-4)   # returns: f32 running sums
    assert abs(stats.mean_return - np.mean([r for _, r in exp])) < 1e-4
    assert int(env.counters[_abi.CNT_EPISODES].item()) == 0                                     # drained
    # against the oracle: its own log has the same number of episodes up to the robots whose done flags flipped,
    # and episodes with an identical done history agree on the return (per-step reward tolerance 3e-3)
    o_n = int(orc.counters[_abi.CNT_EPISODES])
    assert abs(o_n - n_eps) <= max(4, n // 16), (o_n, n_eps)
    o_log = orc.ep_log[:o_n]
    np.testing.assert_array_equal(np.sort(o_log[:, 1]), np.sort([e[0] for e in exp_o]))
    assert len(pairs) > n // 2
    # 20 env steps of contact dynamics amplify float32 rounding, so returns are compared statistically (as the 10-step rollout test does)
    diff = np.array([abs(p[1] - p[2]) for p in pairs])
    assert np.median(diff) < 0.02 and np.percentile(diff, 95) < 0.3, (np.median(diff), np.percentile(diff, 95), diff.max())
    assert abs(np.mean([p[1] for p in pairs]) - np.mean([p[2] for p in pairs])) < 0.02
    env.close(); orc.close()


def test_soak_2000_steps_4096_robots():
    import torch
    n, steps = 4096, 2000
    env, orc = make_pair("laikago", n=n, randomizer=True, auto_reset=True, mode="train", seed=5)
    orc.close()
    obs = env.reset()
    jom = torch.tensor(env.models[0]["joint_of_motor"], dtype=torch.long, device=env.device)
    mdir = torch.tensor(env.models[0]["motor_dir"], dtype=torch.float32, device=env.device)
    off = torch.tensor(env.models[0]["motor_offset"], dtype=torch.float32, device=env.device)
    init = torch.tensor(env.models[0]["init_motor_angles"], dtype=torch.float32, device=env.device)
    gen = torch.Generator(device=env.device); gen.manual_seed(1)
    bad = torch.zeros((), dtype=torch.int64, device=env.device)
    total_done = torch.zeros((), dtype=torch.int64, device=env.device)
    rsum = torch.zeros((), dtype=torch.float64, device=env.device)
    low = torch.tensor(env.observation_space.low, device=env.device)
    high = torch.tensor(env.observation_space.high, device=env.device)
    logged = 0
    for k in range(steps):
        tar = obs[:, 84 + 7:84 + 19].index_select(1, jom)
        a = ((tar - off) * mdir - init + torch.randn(n, 12, generator=gen, device=env.device) * 0.125).clamp(-2 * np.pi, 2 * np.pi)
        obs, rew, done, _ = env.step(a)
        bad += (~torch.isfinite(obs)).sum() + (~torch.isfinite(rew)).sum() + (rew < 0).sum() + (rew > 1.0 + 1e-6).sum()
        # motor angles and target frames stay inside the observation space the policy zips were trained with (F6).  Not checked:
        # LastAction (its +-1 bound ignores the INIT_MOTOR_ANGLES offset) and the joint targets of warm-up
        # episodes (the default pose lies outside the clip's joint range; same in the reference, imitation_task.py:985-1009)
        tol = 1e-4
        out = (obs < low - tol) | (obs > high + tol)
        out[:, 12:48] = False
        warm = env.field_int("WARMUP")[:, 0] != 0
        out[:, 84:] &= ~warm[:, None]
        bad += out.sum()
        total_done += done.sum()
        rsum += rew.double().sum()
        if k % 256 == 255:
            r, l, dropped = env.episode_log(with_dropped=True)
            assert dropped == 0 and (l >= 1).all() and (l <= 600).all() and torch.isfinite(r).all()
            logged += int(r.numel())
    torch.cuda.synchronize()
    assert int(bad) == 0
    logged += int(env.episode_log()[0].numel())
    cnt = env.counters.cpu().numpy()
    assert cnt[_abi.CNT_TOTAL_TIMESTEPS] == steps * n
    assert cnt[_abi.CNT_TOTAL_STEP_COUNT] == int(total_done) == logged
    assert cnt[_abi.CNT_TICKET] == 0 and cnt[_abi.CNT_DONE_ACCUM] == 0
    assert int(env.field_int("DONE_REASON").bitwise_and(_abi.DONE_NAN).sum()) == 0
    # open-loop reference-pose actions with N(0, 0.125^2) noise under the randomiser: measured mean reward 0.25 per step
    assert 0.15 < float(rsum) / (steps * n) < 0.6
    # curriculum: 4096 robots x 2000 steps = 8.2e6 robot steps is far below 3e7 resets, time limit still ~20
    assert 20 <= int(env.field_int("MAX_EP_STEPS").max()) <= 21
    env.close()


@pytest.mark.parametrize("robot", ["laikago", "mini_cheetah"])
def test_done_disagreements_are_threshold_cases(robot):
    """float32 device vs float64 oracle: a robot may end its episode one step apart only when the oracle itself sees it
    within a small margin of a termination threshold (fall-proxy clearance, root distance, root rotation)."""
    import ctypes as C
    import torch
    n = 1024
    env, orc = make_pair(robot, n=n, randomizer=True, mode="train", seed=41)
    orc.L.orc_set_margins_out.argtypes = [C.c_void_p, ol.dp]
    margins = np.zeros((n, 3))
    orc.L.orc_set_margins_out(orc.h, ol.P(margins))
    env.reset(); orc.reset()
    rng = np.random.RandomState(7)
    # spread the batch: tilted, dropped, pushed robots so that a good share of them terminates during these steps
    st = gpu_state64(env)
    lay = env.layout
    st[:, lay.sl("POS")][:, 2] += rng.uniform(-0.05, 0.10, n)
    st[:, lay.sl("LINVEL")] += rng.randn(n, 3) * 1.5
    st[:, lay.sl("ANGVEL")] += rng.randn(n, 3) * 6.0
    st = statemod.to_float64(lay, statemod.from_float64(lay, st))
    env.state.copy_(torch.from_numpy(statemod.from_float64(lay, st)).to(env.device)); orc.state[:] = st
    n_done, n_dis = 0, 0
    for k in range(12):
        a = rng.uniform(-0.4, 0.4, (n, 12)).astype(np.float32)
        og, rg, dg, _ = env.step(torch.from_numpy(a).to(env.device))
        oo, ro, do = orc.step(a.astype(np.float64))
        dg = dg.cpu().numpy().astype(bool)
        dis = dg != do
        n_done += int(do.sum()); n_dis += int(dis.sum())
        # every disagreement is a near-threshold case for the oracle: 3 mm / 3 mm / 5 mrad
        near = (np.abs(margins[:, 0]) < 3e-3) | (np.abs(margins[:, 1]) < 3e-3) | (np.abs(margins[:, 2]) < 5e-3)
        assert near[dis].all(), (k, np.nonzero(dis & ~near)[0][:8], margins[dis & ~near][:8])
        orc.state[:] = gpu_state64(env)         # next step starts from identical states again
    assert n_done > n // 20, n_done               # the sample really contains terminations
    assert n_dis <= max(3, n_done // 20), (n_dis, n_done)
    env.close(); orc.close()


def test_model_and_config_preconditions_of_the_kernel_are_checked():
    """The step kernel integrates a joint angle as q += jdir dt v and stores rotations as series in the turn of one sub-step: the
    host entry points refuse a model with |motor_dir| != 1 (the reference's directions are +-1) and keep working otherwise."""
    import ctypes as C
    from openroborl_amd import _lib, robots
    from openroborl_amd.env import VecQuadrupedEnv
    env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=8, seed=1)
    t = next(i for i, mod in enumerate(env.models) if mod is not None)
    m = robots.to_struct(env.models[t])
    good = m.motor_dir[3]
    m.motor_dir[3] = 0.5
    rc = env.L.orr_set_model(env.h, t, C.byref(m))
    assert rc < 0 and b"motor_dir" in env.L.orr_last_error()
    m.motor_dir[3] = good
    assert env.L.orr_set_model(env.h, t, C.byref(m)) == 0
    env.close()
