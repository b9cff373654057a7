"""Host side of the MI355X-native quadruped imitation environment.

VecQuadrupedEnv  tensor-native: reset(mask=None) -> obs[N,160]; step(actions[N,12]) ->
                 (obs[N,160], reward[N], done[N], info) on torch.float32 ROCm tensors, no host sync,
                 per-robot masked auto-reset.  Mirrors the attribute surface the reference's agent
                 reads: num_robot, observation_space, action_space, env_step_counter, seed(), close()
                 (wrapper_env.py:55-56,58-107; quadruped_gym_env.py:59-61,149-152).
LegacyListEnv    the reference's exact list-of-numpy protocol (wrapper_env.py:58-107) including the
                 "caller resets the whole env when any robot is done" flow of
                 agents/imitation_runners.py:185-205, so a stable-baselines-style loop runs unchanged.

PyTorch only owns the device buffers and the stream; all per-robot work happens inside the HIP
kernels behind the C-ABI (include/openroborl_hip.h).
"""
import ctypes as C
import math

import numpy as np

from . import _abi, _lib, config as cfgmod, motion, robots, state as statemod


class Box(object):
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency): low / high / shape / dtype."""

    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def __repr__(self):
        return "Box(%s, %s)" % (self.shape, self.dtype)


def proprio_bounds():
    """Sensor bounds in flattened (sorted-name) order: IMU x3 | LastAction x3 | MotorAngle x3
    (robot_sensors.py:52-71,111-133; environment_sensors.py:37-38; sensor_wrappers.py:108-110)."""
    imu = np.array([2 * np.pi, 2 * np.pi, 2000 * np.pi, 2000 * np.pi])
    high = np.concatenate([np.tile(imu, 3), np.ones(36), np.pi * np.ones(36)])
    return -high, high


def target_bounds(clips):
    """ImitationTask.get_target_obs_bounds (imitation_task.py:303-335)."""
    low = np.inf * np.ones(_abi.POSE_DIM)
    high = -np.inf * np.ones(_abi.POSE_DIM)
    for c in clips:
        lo, hi = c.joint_bounds()
        low = np.minimum(low, lo)
        high = np.maximum(high, hi)
    low[0:3], high[0:3] = -2.0, 2.0
    low[3:7], high[3:7] = -1.0, 1.0
    return np.tile(low, 4), np.tile(high, 4)


def observation_space(clips):
    """WrapperEnv._build_observation_space (wrapper_env.py:127-145)."""
    pl, ph = proprio_bounds()
    tl, th = target_bounds(clips)
    return Box(np.concatenate([pl, tl]), np.concatenate([ph, th]), dtype=np.float32)


def action_space():
    """minitaur.py:145-148."""
    return Box(np.array([-2 * math.pi] * 12), np.array([2 * math.pi] * 12), dtype=np.float32)


class VecQuadrupedEnv(object):
    """N independent quadrupeds on one GPU; four robots per wavefront, 16 lanes each (see csrc/orr_kernels.hip, orr_physics.h)."""

    def __init__(self, task_name=None, training_yaml=None, sim_yaml=None, device="cuda", num_robot=None, seed=None,
                 robot=None, motion_file=None, mode=None, enable_randomizer=None, auto_reset=True, num_procs=1,
                 robot_index_offset=0, legacy_grid=False, mixed_robots=None, ep_log_capacity=65536, config_overrides=None,
                 model_overrides=None):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("VecQuadrupedEnv needs a ROCm GPU (the HIP path has no CPU fallback)")
        self.L = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("device must be a cuda (ROCm) device")
        if self.device.index is not None:
            torch.cuda.set_device(self.device)   # the C-ABI library allocates / launches on the current HIP device
        params = {}
        if task_name is not None:
            params = cfgmod.load_training_params(task_name, training_yaml)
        sim = cfgmod.load_sim_params(sim_yaml)
        robot = robot or params.get("robot", "laikago")
        if robot not in robots.ROBOTS:
            raise ValueError("wrong robot select")                       # minitaur.py:97
        mode = mode or params.get("mode", "train")
        if enable_randomizer is None:
            enable_randomizer = bool(params.get("enable_env_randomizer", True)) and mode == "train"   # run.py:205-206
        num_robot = int(num_robot if num_robot is not None else params.get("num_robot", 1))
        seed = int(seed if seed is not None else params.get("seed", 0))
        motion_file = motion_file or params.get("motion_file")
        if motion_file is None:
            raise ValueError("no input robot or task")                   # quadruped_gym_env.py:50-51
        motion_files = list(motion_file) if isinstance(motion_file, (list, tuple)) else [motion_file]
        self.num_robot = num_robot
        self.mode = mode
        self.cfg = cfgmod.make_config(num_robot, sim_params=sim, mode=mode, enable_randomizer=enable_randomizer, seed=seed,
                                      num_procs=num_procs, auto_reset=auto_reset, legacy_grid=legacy_grid)
        for k, v in (config_overrides or {}).items():     # e.g. a shorter episode-length curriculum (WrapperEnv arguments, run.py:54-75)
            if not hasattr(self.cfg, k):
                raise ValueError("unknown orr_config field %r" % (k,))
            setattr(self.cfg, k, v)
        # robots: homogeneous batch, or interleaved heterogeneous batch (BASELINE config 5)
        if mixed_robots:
            self.robot_names = list(mixed_robots)
            robot_type = np.array([robots.ROBOT_TYPE_ID[self.robot_names[i % len(self.robot_names)]]
                                   for i in range(num_robot)], dtype=np.int32)
        else:
            self.robot_names = [robot]
            robot_type = np.full(num_robot, robots.ROBOT_TYPE_ID[robot], dtype=np.int32)
        self.models = [None] * _abi.MAX_ROBOT_TYPES
        # model_overrides = {robot name: {table entry: value}}: experiments on the hand-authored (parity-unpinned) entries of robots.py;
        # the entry "_build" = {keyword: value} replaces arguments of the table builder (link masses, COMs, hip position, ...)
        model_overrides = {k: dict(v) for k, v in (model_overrides or {}).items()}
        for name in set(self.robot_names):
            self.models[robots.ROBOT_TYPE_ID[name]] = robots.ROBOTS[name](**model_overrides.get(name, {}).pop("_build", {}))
        for name, over in model_overrides.items():
            m = self.models[robots.ROBOT_TYPE_ID[name]] if name in robots.ROBOT_TYPE_ID else None
            if m is None:
                raise ValueError("model_overrides names robot %r, which is not in this batch" % (name,))
            for k, v in over.items():
                if k not in m:
                    raise ValueError("unknown model table entry %r" % (k,))
                m[k] = np.asarray(v, dtype=np.asarray(m[k]).dtype).reshape(np.shape(m[k])) if np.ndim(m[k]) else type(m[k])(v)
        # clips: one per robot type in a mixed batch, else the task's clip
        self.clips = [motion.MotionClip(f) for f in motion_files]
        if mixed_robots:
            if len(self.clips) != len(self.robot_names):
                raise ValueError("mixed batch needs one motion file per robot name")
            type_to_clip = {robots.ROBOT_TYPE_ID[n]: i for i, n in enumerate(self.robot_names)}
            clip_id = np.array([type_to_clip[t] for t in robot_type], dtype=np.int32)
        else:
            clip_id = np.zeros(num_robot, dtype=np.int32)
        self.robot_type = robot_type
        self.clip_id = clip_id

        h = C.c_void_p()
        _lib.check(self.L.orr_create(C.byref(self.cfg), C.byref(h)), self.L)
        self.h = h
        for t, m in enumerate(self.models):
            if m is not None:
                _lib.check(self.L.orr_set_model(self.h, t, C.byref(robots.to_struct(m))), self.L)
        self._clip_tensors = []
        for i, c in enumerate(self.clips):
            fr = torch.tensor(c.frames, dtype=torch.float32, device=self.device).contiguous()
            fv = torch.tensor(c.frame_vels, dtype=torch.float32, device=self.device).contiguous()
            self._clip_tensors.append((fr, fv))
            cd = (C.c_float * 4)(*[float(x) for x in c.cycle_delta])
            _lib.check(self.L.orr_set_motion(self.h, i, fr.data_ptr(), fv.data_ptr(), c.num_frames,
                                             float(c.frame_duration), c.flags, cd), self.L)
        self.layout = statemod.Layout(self.L, "orr")
        idx = np.arange(num_robot, dtype=np.int32) + int(robot_index_offset)
        st = statemod.default_state(self.layout, num_robot, self.models, robot_type, clip_id, idx,
                                    legacy_grid=legacy_grid, ctrl_latency=cfgmod.CTRL_LATENCY,
                                    max_ep_steps=self.cfg.ep_len_end)
        self.state = torch.from_numpy(st).to(self.device).contiguous()
        self.counters = torch.zeros(_abi.NUM_COUNTERS, dtype=torch.int64, device=self.device)
        self.ep_log = torch.zeros((max(int(ep_log_capacity), 1), 2), dtype=torch.float32, device=self.device)
        _lib.check(self.L.orr_bind(self.h, self.state.data_ptr(), self.counters.data_ptr(), self.ep_log.data_ptr(),
                                   int(ep_log_capacity)), self.L)
        # the three outputs of a step are views into ONE device buffer [obs N x 160 f32 | reward N f32 | done N u8], so that a host-side
        # consumer (LegacyListEnv) fetches them with a single copy
        nb_obs, nb_rew = num_robot * _abi.OBS_DIM * 4, num_robot * 4
        self._out = torch.zeros(nb_obs + nb_rew + num_robot, dtype=torch.uint8, device=self.device)
        self.obs = self._out[:nb_obs].view(torch.float32).view(num_robot, _abi.OBS_DIM)
        self.reward = self._out[nb_obs:nb_obs + nb_rew].view(torch.float32)
        self.done = self._out[nb_obs + nb_rew:]
        self.observation_space = observation_space(self.clips)
        self.action_space = action_space()
        self._env_step_counter = 0
        self._closed = False
        self.launch_params_generation = 0     # bumped whenever something a launch takes by value changes (seed()): see there

    # ---- reference attribute surface -------------------------------------------------------
    @property
    def env_step_counter(self):
        """quadruped_gym_env.py:336-337 (env-global in the reference; here: steps since the last full reset)."""
        return self._env_step_counter

    @property
    def env_time_step(self):
        return self.cfg.action_repeat * self.cfg.sim_dt

    def seed(self, seed=None):
        """quadruped_gym_env.py:59-61.  The RNG is counter-based (Philox keyed by seed, global robot index, episode index):
        a new seed takes effect for every episode that starts after this call.  Returns [seed] like gym."""
        if seed is not None and (int(seed) & 0xFFFFFFFFFFFFFFFF) != int(self.cfg.seed):
            self.cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
            _lib.check(self.L.orr_set_seed(self.h, self.cfg.seed), self.L)
            # orr_step / orr_reset pass the handle's configuration (seed included) BY VALUE as a kernel argument: a hipGraph captured
            # before this call replays the old seed.  Holders of such graphs (rollout.GraphRollout) compare this counter and re-capture
            self.launch_params_generation += 1
        return [int(self.cfg.seed)]

    def close(self):
        if not self._closed and self.h:
            self.torch.cuda.synchronize(self.device)
            self.L.orr_destroy(self.h)
            self.h = None
            self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- hot path ------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self, mask=None):
        """WrapperEnv.reset (wrapper_env.py:87-107).  mask: optional bool/uint8 tensor [N]; rows of robots
        that are not reset keep their previous observation."""
        mp = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=self.torch.uint8).contiguous()
            mp = mask.data_ptr()
        else:
            self._env_step_counter = 0
        _lib.check(self.L.orr_reset(self.h, mp, self.obs.data_ptr(), self._stream()), self.L)
        return self.obs

    def step(self, actions):
        """WrapperEnv.step (wrapper_env.py:58-85).  actions: float32 [N,12] on the env device (policy
        outputs, clipped to +-2 pi by the caller as in imitation_runners.py:140-143)."""
        t = self.torch
        if actions.dtype != t.float32 or actions.device != self.obs.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=t.float32).contiguous()
        if tuple(actions.shape) != (self.num_robot, _abi.NUM_MOTORS):
            raise ValueError("actions must have shape (%d, %d)" % (self.num_robot, _abi.NUM_MOTORS))
        _lib.check(self.L.orr_step(self.h, actions.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(),
                                   self.done.data_ptr(), self._stream()), self.L)
        self._env_step_counter += 1
        return self.obs, self.reward, self.done, {}

    def step_into(self, actions, obs_out, reward_out, done_out):
        """step() writing its three outputs into the caller's tensors (rows of a rollout buffer) instead of env.obs / env.reward /
        env.done: contiguous float32 [N,160] (16-byte aligned), float32 [N], uint8 [N] on the env device.  Nothing here depends on
        host state that changes from call to call, so a sequence of these calls can be captured into a hipGraph and replayed - with ONE
        restriction: the launch takes the handle's configuration (the seed included) by value, so a graph captured before env.seed(new)
        replays the old seed; `launch_params_generation` counts such changes and rollout.GraphRollout re-captures when it moves."""
        t = self.torch
        n = self.num_robot
        for x, shape, dt in ((actions, (n, _abi.NUM_MOTORS), t.float32), (obs_out, (n, _abi.OBS_DIM), t.float32), (reward_out, (n,), t.float32),
                             (done_out, (n,), t.uint8)):
            if x.dtype != dt or x.device != self.obs.device or not x.is_contiguous() or tuple(x.shape) != shape:
                raise ValueError("step_into: expected a contiguous %s tensor of shape %s on %s" % (dt, shape, self.device))
        _lib.check(self.L.orr_step(self.h, actions.data_ptr(), obs_out.data_ptr(), reward_out.data_ptr(), done_out.data_ptr(), self._stream()), self.L)
        self._env_step_counter += 1

    def time_steps(self, actions, num_steps):
        """Bench helper: num_steps back-to-back launches timed with hipEvents on the launch stream (ms)."""
        ms = C.c_float()
        _lib.check(self.L.orr_time_steps(self.h, actions.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(),
                                         self.done.data_ptr(), self._stream(), int(num_steps), C.byref(ms)), self.L)
        self._env_step_counter += int(num_steps)
        return float(ms.value)

    def stress_actions(self, obs, noise, out):
        """Bench helper: the policy-free stress actions of SURVEY.md section 8d (i) for any mix of robot types, one launch:
        out[i] = clip((first target frame's joints of obs[i], joint -> motor space of robot i) - INIT_MOTOR_ANGLES + noise[i], +-2 pi)."""
        t = self.torch
        for x, cols in ((obs, _abi.OBS_DIM), (noise, _abi.NUM_MOTORS), (out, _abi.NUM_MOTORS)):
            if x.dtype != t.float32 or x.device != self.obs.device or not x.is_contiguous() or tuple(x.shape) != (self.num_robot, cols):
                raise ValueError("stress_actions: contiguous float32 [%d, %d] on the env device expected" % (self.num_robot, cols))
        _lib.check(self.L.orr_stress_actions(self.h, obs.data_ptr(), noise.data_ptr(), out.data_ptr(), self._stream()), self.L)
        return out

    def replay_reset(self, uniforms):
        """Parity entry: reset of all robots with the given draws ([N,28] in [0,1)) instead of the Philox stream."""
        _lib.check(self.L.orr_debug_replay_reset(self.h, uniforms.data_ptr(), self.obs.data_ptr(), self._stream()), self.L)
        self._env_step_counter = 0
        return self.obs

    def replay_step(self, actions, traj, eff, fall, tau_out):
        """Parity entry: one env step with the physics sub-steps replaced by recorded states (include/openroborl_hip.h)."""
        _lib.check(self.L.orr_debug_replay_step(self.h, actions.data_ptr(), traj.data_ptr(), eff.data_ptr(), fall.data_ptr(), self.obs.data_ptr(),
                                                self.reward.data_ptr(), self.done.data_ptr(), tau_out.data_ptr(), self._stream()), self.L)
        self._env_step_counter += 1
        return self.obs, self.reward, self.done

    def debug_physics(self, torques, nsub):
        fall = self.torch.zeros(self.num_robot, dtype=self.torch.uint8, device=self.device)
        _lib.check(self.L.orr_debug_physics(self.h, torques.data_ptr(), fall.data_ptr(), int(nsub), self._stream()), self.L)
        return fall

    # ---- state access (checkpoint / parity injection) -------------------------------------------
    def field(self, name):
        """View of a float field of the state tensor: [N, size]."""
        return self.state[:, self.layout.sl(name)]

    def field_int(self, name):
        return self.state.view(self.torch.int32)[:, self.layout.sl(name)]

    def state_dict(self):
        return {"state": self.state.clone(), "counters": self.counters.clone(), "env_step_counter": self._env_step_counter}

    def load_state_dict(self, d):
        self.state.copy_(d["state"])
        self.counters.copy_(d["counters"])
        self._env_step_counter = int(d["env_step_counter"])

    def stats(self):
        """Diagnostic counters (metrics): totals + histogram of the last done reasons.  Syncs."""
        c = self.counters.cpu().numpy()
        reasons = self.field_int("DONE_REASON")[:, 0].cpu().numpy()
        names = (("contact_fall", _abi.DONE_CONTACT_FALL), ("root_pos", _abi.DONE_ROOT_POS), ("root_rot", _abi.DONE_ROOT_ROT),
                 ("time_limit", _abi.DONE_TIME_LIMIT), ("non_finite", _abi.DONE_NAN), ("motion_over", _abi.DONE_MOTION_OVER))
        return {"total_timesteps": int(c[_abi.CNT_TOTAL_TIMESTEPS]), "curriculum_counter": int(c[_abi.CNT_TOTAL_STEP_COUNT]),
                "episodes_logged": int(c[_abi.CNT_EPISODES]), "episodes_dropped": int(c[_abi.CNT_EPLOG_DROPPED]),
                "last_done_reason": {k: int(((reasons & bit) != 0).sum()) for k, bit in names},
                "max_episode_steps": int(self.field_int("MAX_EP_STEPS").max().item())}

    def episode_log_device(self):
        """(log[K,2] snapshot, count, dropped) of the episodes finished since the last call, all on the device and
        without a host sync (count / dropped are 0-d int64 tensors; rows >= count are stale); clears the log."""
        t = self.torch
        log = self.ep_log.clone()
        cnt = self.counters[_abi.CNT_EPISODES].clone()
        dropped = self.counters[_abi.CNT_EPLOG_DROPPED].clone()
        self.counters[_abi.CNT_EPISODES:_abi.CNT_EPLOG_DROPPED + 1] = 0
        return log, t.clamp(cnt, max=log.shape[0]), dropped

    def episode_stats_packed(self, total_timesteps, capacity):
        """The rank's payload of the rollout-boundary all-gather (dist.py layout, float64 [6 + 2 capacity]) in one launch; clears the log."""
        out = self.torch.empty(6 + 2 * int(capacity), dtype=self.torch.float64, device=self.device)
        _lib.check(self.L.orr_episode_stats(self.h, float(total_timesteps), int(capacity), out.data_ptr(), self._stream()), self.L)
        return out

    def episode_log(self, with_dropped=False):
        """(returns[K], lengths[K]) of the episodes finished since the last call (+ the number of episodes that did not fit
        the device log when with_dropped); clears the log.  Syncs."""
        log, cnt, dropped = self.episode_log_device()
        k = int(cnt.item())
        if with_dropped:
            return log[:k, 0], log[:k, 1], int(dropped.item())
        return log[:k, 0], log[:k, 1]


class LegacyListEnv(object):
    """The reference's list-of-numpy env protocol on top of a VecQuadrupedEnv built with auto_reset=False.

    reset() -> list[N] of float64 arrays (160,)
    step(list[N] of arrays (12,)) -> (obs list, reward list[float], done list[bool], info list[dict])
    (wrapper_env.py:58-107).  As in the reference, the caller resets the WHOLE env when any (train,
    imitation_runners.py:185-205) or all (test, run.py:169) robots are done, `info[i]["terminated"]` aliases
    the done list, the time limit uses the env-global step counter, and the caller's action arrays get
    INIT_MOTOR_ANGLES added in place (minitaur.py:281).
    """

    def __init__(self, env, mutate_actions=True):
        if env.cfg.flags & _abi.FLAG_AUTO_RESET:
            raise ValueError("LegacyListEnv needs a VecQuadrupedEnv created with auto_reset=False")
        self._env = env
        self._mutate = mutate_actions
        self.num_robot = env.num_robot
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self._init_angles = [np.asarray(env.models[t]["init_motor_angles"], dtype=np.float64) for t in env.robot_type]
        # host <-> device staging in pinned memory: one asynchronous upload (actions), one asynchronous download (obs | reward | done)
        # and ONE synchronisation per step, instead of four blocking pageable copies
        t = env.torch
        n = env.num_robot
        self._act_host = t.empty((n, _abi.NUM_MOTORS), dtype=t.float32).pin_memory()
        self._act_dev = t.empty((n, _abi.NUM_MOTORS), dtype=t.float32, device=env.device)
        self._out_host = t.empty(env._out.shape, dtype=t.uint8).pin_memory()
        nb_obs, nb_rew = n * _abi.OBS_DIM * 4, n * 4
        out_np = self._out_host.numpy()
        self._obs_np = out_np[:nb_obs].view(np.float32).reshape(n, _abi.OBS_DIM)
        self._rew_np = out_np[nb_obs:nb_obs + nb_rew].view(np.float32)
        self._done_np = out_np[nb_obs + nb_rew:]

    def _fetch(self):
        """obs | reward | done of the last reset / step -> pinned host buffer (one copy, one sync)."""
        t = self._env.torch
        self._out_host.copy_(self._env._out, non_blocking=True)
        t.cuda.current_stream(self._env.device).synchronize()

    def __getattr__(self, attr):           # wrapper_env.py:55-56
        return getattr(self._env, attr)

    def reset(self):
        self._env.reset()
        self._fetch()
        return list(self._obs_np.astype(np.float64))

    def step(self, action):
        t = self._env.torch
        try:
            a = np.asarray(action, dtype=np.float32)              # list of equal-length arrays: one conversion
        except (ValueError, TypeError):
            a = np.stack([np.asarray(action[i], dtype=np.float32) for i in range(self.num_robot)])
        if a.shape != (self.num_robot, _abi.NUM_MOTORS):
            a = np.stack([np.asarray(action[i], dtype=np.float32).reshape(_abi.NUM_MOTORS) for i in range(self.num_robot)])
        self._act_host.numpy()[...] = a
        self._act_dev.copy_(self._act_host, non_blocking=True)
        self._env.step(self._act_dev)
        if self._mutate:                   # minitaur.py:281 adds INIT_MOTOR_ANGLES to the caller's arrays in place (done while the GPU works)
            init = self._init_angles
            for i in range(self.num_robot):
                ai = action[i]
                if isinstance(ai, np.ndarray):
                    ai += init[i]
        self._fetch()
        obs = self._obs_np.astype(np.float64)
        rew_list = self._rew_np.astype(np.float64).tolist()
        done_np = self._done_np.astype(bool)
        ndone = int(done_np.sum())
        if ndone > 0:
            # wrapper_env.py:82-83: the curriculum counter advances by num_robot per step in which ANY robot finished
            # (the kernel already added one per finished robot)
            self._env.counters[_abi.CNT_TOTAL_STEP_COUNT] += self.num_robot - ndone
        done_list = done_np.tolist()
        info = [{"terminated": done_list} for _ in range(self.num_robot)]
        return list(obs), rew_list, done_list, info


def build_env(task_name, num_robot=None, mode=None, enable_randomizer=None, legacy=False, **kw):
    """Counterpart of run.py:49-97 build_env for the two imitation tasks."""
    if task_name not in cfgmod.TASKS:
        raise ValueError("unknown task %r" % (task_name,))
    env = VecQuadrupedEnv(task_name=task_name, num_robot=num_robot, mode=mode, enable_randomizer=enable_randomizer,
                          auto_reset=not legacy, **kw)
    return LegacyListEnv(env) if legacy else env
