"""Time the fused actor / critic forward pass (csrc/orr_policy.hip) against the plain PyTorch path it replaces.

usage (GPU box):  python tools/bench_policy.py [robots]
HISTORY.md section 3a quotes these numbers; `rocprofv3 --kernel-trace --stats -- python3 tools/bench_policy.py` gives the
per-kernel view stored in profiles/r01_policy_kernel_stats.csv.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from openroborl_amd import ppo  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
obs = torch.randn(n, 160, device=dev)
noise = torch.randn(n, 12, device=dev)


def timeit(f, reps=200):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


torch_model = ppo.ActorCritic(dev, seed=0)
fused_model = ppo.ActorCritic(dev, seed=0).enable_fused()
t_torch = timeit(lambda: torch_model.act(obs, noise=noise))
t_fused = timeit(lambda: fused_model.act(obs, noise=noise))
t_pack = timeit(lambda: fused_model.fused.refresh())
flops = 2.0 * n * 2 * (160 * 512 + 512 * 256) + 2.0 * n * 256 * 13
print("robots %d: torch act() %.1f us, fused act() %.1f us (%.1f TFLOP/s f32), weight re-pack %.1f us"
      % (n, t_torch, t_fused, flops / t_fused * 1e-6, t_pack))
