"""openroborl_amd: MI355X-native vectorised quadruped motion-imitation environment.

Drop-in for the env.reset()/env.step() hot path of Derek-TH-Wang/OpenRoboRL:
hand-written HIP kernels (gfx950) behind a C-ABI library, a Python host mirroring the reference's
Gym-style surface and YAML configuration.
"""
__version__ = "0.1.0"
