"""PPO learner (PyTorch-ROCm): `ppo.train.train(environment, ...)`, `ppo.train.ppo_config()`."""
from . import networks, train  # noqa: F401
