"""agents/algorithms/rl/sac/storage.py holds the same ReplayBuffer as ddpg/storage.py: one implementation here."""
from ..ddpg.storage import ReplayBuffer  # noqa: F401
