"""Back-compat alias: the seeded parameter factory is data, not oracle logic, and lives in synthetic/weights.py."""
from synthetic.weights import *  # noqa: F401,F403
from synthetic.weights import _Gen  # noqa: F401
