"""Back-compat alias for synthetic/inputs.py."""
from synthetic.inputs import *  # noqa: F401,F403
