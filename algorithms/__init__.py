"""Drop-in import surface of the reference: `from algorithms import ppo, dagger, bc`
(train.py:1, algorithms/__init__.py:1-3) and `from algorithms.algo_utils import ...` (ppo.py:4)
resolve to the MI355X-native implementations in `partmanip_amd`."""
from partmanip_amd.algorithms import ppo, dagger, bc  # noqa: F401
