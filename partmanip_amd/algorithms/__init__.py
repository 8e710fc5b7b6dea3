from .ppo import ppo
from .dagger import dagger
