from .ppo import ppo
from .dagger import dagger
from .bc import bc
