from .actor_critic import ActorCritic
from .RMS import AdvScaling, Normalization
from .storage import RolloutStorage
from .optim import FusedAdam
