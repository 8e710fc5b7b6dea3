from partmanip_amd.algo_utils import ActorCritic, AdvScaling, Normalization, RolloutStorage, FusedAdam  # noqa: F401
