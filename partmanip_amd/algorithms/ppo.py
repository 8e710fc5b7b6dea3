"""PPO runner with the reference's interface (algorithms/ppo.py: `ppo(vec_env, cfg, logger)`,
`run()`, `update(it)`, `eval()`, `save(it)`, `resume(path)`; same cfg keys, same public
attributes, same checkpoint dict) whose learner is the MI355X path:

  compute_returns  -> HIP GAE scan                                   (storage.py:96-114)
  update           -> per mini-batch: HIP forward, fused loss fwd+bwd, HIP backward,
                      [one RCCL all-reduce], fused clip+Adam          (ppo.py:307-411)

Differences that are deliberate and semantics-preserving:
  * no host sync inside `update`: the KL early-stop `continue` (ppo.py:337-338) is a device
    predicate consumed by the optimiser kernel; the running sums behind the `Train/*` scalars
    are accumulated on the device and read back once at the end;
  * each pass runs only the network it updates (the reference's `update_act_cri` also runs the
    other one and discards the result, actor_critic.py:72,80);
  * sequential mini-batches are zero-copy slices of the (T*N, D) views.
"""
import os
import sys
import time
from copy import deepcopy
from os.path import join as pjoin

import numpy as np
import torch

from ..algo_utils import RolloutStorage, ActorCritic, Normalization, FusedAdam
from ..algo_utils.network import chains_forward, chains_backward
from .. import ops, dist as pdist

try:                                        # simulator-side helper of the reference (utils/img2video.py)
    from utils import path2video            # noqa: F401
except Exception:                           # not present when running against the feeder
    def path2video(*a, **k):
        return None


class ppo:
    def __init__(self, vec_env, cfg, logger):
        self.vec_env = vec_env
        self.num_envs = cfg['num_envs']
        self.obs_mode = cfg['obs_mode']
        self.num_obs = vec_env.num_obs[self.obs_mode]
        self.num_actions = vec_env.num_actions
        self.max_episode_length = vec_env.max_episode_length
        self.default_succ_value = cfg['succ_value']

        self.model_cfg = cfg['model']
        self.max_iter = cfg['max_iterations']
        self.n_steps = cfg['n_steps']
        self.n_updates = cfg['n_updates']
        self.num_mini_batches = cfg['n_minibatches']
        self.device = cfg['device']

        self.eval_round = cfg['eval_round']
        self.eval_freq = cfg['eval_frequence']
        self.save_freq = cfg['save_frequence']
        self.test_only = cfg['test_only']
        self.save_pose = cfg['save_pose']
        self.save_video = cfg['save_video']
        self.save_ckpt_dir = logger.save_ckpt_dir

        self.lr_schedule = cfg['lr_schedule']
        self.lr = cfg['lr']
        self.desired_kl = cfg['desired_kl']
        assert self.desired_kl > 0
        if self.lr_schedule not in ('fixed', 'linear_decay', 'step_decay'):
            raise NotImplementedError

        self.epsilon_clip = cfg['epsilon_clip']
        self.gamma = cfg['gamma']
        self.lam = cfg['lam']

        self.tricks_keys = ['mini_adv_norm', 'whole_adv_norm', 'use_state_norm', 'use_clipped_value_loss',
                            'use_grad_clip']
        self.tricks = {k: cfg['tricks'][k] for k in self.tricks_keys}
        if self.tricks['use_grad_clip']:
            self.max_grad_norm = cfg['tricks']['max_grad_norm']
        if self.tricks['use_state_norm']:
            self.state_norm = Normalization(shape=self.num_obs, device=self.device)
            self.update_RMS = True

        self.actor_critic = ActorCritic(self.num_obs, self.num_actions, self.model_cfg).to(self.device)
        self.storage = RolloutStorage(self.num_envs, self.n_steps, self.num_obs, self.num_actions, self.device,
                                      self.default_succ_value, self.tricks['whole_adv_norm'], cfg['sampler'])
        ac = self.actor_critic
        f = ac.flat()
        n_a, n_c = f['n_actor'], f['n_critic']
        # two optimisers over the two flat buffers, param groups as ppo.py:73-74
        self.optimizer_actor = FusedAdam(f['actor'], f['grad_actor'][:n_a + self.num_actions],
                                         [list(ac.actor.parameters()), [ac.log_std]], lr=self.lr)
        self.optimizer_critic = FusedAdam(f['critic'], f['grad_critic'][:n_c], [list(ac.critic.parameters())],
                                          lr=self.lr)
        self.sync = pdist.maybe_sync()              # None on a single GPU
        dev = f['actor'].device
        self._acc = torch.zeros(8, device=dev)       # device-side running sums of `update`
        self._mom = torch.zeros(2, dtype=torch.float64, device=dev)
        self._ws = ops.Workspace(dev)
        self._ws_loss = ops.Workspace(dev)
        self._ws_vloss = ops.Workspace(dev)
        self._stage, self._stage_c = {}, {}
        self._pending_critic = []
        self._side = None
        # actor || critic on two HIP streams: pays in the small-kernel regime (MLP backbones: +12 % measured);
        # the fused point-cloud encoders already fill every CU (+3 %, and it blurs per-kernel timing), so they
        # default to the reference's serial order.  PARTMANIP_OVERLAP=0/1 overrides.
        ov = os.environ.get("PARTMANIP_OVERLAP")
        is_mlp = self.model_cfg['network']['name'] == 'MLP'
        # (round 4) PointNet++ too: since its set-abstraction levels run over the distinct rows only, a network step is ~25 launches
        # of 0.1-1.6 ms whose persistent work-groups end ragged -- the other network's launches fill the tails: 27.1 k -> 29.6 k
        # env-steps/s at cfg 3's shape (alternating A/B runs on one box)
        self.overlap = (ov == "1") if ov in ("0", "1") else (is_mlp or self.model_cfg['network']['name'] == 'PointNet2')
        # hipGraph replay of the per-mini-batch launch chains: the MLP step is ~24 kernels of 5-15 us, i.e.
        # host-launch-bound (3.7 us per launch measured).  Needs constant kernel arguments per step: sequential
        # sampler (slices of persistent storage), a fixed learning rate, no collective inside the chain.
        gr = os.environ.get("PARTMANIP_GRAPHS")
        self.use_graphs = (gr != "0") and is_mlp and self.overlap and cfg['sampler'] == 'sequential' and \
            self.lr_schedule == 'fixed'
        # Under data parallelism a gradient all-reduce sits inside every step.  "capture": RCCL collectives are stream-ordered
        # device work and are captured INTO the multi-step graphs like any kernel (backend nccl); "split": each step replays as
        # two graphs with the collective issued between them (any backend -- gloo on the 1-GPU test box).  Steps that need a
        # further collective BEFORE their loss (mini_adv_norm moments, the clipped value loss's batch-mean width) run eagerly.
        self.dp_graph_mode = None
        self.dp_degraded = None
        if self.sync is not None:
            be = torch.distributed.get_backend()
            self.dp_graph_mode = os.environ.get("PARTMANIP_DP_GRAPHS") or ("capture" if be == "nccl" else "split")
            if self.dp_graph_mode not in ("capture", "split") or self.tricks['mini_adv_norm'] or self.tricks['use_clipped_value_loss'] \
                    or os.environ.get("PARTMANIP_SOLO_GROUP", "1") != "1":    # the pre / post split exists for the grouped step only
                self.use_graphs, self.dp_graph_mode = False, None
            if not self.use_graphs:
                self.dp_graph_mode = None
            # actor and critic steps run on two streams: give the critic's all-reduces their own communicator, so that two
            # concurrently replayed graphs never interleave collectives of ONE communicator in rank-dependent order
            self.sync_c = pdist.GradSync(group=torch.distributed.new_group(), name="critic") if self.overlap else self.sync
            # The first collective of every communicator is time-boxed and its outcome agreed between the ranks over the
            # rendezvous store (dist.GradSync.probe): a communicator that cannot be built says so -- rank, device, communicator,
            # launch structure, what to set -- instead of hanging the first optimiser step.  The default communicator failing
            # leaves nothing to fall back to; the second one failing degrades ALL ranks to the conservative structure: one
            # communicator, actor and critic steps on one stream, eager launches (no collective inside a hipGraph).
            self.sync.mode = self.sync_c.mode = self.dp_graph_mode or "eager"
            if self.sync.world > 1:
                bad = self.sync.probe(dev)
                if bad is not None:
                    raise pdist.CollectiveError("data-parallel PPO cannot start: " + bad)
                bad = self.sync_c.probe(dev) if self.sync_c is not self.sync else None
                if bad is not None:
                    print(f"[ppo] {bad}\n[ppo] DEGRADED: retrying with ONE communicator, actor and critic steps on one stream, eager "
                          "launches", file=sys.stderr)
                    self.dp_degraded = dict(reason=bad, mode="one communicator ('actor'), actor and critic steps on one stream, eager launches "
                                                            "(no collective inside a hipGraph)")
                    self.sync_c, self.overlap, self.use_graphs, self.dp_graph_mode = self.sync, False, False, None
                    self.sync.mode = "eager (degraded)"
                    bad = self.sync.probe(dev)                     # the retry: the remaining communicator must still work everywhere
                    if bad is not None:
                        raise pdist.CollectiveError("data-parallel PPO cannot continue after degrading: " + bad)
        self.cu_split = os.environ.get("PARTMANIP_CU_SPLIT", "0") == "1" and self.overlap and self.sync is None
        self._graphs = {}
        self.graph_status = None           # set once a hipGraph capture has failed and the run continued eagerly (text for the logs)
        self._obs_pad = None
        self.fused_head = os.environ.get("PARTMANIP_FUSED_HEAD", "1") == "1"
        # mini-batch steps per graph: 16 consecutive steps of a network replay as one graph (cfg 2: 1 -> 1.88 M env-steps/s,
        # 4 -> 1.916 M, 16 -> 1.919 M, 64 -> 1.922 M: the boundary between two graphs costs little more than a kernel boundary)
        self.graph_steps = max(1, int(os.environ.get("PARTMANIP_GRAPH_STEPS", "16")))
        # small-step regime (MLP backbones), opt-in (PARTMANIP_PAIR=1): actor step k and critic step k as ONE launch chain --
        # every layer of the two networks is a grouped launch (two problems per grid), all eight weight gradients are two
        # launches whose split-K slabs the grouped optimiser launch sums: ~15 launches per step PAIR instead of ~48 on two
        # streams.  Identical arithmetic per network.  Measured at cfg 2 (round 2): 1.66 M env-steps/s against 1.75 M for
        # the two-stream form -- kernels of DIFFERENT kinds co-scheduled from two streams fill the CUs better than two
        # problems of the same kind in one grid -- so the two streams stay the default.
        self.pair = is_mlp and self.sync is None and os.environ.get("PARTMANIP_PAIR", "0") == "1"
        # ... and the same grouping WITHOUT pairing the networks (PARTMANIP_SOLO_GROUP, default on for MLP backbones without a
        # collective in the step): each network keeps its stream, but its four weight gradients are one or two grouped launches
        # whose split-K slabs the grouped optimiser launch sums -- no slab-reduce launches (4 per step, 12 % of the kernel time
        # at cfg 2), no separate norm pass.
        self.solo_group = is_mlp and os.environ.get("PARTMANIP_SOLO_GROUP", "1") == "1"
        # neighbourhood tables (FPS centres + ball-query indices) once per rollout; PARTMANIP_GEOM_CACHE=0 recomputes
        # them in every forward (A/B; identical results)
        self.cache_geometry = os.environ.get("PARTMANIP_GEOM_CACHE", "1") != "0"
        self._geom = None

        self.logger = logger
        self.total_envsteps = 0
        self.total_time = 0
        self.curr_iter = 0
        self.resume(cfg['resume'])
        self._broadcast_state()

    def _broadcast_state(self):
        """Data parallelism keeps ONE model: whatever each rank initialised or loaded, rank 0's parameters, Adam
        moments / step counters and observation statistics are what every replica starts from (gradients are the only
        thing exchanged afterwards, so replicas that start equal stay equal)."""
        if self.sync is None:
            return
        f = self.actor_critic.flat()
        for opt, key in ((self.optimizer_actor, 'actor'), (self.optimizer_critic, 'critic')):
            for t in (f[key], opt.m, opt.v, opt.state_dev):
                self.sync.broadcast_(t)
        if self.tricks['use_state_norm']:
            rms = self.state_norm.running_ms
            for k in ('mean', 'S', 'std'):
                setattr(rms, k, self.sync.broadcast_(getattr(rms, k).to(f['actor'].device).float().contiguous()))
            n = torch.tensor([rms.n], dtype=torch.int64, device=f['actor'].device)
            rms.n = int(self.sync.broadcast_(n).item())
            rms.sync = self.sync                         # batch moments are summed over ranks from now on (RMS.py)

    # ------------------------------------------------------------------ checkpoints (ppo.py:83-137)
    def save(self, it):
        if self.sync is not None and self.sync.rank != 0:
            return                                       # replicas are identical: rank 0 writes the checkpoint
        os.makedirs(self.save_ckpt_dir, exist_ok=True)
        save_path = pjoin(self.save_ckpt_dir, f'model_{it}.pth')
        save_dict = {
            'iteration': it,
            'model_state_dict': {k: v.clone() for k, v in self.actor_critic.state_dict().items()},
            'optimizer_actor': self.optimizer_actor.state_dict(),
            'optimizer_critic': self.optimizer_critic.state_dict(),
            'total_steps': self.total_envsteps,
            'tricks': self.tricks,
            'obs_mode': self.obs_mode,
            'model_cfg': self.model_cfg,
        }
        if self.tricks['use_state_norm']:
            save_dict['state_running_ms'] = self.state_norm.running_ms.save()
        torch.save(save_dict, save_path)
        print(f'save ckpt to {save_path}!')

    def resume(self, ckpt_path):
        self.ckpt_path = ckpt_path
        if ckpt_path is None:
            return
        print(f'load ckpt from {ckpt_path}!')
        assert os.path.exists(ckpt_path)
        ckpt = torch.load(ckpt_path, map_location=self.device, weights_only=False)
        self.actor_critic.load_state_dict(ckpt["model_state_dict"])
        self.optimizer_actor.load_state_dict(ckpt["optimizer_actor"])
        self.optimizer_critic.load_state_dict(ckpt["optimizer_critic"])
        self.curr_iter = ckpt["iteration"]
        self.total_envsteps = ckpt["total_steps"]
        for k in self.tricks_keys:
            if self.tricks[k] != ckpt['tricks'][k]:
                print(f"WARNING: trick {k} is not consistent with ckpt! saved: {ckpt['tricks'][k]}, now: {self.tricks[k]}")
                if k == 'use_state_norm':
                    print('this is not allowed')
                    exit(1)
        if self.tricks['use_state_norm']:
            self.state_norm.running_ms.load(ckpt['state_running_ms'])
        assert self.obs_mode == ckpt['obs_mode']

    # ------------------------------------------------------------------ learner
    def _views(self):
        st = self.storage
        v = lambda t: t.view(-1, t.size(-1))
        return dict(obs=v(st.observations), actions=v(st.actions), values=v(st.values), returns=v(st.returns),
                    old_logp=v(st.actions_log_prob), adv=v(st.advantages), old_mu=v(st.mu), old_sigma=v(st.sigma))

    def _epoch_plan(self, batch):
        """One pass over `mini_batch_generator`'s BatchSampler (storage.py:125-138) WITHOUT building
        Python lists of ints (5 M int objects per iteration at 4096 x 128): sequential -> (lo, n)
        ranges; random -> slices of one `torch.randperm`, which is exactly the draw
        SubsetRandomSampler.__iter__ makes from the global RNG (same sample sets, same RNG state)."""
        n_all, mb = self.storage.cur_buf_size, batch.batch_size
        n_mb = n_all // mb                                     # drop_last
        if self.storage.sampler == "sequential":
            return [(k * mb, mb) for k in range(n_mb)]
        perm = torch.randperm(n_all)
        return [perm[k * mb:(k + 1) * mb] for k in range(n_mb)]

    def _minibatch(self, views, indices, keys, stage):
        """Sequential sampler -> contiguous slices (zero copy); random -> HIP row gather (K3)."""
        if isinstance(indices, tuple):
            lo, n = indices
            mb = {k: views[k][lo:lo + n] for k in keys}
            pad = views.get('obs_pad')
            if pad is not None and 'obs' in keys:          # the same rows inside 16-byte-padded strides (weight gradient of layer 0)
                xw = pad[lo:lo + n]
                xw._pm_cols = pad._pm_cols
                mb['obs_pad'] = xw
            return mb
        n = len(indices)
        dev = views['obs'].device
        idx = indices.to(dev, non_blocking=True)
        out = {}
        for k in keys:
            src = views[k]
            buf = stage.get(k)
            if buf is None or buf.shape[0] != n:
                buf = stage[k] = torch.empty(n, src.shape[1], device=dev)
            ops.gather_rows(src, idx, buf)
            out[k] = buf
        return out

    def _slabs(self, B):
        """Split-K slabs of the weight gradients of a B-row mini-batch (ActorCritic.GRAD_SLABS, lowered until every slab owns
        rows once its range is rounded up to the kernels' K-step of 32; small mini-batches are not split)."""
        S = self.actor_critic.GRAD_SLABS if B >= 1024 else 1
        while S > 1 and (S - 1) * ((-(-B // S) + 31) // 32 * 32) >= B:
            S -= 1
        return S

    def _solo_adam(self, f, which, S, phase=None):
        """Grouped optimiser launch of one network on its own stream (small-step regime): the norm pass sums the S split-K slabs of
        the weight gradients and carries the step's running sums (one launch less), then clip + Adam.  log_std belongs to the
        actor's optimiser but is outside the clipped norm (ppo.py:351); the actor's KL predicate is the skip flag (ppo.py:337-338).

        Data parallel (self.sync): the slabs are folded first (one small launch), ONE all-reduce SUM carries the gradient and
        the step's scalars in its tail, and the optimiser's norm pass turns the sum into the mean, re-takes the KL predicate
        from the reduced KL and accumulates the reduced scalars (pm_clip_adam_desc.grad_scale / dp_scal): no separate scale /
        flag / statistics launches.  phase 'pre' / 'post': the part before / after the collective (split graphs)."""
        clip = self.tricks['use_grad_clip']
        mn = self.max_grad_norm if clip else 0.0
        sync = self.sync if which == 'actor' else getattr(self, 'sync_c', self.sync)
        n_a, n_c, A = f['n_actor'], f['n_critic'], self.num_actions
        if sync is not None:
            if phase in (None, 'pre') and S > 1:
                ops.grad_slab_sum(f[f'grad_{which}'], f[f'extra_{which}'], f[f'slab_stride_{which}'], n_a if which == 'actor' else n_c, S - 1)
            if phase is None:
                sync.sum_(f[f'grad_{which}'])
            if phase == 'pre':
                return
            S, dp = 1, (1.0 / sync.world, f[f'scal_{which}'], self.desired_kl if which == 'actor' else 0.0)
        else:
            dp = None
        if which == 'actor':
            scal = f['scal_actor']
            item = self.optimizer_actor.group_item(n=n_a + A, n_clip=n_a if clip else 0, max_norm=mn, skip_flag=scal[2:3],
                                                   extra=f['extra_actor'], extra_stride=f['slab_stride_actor'], n_sum=n_a, n_extra=S - 1,
                                                   stats=(self._acc, scal, 0), dp=dp)
        else:
            item = self.optimizer_critic.group_item(n=n_c, n_clip=n_c if clip else 0, max_norm=mn, extra=f['extra_critic'],
                                                    extra_stride=f['slab_stride_critic'], n_sum=n_c, n_extra=S - 1,
                                                    stats=(self._acc, f['scal_critic'], 1), dp=dp)
        ops.clip_adam_group([item])

    def _actor_step(self, f, views, indices, stage, phase=None):
        """One mini-batch of ppo.py:316-357 (forward, loss fwd+bwd, backward, [all-reduce], clip+Adam).  phase 'pre' / 'post':
        the part in front of / behind the data-parallel all-reduce (small-step path only: the two halves replay as graphs)."""
        ac, tricks, sync = self.actor_critic, self.tricks, self.sync
        n_a, A = f['n_actor'], self.num_actions
        scal_a = f['scal_actor']
        clip = tricks['use_grad_clip']
        if phase == 'post':
            self._solo_adam(f, 'actor', self._slabs(indices[1]), phase='post')
            return
        mb = self._minibatch(views, indices, ('obs', 'actions', 'old_logp', 'adv', 'old_mu', 'old_sigma'), stage)
        B = mb['obs'].shape[0]
        if self._geom is not None:
            ac.actor.use_geometry(self._geom, indices)
        mom, cnt = None, 0.0
        if tricks['mini_adv_norm']:                          # ppo.py:329
            ops.moments(mb['adv'].reshape(-1), self._mom, self._ws)
            cnt = sync.moments_sync(self._mom, B) if sync else B
            mom = self._mom
        chain = getattr(ac.actor, '_chain', None) if (self.solo_group and self.fused_head) else None
        if chain is not None and len(chain.linears) >= 2:
            # small-step regime: policy head + loss + head data gradient as one launch (bit-identical to the separate ones)
            h = chain.forward_hidden(mb['obs'], mb.get('obs_pad'))
            lin = chain.linears[-1]
            dh = torch.empty_like(h)
            if ops.ppo_actor_head_supported(h, lin.weight.data, dh):
                dmu = ops.padded_cols(B, A, h.device)     # (B, A) inside 16-byte rows: the head's weight gradient loads float4s
                ops.ppo_actor_head(h, lin.weight.data, lin.bias.data, chain.act, ac.log_std.data, mb['actions'], mb['old_logp'],
                                   mb['adv'], mb['old_mu'], mb['old_sigma'], ac.max_action, ac.action_activate == 'tanh',
                                   self.epsilon_clip, self.desired_kl, mom, cnt, scal_a, dmu, dh, f['grad_log_std'], self._ws_loss)
                S = self._slabs(B)
                chains_backward([chain], [dmu], [f['slab_stride_actor']], S, head_dz=[dh])
                self._solo_adam(f, 'actor', S, phase)
                return
            mu = torch.empty(B, A, device=h.device)          # (shape outside the fused kernel: finish the forward separately)
            ops.linear_fwd(h, lin.weight.data, lin.bias.data, mu, ops.ACT_NONE)
        else:
            mu = ac.actor.hip_forward(mb['obs'])
        dmu = torch.empty(B, A, device=mu.device)
        ops.ppo_actor_loss(mu, ac.log_std.data, mb['actions'], mb['old_logp'], mb['adv'], mb['old_mu'],
                           mb['old_sigma'], ac.max_action, ac.action_activate == 'tanh', self.epsilon_clip,
                           self.desired_kl, mom, cnt, scal_a, dmu, f['grad_log_std'], self._ws_loss)
        if self.solo_group:
            S = self._slabs(B)
            chains_backward([ac.actor._chain], [dmu], [f['slab_stride_actor']], S)
            self._solo_adam(f, 'actor', S, phase)
            return
        assert phase is None
        ac.actor.hip_backward(dmu)
        if sync:                                              # ONE all-reduce: grads + loss/kl in the tail; mean, KL predicate,
            self._solo_adam(f, 'actor', 1)                    # running sums and clip + Adam in the optimiser launch pair
            return
        ops.ppo_accumulate_stats(self._acc, scal_a, 0)
        # log_std belongs to this optimiser but is outside the clipped norm (ppo.py:351)
        self.optimizer_actor.step(n=n_a + A, n_clip=n_a if clip else 0,
                                  max_norm=self.max_grad_norm if clip else 0.0, skip_flag=scal_a[2:3])

    def _critic_step(self, f, views, indices, stage, phase=None):
        """One mini-batch of ppo.py:360-384 (phase: as `_actor_step`)."""
        ac, tricks = self.actor_critic, self.tricks
        sync = getattr(self, 'sync_c', self.sync)
        n_c, scal_c = f['n_critic'], f['scal_critic']
        clip = tricks['use_grad_clip']
        if phase == 'post':
            self._solo_adam(f, 'critic', self._slabs(indices[1]), phase='post')
            return
        mb = self._minibatch(views, indices, ('obs', 'returns', 'values'), stage)
        B = mb['obs'].shape[0]
        if self._geom is not None:
            ac.critic.use_geometry(self._geom, indices)
        cchain = getattr(ac.critic, '_chain', None) if self.solo_group else None
        if cchain is not None and self.fused_head and len(cchain.linears) >= 2 and not (tricks['use_clipped_value_loss'] and sync):
            # small-step regime: value head + loss + head data gradient as one launch (as the actor's, _actor_step)
            h = cchain.forward_hidden(mb['obs'], mb.get('obs_pad'))
            lin = cchain.linears[-1]
            dh = torch.empty_like(h)
            if ops.value_head_supported(h, lin.weight.data, dh):
                dv = ops.padded_cols(B, 1, h.device)
                ops.value_head(h, lin.weight.data, lin.bias.data, cchain.act, mb['returns'], mb['values'],
                               tricks['use_clipped_value_loss'], self.epsilon_clip, None, 1.0, scal_c, dv, dh, self._ws_vloss)
                S = self._slabs(B)
                chains_backward([cchain], [dv], [f['slab_stride_critic']], S, head_dz=[dh])
                self._solo_adam(f, 'critic', S, phase)
                return
            value = torch.empty(B, 1, device=h.device)
            ops.linear_fwd(h, lin.weight.data, lin.bias.data, value, ops.ACT_NONE)
        else:
            value = ac.critic.hip_forward(mb['obs'], x_w=mb.get('obs_pad')) if cchain is not None else ac.critic.hip_forward(mb['obs'])
        clip_mean = None
        if tricks['use_clipped_value_loss'] and sync:
            clip_mean = sync.mean_((self.epsilon_clip * mb['values']).abs().mean().reshape(1))
        dv = ops.padded_cols(B, 1, value.device) if cchain is not None else torch.empty(B, 1, device=value.device)
        ops.value_loss(value, mb['returns'], mb['values'], tricks['use_clipped_value_loss'], self.epsilon_clip,
                       clip_mean, 1.0, scal_c, dv)
        if self.solo_group:
            S = self._slabs(B)
            chains_backward([ac.critic._chain], [dv], [f['slab_stride_critic']], S)
            self._solo_adam(f, 'critic', S, phase)
            return
        assert phase is None
        ac.critic.hip_backward(dv)
        if sync:
            self._solo_adam(f, 'critic', 1)
            return
        ops.ppo_accumulate_stats(self._acc, scal_c, 1)
        self.optimizer_critic.step(n=n_c, n_clip=n_c if clip else 0, max_norm=self.max_grad_norm if clip else 0.0)

    def _pair_step(self, f, views, ia, ic):
        """Actor mini-batch `ia` (ppo.py:316-357) and critic mini-batch `ic` (ppo.py:360-384) in one grouped launch chain."""
        ac, tricks = self.actor_critic, self.tricks
        n_a, n_c, A = f['n_actor'], f['n_critic'], self.num_actions
        scal_a, scal_c = f['scal_actor'], f['scal_critic']
        clip = tricks['use_grad_clip']
        mba = self._minibatch(views, ia, ('obs', 'actions', 'old_logp', 'adv', 'old_mu', 'old_sigma'), self._stage)
        mbc = self._minibatch(views, ic, ('obs', 'returns', 'values'), self._stage_c)
        B = mba['obs'].shape[0]
        chains = [ac.actor._chain, ac.critic._chain]
        mu, value = chains_forward(chains, [mba['obs'], mbc['obs']])
        mom, cnt = None, 0.0
        if tricks['mini_adv_norm']:
            ops.moments(mba['adv'].reshape(-1), self._mom, self._ws)
            mom, cnt = self._mom, B
        dmu = torch.empty(B, A, device=mu.device)
        ops.ppo_actor_loss(mu, ac.log_std.data, mba['actions'], mba['old_logp'], mba['adv'], mba['old_mu'], mba['old_sigma'],
                           ac.max_action, ac.action_activate == 'tanh', self.epsilon_clip, self.desired_kl, mom, cnt, scal_a,
                           dmu, f['grad_log_std'], self._ws_loss)
        dv = torch.empty(mbc['obs'].shape[0], 1, device=value.device)
        ops.value_loss(value, mbc['returns'], mbc['values'], tricks['use_clipped_value_loss'], self.epsilon_clip, None, 1.0,
                       scal_c, dv)
        S = self._slabs(B) if mbc['obs'].shape[0] == B else 1
        chains_backward(chains, [dmu, dv], [f['slab_stride_actor'], f['slab_stride_critic']], S)
        ops.ppo_accumulate_stats(self._acc, scal_a, 0)
        ops.ppo_accumulate_stats(self._acc, scal_c, 1)
        mn = self.max_grad_norm if clip else 0.0
        ops.clip_adam_group([
            self.optimizer_actor.group_item(n=n_a + A, n_clip=n_a if clip else 0, max_norm=mn, skip_flag=scal_a[2:3],
                                            extra=f['extra_actor'], extra_stride=f['slab_stride_actor'], n_sum=n_a, n_extra=S - 1),
            self.optimizer_critic.group_item(n=n_c, n_clip=n_c if clip else 0, max_norm=mn, extra=f['extra_critic'],
                                             extra_stride=f['slab_stride_critic'], n_sum=n_c, n_extra=S - 1)])

    # ---- hipGraph replay of the per-mini-batch launch chains (small-step regime) -----------------
    def _graph_table(self, views):
        """Graphs are valid for one (rollout buffers, learning rates) configuration: every kernel argument of a
        mini-batch step is then a constant -- device pointers into persistent storage / flat buffers, sizes,
        hyper-parameters -- while everything that changes between steps (parameters, Adam moments and step
        counter, the KL skip flag, the running sums) lives in device memory."""
        f = self.actor_critic.flat()
        key = (views['obs'].data_ptr(), views['obs_pad'].data_ptr() if 'obs_pad' in views else 0, views['adv'].data_ptr(), views['returns'].data_ptr(),
               f['actor'].data_ptr(), f['critic'].data_ptr(), self.optimizer_actor.param_groups[0]['lr'], self.optimizer_critic.param_groups[0]['lr'])
        if self._graphs.get('key') != key:
            self._graphs = {'key': key, 'seen': set(),
                            'pool': {'a': torch.cuda.graph_pool_handle(), 'c': torch.cuda.graph_pool_handle(),
                                     'p': torch.cuda.graph_pool_handle()}}
        return self._graphs

    def _replay(self, graphs, k, fn, stream):
        """1st encounter of step k: run eagerly (sizes workspaces, warms allocator); 2nd: capture; then replay."""
        with torch.cuda.stream(stream):
            g = graphs.get(k)
            if g is not None:
                g.replay()
            elif k not in graphs['seen']:
                graphs['seen'].add(k)
                fn()
            else:
                # capture needs a non-default stream; replay may use any stream (the actor's is the default one)
                if graphs.get('broken'):                       # an earlier capture failed: the rest of the run is eager
                    fn()
                    return
                cap = self._side if stream is not self._side else self._cap
                cap.wait_stream(stream)
                g = torch.cuda.CUDAGraph()
                err = None
                try:
                    with torch.cuda.graph(g, pool=graphs['pool'][k[0]], stream=cap):
                        fail = os.environ.get("PARTMANIP_TEST_CAPTURE_FAIL")
                        if fail == "1" or (fail is not None and fail.startswith("rank") and self.sync is not None
                                           and int(fail[4:]) == self.sync.rank):
                            raise RuntimeError("PARTMANIP_TEST_CAPTURE_FAIL")
                        fn()
                except Exception as e:                         # e.g. a collective that cannot be captured on this stack
                    err = e
                stream.wait_stream(cap)
                # Nothing of the step has executed yet (capture only records).  Under data parallelism the ranks must take the
                # SAME path -- a rank replaying a graph with the all-reduce inside while another issues it eagerly would order
                # the collectives of one communicator differently and hang -- so they agree on the outcome first (a capture can
                # fail on one rank alone: memory).  One tiny all-reduce per captured graph, on this network's communicator.
                sy = getattr(self, 'sync_c', self.sync) if k[0] == 'c' else self.sync
                failed_here = err is not None
                if sy is not None and sy.world > 1:
                    failed = sy.any_(failed_here, views_device=self.device)
                else:
                    failed = failed_here
                if failed:
                    # run it eagerly and stop capturing; graphs captured before stay valid.  `dp_graph_mode` keeps naming the
                    # launch structure of the data-parallel step (the '== "split"' checks depend on it); what happened is
                    # said in `graph_status` (bench.py prints it).
                    graphs['broken'] = True
                    why = f"{type(err).__name__}: {str(err)[:120]}" if failed_here else "capture failed on another rank"
                    self.graph_status = f"eager ({self.dp_graph_mode or 'graphs'}: capture failed: {why})"
                    print(f"[ppo] hipGraph capture of step {k[:2]} failed, continuing eagerly: {why}", file=sys.stderr)
                    del g
                    fn()
                    return
                graphs[k] = g
                g.replay()

    def update(self, it):
        """ppo.py:307-411.  The reference runs all actor epochs, then all critic epochs; the two loops touch
        disjoint parameters / optimisers and only read the rollout, so step k of the critic loop is issued on
        a second HIP stream next to step k of the actor loop (bit-identical results): small kernels of one
        network fill the CUs the other leaves idle, and under data parallelism each all-reduce overlaps the
        other network's compute.  Index lists are drawn in the reference's order (all actor epochs first)."""
        ac = self.actor_critic
        f = ac.flat()
        if not self.optimizer_actor.bound_to(f['actor']):        # the flat buffers were rebuilt (.to() / .float() / ...)
            self.optimizer_actor.rebind(f['actor'], f['grad_actor'][:f['n_actor'] + self.num_actions])
        if not self.optimizer_critic.bound_to(f['critic']):
            self.optimizer_critic.rebind(f['critic'], f['grad_critic'][:f['n_critic']])
        views = self._views()
        if self.solo_group and views['obs'].shape[1] % 4 != 0 and self.storage.sampler == "sequential":
            # small-step regime: the rollout's observations once more in rows padded to 16 bytes (53 -> 56 floats), so that the
            # input layer's weight gradient joins the other layers' LDS-DMA launch (one copy per rollout, 128 of 0.28 ms at cfg 2)
            o = views['obs']
            if self._obs_pad is None or self._obs_pad.shape != o.shape or self._obs_pad.device != o.device:
                self._obs_pad = ops.padded_cols(o.shape[0], o.shape[1], o.device, zero=True)
            self._obs_pad.copy_(o)
            views['obs_pad'] = self._obs_pad
        self._acc.zero_()
        # backbones whose sampling / grouping depends on the coordinates only (PointNet2) build their
        # neighbourhood tables once per rollout; every epoch of both networks then reuses them
        self._geom = None
        if self.cache_geometry and hasattr(ac.actor, 'precompute_geometry'):
            self._geom = ac.actor.precompute_geometry(views['obs'])
        batch = self.storage.mini_batch_generator(self.num_mini_batches)
        lists_a = [self._epoch_plan(batch) for _ in range(self.n_updates)]          # ppo.py:315-316
        lists_c = [self._epoch_plan(batch) for _ in range(self.n_updates)]          # ppo.py:359-360
        if self._geom is not None and hasattr(ac.actor, 'precompute_plans'):
            # sequential mini-batches are the same slices in every epoch and for both networks: their packed-row plans are built
            # here, back to back, and sized by ONE host read -- not one read per (level, slice) in the middle of the first epoch,
            # where it would stall the two streams of actor || critic
            slices = sorted({i for ep in lists_a + lists_c for i in ep if isinstance(i, tuple)})
            for net in (ac.actor, ac.critic):
                net.precompute_plans(self._geom, views['obs'], slices)
        main = outer = torch.cuda.current_stream()
        if self.overlap:
            if self._side is None:
                if self.cu_split:
                    # A/B: each network on its own HALF of every XCD's CUs (a CU mask cannot select XCDs:
                    # ops.cu_masked_stream), so neither chain's work-groups queue behind the other's
                    self._main_masked, self._side = ops.cu_masked_stream(0, 128), ops.cu_masked_stream(128, 128)
                    self._cap = torch.cuda.Stream()
                else:
                    self._side, self._cap = torch.cuda.Stream(), torch.cuda.Stream()
            if self.cu_split:
                main = self._main_masked
                main.wait_stream(outer)
            side = self._side
            side.wait_stream(outer)                                      # returns / advantages are ready
        graphs = self._graph_table(views) if self.use_graphs else None
        chunk_a, chunk_c = [], []
        for ep, (la, lc) in enumerate(zip(lists_a, lists_c)):
            assert len(la) == len(lc)
            for k, (ia, ic) in enumerate(zip(la, lc)):
                if self.pair:
                    if graphs is not None:
                        self._replay(graphs, ('p', ia, ic), lambda: self._pair_step(f, views, ia, ic), main)
                    else:
                        self._pair_step(f, views, ia, ic)
                    continue
                if graphs is not None and self.dp_graph_mode == "split":
                    # data parallel, any backend: [graph: forward .. backward .. slab fold] -> all-reduce -> [graph: optimiser]
                    self._replay(graphs, ('a', ia, 'pre'), lambda: self._actor_step(f, views, ia, self._stage, 'pre'), main)
                    self.sync.sum_(f['grad_actor'])
                    self._replay(graphs, ('a', ia, 'post'), lambda: self._actor_step(f, views, ia, self._stage, 'post'), main)
                    self._replay(graphs, ('c', ic, 'pre'), lambda: self._critic_step(f, views, ic, self._stage_c, 'pre'), side)
                    with torch.cuda.stream(side):
                        self.sync_c.sum_(f['grad_critic'])
                    self._replay(graphs, ('c', ic, 'post'), lambda: self._critic_step(f, views, ic, self._stage_c, 'post'), side)
                    continue
                if graphs is not None:
                    if self.graph_steps > 1:
                        chunk_a.append(ia)
                        chunk_c.append(ic)
                        if len(chunk_a) == self.graph_steps or k == len(la) - 1:       # a chunk never crosses an epoch
                            ca, cc = tuple(chunk_a), tuple(chunk_c)
                            self._replay(graphs, ('a',) + ca, lambda: [self._actor_step(f, views, i, self._stage) for i in ca], main)
                            self._replay(graphs, ('c',) + cc, lambda: [self._critic_step(f, views, i, self._stage_c) for i in cc], side)
                            chunk_a, chunk_c = [], []
                        continue
                    self._replay(graphs, ('a', ia), lambda: self._actor_step(f, views, ia, self._stage), main)
                    self._replay(graphs, ('c', ic), lambda: self._critic_step(f, views, ic, self._stage_c), side)
                    continue
                with torch.cuda.stream(main):
                    self._actor_step(f, views, ia, self._stage)
                if self.overlap:
                    with torch.cuda.stream(side):
                        self._critic_step(f, views, ic, self._stage_c)
                else:
                    self._pending_critic.append(ic)
            assert not chunk_a and not chunk_c, "a mini-batch chunk was left unissued at the end of the epoch"
        if self.overlap:
            outer.wait_stream(side)
            if main is not outer:
                outer.wait_stream(main)
        else:                                                            # reference order: critic loop afterwards
            for ic in self._pending_critic:
                self._critic_step(f, views, ic, self._stage_c)
            self._pending_critic = []

        self._geom = None
        acc = self._acc.tolist()                              # the only host sync of the update
        from ..algo_utils import network as _net
        if _net.CHAIN_LAUNCH:                                 # opt-in chained layers: a stripe hand-off that gave up must not pass silently
            for net in (ac.actor, ac.critic):
                ch = getattr(net, '_chain', None)
                if ch is not None and ch._cws is not None and ops.chain_gave_up(ch._cws):
                    raise RuntimeError("a chained-layer launch ran into its spin limit (PARTMANIP_CHAIN=1): its results are invalid")
        sum_surr, sum_kl, kl_max, count, sum_v, n_v = acc[:6]
        if not all(np.isfinite(acc[:6])):
            print("WARNING: non-finite training statistics (loss / KL sums: "
                  f"{acc[:6]}): the policy has diverged or the rollout holds NaN/Inf")
        mean_value_loss = sum_v / (self.n_updates * len(lists_c[0]))
        mean_surrogate_loss = sum_surr / count                # ZeroDivisionError if every mb was skipped, as ppo.py:387
        mean_kl_mean = sum_kl / count

        if self.lr_schedule == 'linear_decay':
            lr_now = max(self.lr * (1 - it / self.max_iter), 1e-5)
        elif self.lr_schedule == 'step_decay':
            lr_now = 1e-5 if it > self.max_iter // 2 else self.lr
        else:
            lr_now = None
        if lr_now is not None:                                # actor optimiser only (ppo.py:392,399)
            for g in self.optimizer_actor.param_groups:
                g['lr'] = lr_now

        self.log_dict['Train/value_gt_return_mean'] = self.storage.returns.mean()
        self.log_dict['Train/value_gt_return_max'] = self.storage.returns.max()
        self.log_dict['Train/learning_rate'] = self.optimizer_actor.param_groups[0]['lr']
        self.log_dict['Train/value_function_loss'] = mean_value_loss
        self.log_dict['Train/surrogate_loss'] = mean_surrogate_loss
        self.log_dict['Train/kl'] = mean_kl_mean
        self.log_dict['Train/kl_max'] = kl_max
        self.log_dict['Train/kl_update_count'] = int(count)

    def learn(self, last_values):
        """The `learn_time` window of ppo.py:256-262: returns + update + clear."""
        ms = self.sync.moments_sync if self.sync else None
        self.storage.compute_returns(last_values, self.gamma, self.lam, moments_sync=ms)
        self.update(self.curr_iter)
        self.storage.clear()

    # ------------------------------------------------------------------ rollout / eval (simulator-bound)
    def _norm(self, obs, update):
        return self.state_norm(obs, update=update) if self.tricks['use_state_norm'] else obs

    def use_info_update_logdict(self, info_lst, mode):
        """ppo.py:295-305: per-key mean and mean-of-per-env-max over the collected steps."""
        for key in info_lst[0]:
            assert len(info_lst[0][key].shape) == 1, f"{key}: {info_lst[0][key].shape}"
            allv = torch.stack([info[key].float() for info in info_lst], dim=-1)
            self.log_dict[f'{mode}/{key}_mean'] = torch.mean(allv)
            self.log_dict[f'{mode}/{key}_max'] = torch.mean(allv.max(dim=-1)[0])

    def eval(self):
        """ppo.py:139-203."""
        self.actor_critic.eval()
        self.vec_env.train_test_flag = 'test'
        if self.test_only:
            self.log_dict = {}
        ep_infos = []
        with torch.no_grad():
            for _ in range(self.eval_round):
                poses = []
                curr_obs = self.vec_env.reset()[self.obs_mode]
                for i in range(self.max_episode_length):
                    curr_obs = self._norm(curr_obs, False)
                    actions, _ = self.actor_critic.act_cri(curr_obs)
                    img = pjoin(self.logger.save_video_dir, f"Iter{self.curr_iter}", f"{i}.png") if self.save_video else None
                    next_obs, rews, _, infos = self.vec_env.step(actions, save_image_path=img)
                    infos['action_t'] = actions[:, :3].mean(dim=-1)
                    infos['action_r'] = actions[:, 3:6].mean(dim=-1)
                    infos['action_gripper'] = actions[:, -1]
                    infos['succ_rate'] = self.vec_env.success
                    ep_infos.append(deepcopy(infos))
                    if self.save_pose:
                        d = self.vec_env.save_scene_pose(pjoin(self.logger.save_pose_dir, f"Iter{self.curr_iter}", f"{i}.npy"))
                        d['state'], d['action'] = curr_obs.cpu().numpy(), actions.cpu().numpy()
                        poses.append(deepcopy(d))
                    curr_obs = next_obs[self.obs_mode]
                if self.save_pose:
                    for i, d in enumerate(poses):
                        d['success'] = ep_infos[-1]['obj_up_flag'].cpu().numpy()
                        np.save(pjoin(self.logger.save_pose_dir, f"Iter{self.curr_iter}", f"{i}.npy"), d)
                if self.save_video:
                    path2video(pjoin(self.logger.save_video_dir, f"Iter{self.curr_iter}"))
        mode = 'Test' if self.test_only else 'Val'
        self.use_info_update_logdict(ep_infos, mode)
        if self.tricks['use_state_norm'] and self.update_RMS:
            rate = self.log_dict[f'{mode}/succ_rate_max']
            if self.sync is not None:                    # one decision for all replicas: the mean over the env shards
                rate = self.sync.mean_(torch.as_tensor(rate, dtype=torch.float32, device=self.device).reshape(1).clone())[0]
            if rate > 0.5:
                self.update_RMS = False

    def run(self):
        """ppo.py:205-293: collect n_steps transitions per env, learn, log."""
        if self.test_only:
            self.eval()
            self.logger.info(self.log_dict, self.curr_iter)
            return
        upd = lambda: getattr(self, 'update_RMS', False)
        curr_obs = self._norm(self.vec_env.reset()[self.obs_mode], upd())
        while self.curr_iter < self.max_iter:
            self.curr_iter += 1
            self.actor_critic.train()
            self.vec_env.train_test_flag = 'train'
            self.log_dict = {}
            ep_infos = []
            t0 = time.time()
            for _ in range(self.n_steps):
                actions, logp, values, mu, sigma = self.actor_critic.random_act_cri(curr_obs)
                next_obs, rews, dones, infos = self.vec_env.step(actions)
                self.storage.add_transitions(curr_obs, actions, rews, dones, self.vec_env.reset_succ, values, logp,
                                             mu, sigma)
                infos['action_t'] = actions[:, :3].abs().mean(dim=-1)
                infos['action_r'] = actions[:, 3:6].abs().mean(dim=-1)
                infos['action_gripper'] = actions[:, -1].abs()
                infos['value_pred'] = values.squeeze(-1)
                curr_obs = self._norm(next_obs[self.obs_mode], upd())
                ep_infos.append(deepcopy(infos))
            last_values = self.actor_critic.cri(curr_obs)
            torch.cuda.synchronize()
            collection_time = time.time() - t0

            t0 = time.time()
            self.learn(last_values)
            torch.cuda.synchronize()
            learn_time = time.time() - t0

            self.total_envsteps += self.n_steps * self.vec_env.num_envs
            self.total_time += collection_time + learn_time
            action_std = self.actor_critic.log_std.exp()
            self.log_dict['Progress/total_steps'] = self.curr_iter
            self.log_dict['Progress/collection_time'] = collection_time
            self.log_dict['Progress/learn_time'] = learn_time
            self.log_dict['Progress/FPS'] = int(self.n_steps * self.vec_env.num_envs / (collection_time + learn_time))
            self.log_dict['Progress/learner_env_steps_per_s'] = self.n_steps * self.vec_env.num_envs / learn_time
            self.log_dict['Train/mean_action_noise_std'] = action_std.mean().item()
            self.log_dict['Train/mean_t_noise_std'] = action_std[:3].mean()
            self.log_dict['Train/mean_r_noise_std'] = action_std[3:-1].mean()
            self.log_dict['Train/mean_gripper_noise_std'] = action_std[-1]
            self.use_info_update_logdict(ep_infos, 'Train')

            if self.curr_iter % self.eval_freq == 0:
                self.eval()
                curr_obs = self._norm(self.vec_env.reset()[self.obs_mode], upd())
            if self.curr_iter % self.save_freq == 0:
                self.save(self.curr_iter)
            self.logger.info(self.log_dict, self.curr_iter)
