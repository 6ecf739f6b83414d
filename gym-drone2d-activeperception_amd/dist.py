"""Sharding the env batch across the GPUs of one node.

Envs are independent: a global batch of `total_envs` is cut into contiguous ranges, one per rank (one
process per GPU), and every rank steps its shard with zero per-step communication.  Env identity is the
GLOBAL env id (world seed = map_id + global id), so an 8-shard run equals the 1-GPU run env for env.
The only exchange is an all_gather of per-episode statistics (RCCL over xGMI on GPUs; gloo on CPU for
the tests) at reporting time — a few dozen bytes per env, latency-bound, one collective.
"""
import os

import torch


def shard_range(total_envs, rank, world_size):
    """Contiguous [start, stop) of global env ids owned by `rank` (remainder spread over the first ranks)."""
    q, r = divmod(total_envs, world_size)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def rank_info():
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def init_process_group(backend=None):
    """torch.distributed over RCCL ('nccl' on ROCm) when a GPU is present, gloo otherwise."""
    import torch.distributed as dist
    rank, world, local = rank_info()
    if world == 1 or dist.is_initialized():
        return rank, world, local
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local}'))
    else:
        dist.init_process_group(backend)
    return rank, world, local


def make_shard(params, total_envs, device=None, backend=None, planner=None, workers=0, **env_kw):
    """The VecDrone2DEnv of this rank's shard of a `total_envs` global batch (`env_kw`: device_plugins / gaze ...)."""
    from .vec_env import VecDrone2DEnv, build_worlds
    rank, world, local = rank_info()
    start, stop = shard_range(total_envs, rank, world)
    worlds = build_worlds(params, stop - start, env_offset=start, workers=workers)
    if device is None:
        device = f'cuda:{local}'
    return VecDrone2DEnv(params, stop - start, device=device, planner=planner, env_offset=start,
                         backend=backend, worlds=worlds, **env_kw)


def gather_episode_stats(env, total_envs=None):
    """all_gather of VecDrone2DEnv.episode_stats() -> [total_envs, 8] on every rank, ordered by global env id.
    Shards may differ in size by one env; they are padded to the largest for the collective."""
    import torch.distributed as dist
    stats = env.episode_stats()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return stats
    world = dist.get_world_size()
    total = total_envs if total_envs is not None else None
    home = stats.device
    if dist.get_backend() == 'gloo':   # (a dry run of the multi-rank path on one box: the collective runs on host tensors)
        stats = stats.cpu()
    n_local = torch.tensor([stats.shape[0]], dtype=torch.int64, device=stats.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    pad = torch.zeros((cap, stats.shape[1]), dtype=stats.dtype, device=stats.device)
    pad[:stats.shape[0]] = stats
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.cat([p[:n] for p, n in zip(parts, sizes)]).to(home)
    if total is not None:
        assert out.shape[0] == total
    return out
