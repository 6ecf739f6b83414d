"""Test harness, not product: ONE rank of the multi-process run of the HIP backend on a 1-GPU box
(tests/test_gpu_dist.py starts two of these as fresh child processes; every rank uses cuda:0 and the ranks talk over gloo).
The process touches the GPU only here, after it has started -- nothing is re-executed.
  python tests/dist_hip_rank.py <total_envs> <steps> <out.npz>          (RANK / WORLD_SIZE / MASTER_* from the environment)"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def params(pkg):
    return pkg.Params(planner='Primitive', gaze_method='Oxford', agent_number=10, agent_radius=15, agent_max_speed=20,
                      drone_max_speed=40, map_id=1)


def main():
    total, steps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import numpy as np
    import drone2d_amd as pkg
    from drone2d_amd import dist as d2dist, _lib
    rank, world, _ = d2dist.init_process_group('gloo')
    hip = _lib.HipBackend('cuda:0')
    env = d2dist.make_shard(params(pkg), total, device='cuda:0', backend=hip, planner='Primitive', device_plugins=True, gaze='Oxford')
    lo, hi = d2dist.shard_range(total, rank, world)
    assert env.num_envs == hi - lo and env.env_offset == lo
    for _ in range(2):
        env.closed_loop(steps // 2, auto_reset=True)
    env.sync()
    stats = d2dist.gather_episode_stats(env, total)
    np.savez(out, rank=rank, lo=lo, hi=hi, stats=stats.cpu().numpy(),
             **{k: env.state.t[k].cpu().numpy() for k in ('drone', 'counters', 'agents', 'dmap', 'gt', 'kf', 'flags', 'action')},
             traj_hdr=env.plugins.t['traj_hdr'].cpu().numpy(), seen_step=env.plugins.t['seen_step'].cpu().numpy())
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
