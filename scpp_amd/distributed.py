"""Batch sharding across the GPUs of one node: every trajectory optimisation is
self-contained, so ranks own contiguous, disjoint blocks of instance ids and exchange nothing until the
single all-gather of the result trajectories (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
import numpy as np


def shard_range(total, world, rank):
    """Contiguous static shard [lo, hi) of `total` instances for `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_results(out):
    """[B][K*14 + K*4 + 4] float64 payload per instance: X, U, sigma, nu_norm, sc_iters, converged (7.2 KB at K=50)."""
    B = out["X"].shape[0]
    return np.concatenate(
        [out["X"].reshape(B, -1), out["U"].reshape(B, -1), out["sigma"][:, None], out["nu_norm"][:, None],
         out["sc_iters"][:, None].astype(np.float64), out["converged"][:, None].astype(np.float64)], axis=1)


def unpack_results(buf, K):
    B = buf.shape[0]
    nX, nU = K * 14, K * 4
    return dict(X=buf[:, :nX].reshape(B, K, 14), U=buf[:, nX:nX + nU].reshape(B, K, 4), sigma=buf[:, nX + nU],
                nu_norm=buf[:, nX + nU + 1], sc_iters=buf[:, nX + nU + 2].astype(np.int32), converged=buf[:, nX + nU + 3].astype(np.int32))


def solve_sharded(alg, model, total, seed, dist=None, device=None):
    """Solve instances [0,total) sharded over dist's world; every rank returns the gathered results of ALL instances."""
    import torch

    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    lo, hi = shard_range(total, world, rank)
    x0 = model.randomized_initial_states(hi - lo, seed=seed, first=lo)
    alg.solve(x0)
    mine = torch.from_numpy(pack_results(alg.getSolution()))
    if device is not None:
        mine = mine.to(device)
    return unpack_results(gather_rows(mine, total, dist).cpu().numpy(), alg.opts.K)


def gather_rows(mine, total, dist=None):
    """The one collective of the path: all-gather of the per-instance result rows of every rank's contiguous shard."""
    import torch

    world = dist.get_world_size() if dist is not None else 1
    if dist is None or world == 1:
        return mine
    sizes = [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        full = torch.empty((total, mine.shape[1]), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(full, mine)
        return full
    # uneven shards: pad to the largest shard (collectives need equal sizes), trim after the gather
    smax = max(sizes)
    pad = torch.zeros((smax, mine.shape[1]), dtype=mine.dtype, device=mine.device)
    pad[: mine.shape[0]] = mine
    buf = torch.empty((world * smax, mine.shape[1]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(buf, pad)
    return torch.cat([buf[r * smax: r * smax + sizes[r]] for r in range(world)], dim=0)


def mpc_solve_sharded(alg, model, total, seed, dist=None, device=None):
    """Linear MPC (MPCAlgorithm): controllers [0,total) sharded like the trajectories above; every rank returns the first
    input, both costs, status and iteration count of ALL controllers ([total][2 + 2 + 2] float64 rows)."""
    import torch

    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    lo, hi = shard_range(total, world, rank)
    alg.setInitialState(model.randomized_initial_states(hi - lo, seed=seed, first=lo))
    alg.setFinalState(model.p.x_final)
    alg.solve()
    out = alg.getSolution()
    mine = torch.from_numpy(np.concatenate([out["U"][:, 0, :], out["cost"], out["status"][:, None].astype(np.float64),
                                            out["iters"][:, None].astype(np.float64)], axis=1))
    if device is not None:
        mine = mine.to(device)
    full = gather_rows(mine, total, dist).cpu().numpy()
    return dict(u0=full[:, 0:2], cost=full[:, 2:4], status=full[:, 4].astype(np.int32), iters=full[:, 5].astype(np.int32))
