"""Multi-GPU host logic: one process per GPU, particles sharded in contiguous blocks, no data-path
collective.  STOMP / CHOMP / MPPI problems are independent, so the only communication is
the final gather of the (P,H,d) means -- one RCCL all-gather over xGMI (backend "nccl" on ROCm) -- plus,
for GPMP2 with trust_region=True, a per-iteration all-reduce of an H*2D fp64 vector (quirk Q9).

Everything here is device-agnostic torch.distributed code, so it is exercised on CPU with gloo in
tests/test_cpu_sharding.py.
"""
import torch
import torch.distributed as dist


def shard_range(n_particles, rank, world):
    """Contiguous block [lo, hi) of this rank; sizes differ by at most one."""
    base, rem = divmod(n_particles, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_eps(eps, rank, world, particle_axis=2):
    """Slice pre-drawn standard normals in the reference's (S,d,P,H) order along the particle axis
    (non-contiguous in memory -- SURVEY.md H1) and make the shard contiguous for the C-ABI."""
    lo, hi = shard_range(eps.shape[particle_axis], rank, world)
    return eps.narrow(particle_axis, lo, hi - lo).contiguous()


def gather_means(local_means, n_particles_total, group=None, force=False):
    """All-gather the optimised means of every shard into the full (P,H,d) tensor (same on all ranks).
    force: run the collective at world size 1 too (rehearsals of the RCCL path on a one-GPU box)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return local_means
    sizes = [shard_range(n_particles_total, r, world) for r in range(world)]
    bufs = [torch.empty((hi - lo, *local_means.shape[1:]), dtype=local_means.dtype, device=local_means.device)
            for lo, hi in sizes]
    dist.all_gather(bufs, local_means.contiguous(), group=group)
    return torch.cat(bufs, 0)


def global_diag_mean(local_diag_sum, n_local, group=None):
    """GPMP2 trust-region damping (gpmp2.py:361-367, quirk Q9): batch mean of diag(A^T K A) over ALL shards
    from each shard's local sum (mpb_gpmp2_diag)."""
    s = local_diag_sum.clone()
    n = torch.tensor([float(n_local)], dtype=torch.float64, device=s.device)
    dist.all_reduce(s, group=group)
    dist.all_reduce(n, group=group)
    return s / n
