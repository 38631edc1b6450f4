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
    Equal shards (P divisible by the world size: every BASELINE config) go through ONE all_gather_into_tensor straight into
    the result -- no per-rank staging buffers, no concatenation pass (C5: P extra copy kernels and a second 58.7 MB pass in the
    list form of rounds 1-5).
    force: run the collective at world size 1 too (rehearsals of the RCCL path on a one-GPU box)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return local_means
    local = local_means.contiguous()
    if n_particles_total % world == 0:
        assert local.shape[0] * world == n_particles_total, (local.shape, n_particles_total, world)
        out = torch.empty((n_particles_total, *local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    # ragged split (shard sizes differ by one): every shard padded to the largest, the same single collective, the padding rows
    # cut on arrival (a list all_gather of unequal tensors is refused by gloo and serialised into per-rank broadcasts by RCCL)
    sizes = [hi - lo for lo, hi in (shard_range(n_particles_total, r, world) for r in range(world))]
    assert local.shape[0] == sizes[dist.get_rank(group)], (local.shape, sizes)
    n_max = max(sizes)
    padded = local if local.shape[0] == n_max else torch.cat([local, local.new_zeros((n_max - local.shape[0], *local.shape[1:]))], 0)
    out = torch.empty((world * n_max, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * n_max:r * n_max + n] for r, n in enumerate(sizes)], 0)


def global_diag_mean(local_diag_sum, n_local, group=None):
    """GPMP2 trust-region damping (gpmp2.py:361-367, quirk Q9): batch mean of diag(A^T K A) over ALL shards
    from each shard's local sum (mpb_gpmp2_diag)."""
    s = local_diag_sum.clone()
    n = torch.tensor([float(n_local)], dtype=torch.float64, device=s.device)
    dist.all_reduce(s, group=group)
    dist.all_reduce(n, group=group)
    return s / n
