"""Multi-GPU extraction: one process per GPU, utterances sharded as independent contiguous
blocks, ONE collective (all_gather over RCCL/xGMI) to collect the embeddings.

The reference shards through the filesystem: tools/extract_embedding.sh:39-67 does
`split -l ceil(N/nj)` into contiguous chunks, job k -> GPU k % num_gpus, then `cat xvector_*.scp`.
Here rank r of G takes utterances [r*ceil(N/G), (r+1)*ceil(N/G)) -- the same contiguous rule, so
concatenating the shards in rank order reproduces the list order -- and the `cat` becomes a single
`all_gather_into_tensor` of equal-size padded (ceil(N/G), E) float32 blocks (payload: a few MB,
latency-bound on xGMI; no other collective touches the data path).
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world_size: int):
    """Contiguous shard [lo, hi) of rank `rank` (last shards may be short or empty)."""
    per = (n_total + world_size - 1) // world_size if world_size > 0 else n_total
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def shard_size(n_total: int, world_size: int) -> int:
    return (n_total + world_size - 1) // world_size


def force_group() -> bool:
    """WS_DIST_FORCE_GROUP=1: build the process group and issue every collective also at world size 1.  A lone rank
    needs neither; the switch exists so that a one-GPU box executes the RCCL code paths for real (init with a device id,
    all_gather_into_tensor, device-side all_reduce, the device barrier) -- tests/test_gpu_parity.py::test_rccl_*."""
    return os.environ.get("WS_DIST_FORCE_GROUP") == "1"


def collectives_active(group=None) -> bool:
    """True when the data path has to go through torch.distributed: several ranks, or a forced one-rank group."""
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or force_group())


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_ADDR / MASTER_PORT).  Returns (rank, world_size, local_rank).
    backend 'nccl' is RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_group()) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("WS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Every rank passes its (n_local, E) shard (n_local = len of shard_range); every rank gets the
    full (n_total, E) tensor in list order."""
    if not collectives_active(group):
        return local[:n_total]
    world = dist.get_world_size(group)
    per = shard_size(n_total, world)
    width = local.shape[1:]
    if local.shape[0] == per:                     # the usual case: full shard, nothing to pad
        padded = local.contiguous()
    else:
        padded = torch.zeros((per,) + tuple(width), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(width), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        # gloo has no CUDA all_gather_into_tensor: stage through the host (test / debug path only;
        # the production backend is "nccl" = RCCL over xGMI)
        host = [torch.empty(padded.shape, dtype=padded.dtype) for _ in range(world)]
        dist.all_gather(host, padded.cpu(), group=group)
        return torch.cat(host, 0).to(local.device)[:n_total]
    dist.all_gather_into_tensor(out, padded, group=group)
    return out[:n_total]


class PendingGather:
    """Handle of gather_rows_async: wait() makes the CURRENT stream wait for the collective (no host block with
    RCCL) and returns the full (n_total, E) tensor."""

    def __init__(self, out, n_total, work=None):
        self._out, self._n, self._work = out, n_total, work

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._out[:self._n]


def gather_rows_async(local: torch.Tensor, n_total: int, group=None) -> PendingGather:
    """gather_rows that does not put the collective on the critical path: with RCCL the all_gather is issued with
    async_op=True on the process group's own stream (it waits for `local` by event), the caller's stream goes on
    with the next batch and joins at PendingGather.wait().  A job that gathers every batch -- bench.py --gpus N,
    a streaming server -- then overlaps the (latency-bound, few hundred KB) gather of batch k with the forward of
    batch k + 1, and a slow rank delays its peers by at most the batches they keep in flight instead of at every
    step.  (gloo with CUDA tensors -- the one-GPU debug path -- completes the gather before returning.)"""
    if not collectives_active(group) or (dist.get_backend(group) == "gloo" and local.is_cuda):
        return PendingGather(gather_rows(local, n_total, group), n_total)
    world = dist.get_world_size(group)
    per = shard_size(n_total, world)
    width = tuple(local.shape[1:])
    if local.shape[0] == per:
        padded = local.contiguous()
    else:
        padded = torch.zeros((per,) + width, dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    out = torch.empty((world * per,) + width, dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, padded, group=group, async_op=True)
    return PendingGather(out, n_total, work)


def extract_sharded(extract_fn, wavs: torch.Tensor, batch_size: int = 256, group=None):
    """wavs: (N, samples) on every rank (or at least this rank's shard valid).  extract_fn maps a
    (b, samples) block to (b, E) embeddings on this rank's GPU.  Returns (N, E) on every rank."""
    n = wavs.shape[0]
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_range(n, rank, world)
    outs = []
    for b0 in range(lo, hi, batch_size):
        outs.append(extract_fn(wavs[b0:min(hi, b0 + batch_size)]))
    if outs:
        local = torch.cat(outs, 0)
    else:
        probe = extract_fn(wavs[:1])
        local = probe[:0]
    return gather_rows(local, n, group)


def llr_matrix_sharded(block_fn, n_enroll: int, group=None, to_rank0_only: bool = False):
    """The (n_enroll, n_test) PLDA LLR matrix with the ENROLLMENT rows sharded over the ranks (for
    >= 1e8-trial matrices; the reference scores trial by trial on one process, two_cov_plda.py:246-256).

    block_fn(lo, hi) -> (hi - lo, n_test) tensor: this rank's rows -- e.g.
    `lambda lo, hi: plda.llr_matrix(enroll_t[lo:hi], n, test_t)` with the transformed test table replicated on every
    rank (it is what `gather_rows` of the test embeddings returns).  Rank r takes rows shard_range(n_enroll, r, G); one
    all_gather of the equal-size padded blocks assembles the matrix on every rank (to_rank0_only: ranks > 0 get None and
    skip the copy of the full matrix -- the collective is the same).  Rows never mix: every entry is exactly what the
    unsharded call computes."""
    if not collectives_active(group):
        return block_fn(0, n_enroll)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n_enroll, rank, world)
    block = block_fn(lo, hi)
    full = gather_rows(block, n_enroll, group)
    return full if (rank == 0 or not to_rank0_only) else None
