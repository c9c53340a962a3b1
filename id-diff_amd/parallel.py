"""Sharding of data points over the GPUs of one node and the single exchange step.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the
tests).  Data points are independent (dim_reduction.py:162-202 appends independent lists), so point ``p`` goes
to rank ``p % world`` and the only collective on the path is one all-gather of the per-rank singular spectra
(``[points_per_rank, n_sv]`` fp32, a few MB at most: latency-bound on the xGMI mesh).  The reference has no
multi-device path; this file is new work scoped by SURVEY.md 8(e).

Second, optional axis: the rows of ONE point's score matrix are split over the ranks
(``my_rows``) and the spectrum is assembled from an all-reduce of the fp64 column sums ([D]) and of the fp64
centred Gram ([D, D]: 75 MB at D = 3072, 1.2 GB at D = 12288, bandwidth-bound on the xGMI ring) --
``dim_reduction.row_sharded_spectrum``.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* if a launcher set them.

    A ``torch.distributed.run`` launch with ONE rank initialises the group too (RCCL on a GPU box), so that the
    single-GPU leg of a scaling run goes through the same communicator set-up and collectives as the N-GPU legs.
    Must be called before anything else touches the GPU: the device is selected first, then RCCL binds to it."""
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return 0, 1, 0
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    if not dist.is_initialized():
        if backend is None:
            # IDIFF_DIST_BACKEND=gloo: several ranks sharing ONE card (tests on a one-GPU box; RCCL wants a GPU per rank)
            backend = os.environ.get("IDIFF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local_rank)    # eager communicator on THIS device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def is_grouped():
    """True when a default process group exists -- also at world size 1, where the collectives still go through the
    backend (a one-rank RCCL communicator on a one-GPU box runs the same code path as the N-GPU legs)."""
    return dist.is_available() and dist.is_initialized()


def my_points(num_points, rank, world):
    """Indices of the points this rank owns: round-robin, so results do not depend on the world size."""
    return list(range(rank, num_points, world))


def gather_spectra(local, num_points, n_sv, device, dims=None):
    """All-gather per-rank spectra into point order: ``[num_points, n_sv]`` fp32 on every rank.

    ``local``: [len(my_points), n_sv] fp32 on ``device``.  Ranks may own one point fewer than rank 0; rows are
    padded to ``ceil(num_points / world)`` so a single fixed-size all-gather suffices.  ``dims`` (optional): this rank's
    integer IDs, one per local point (``[P/W]`` int32 beside the spectra); they ride in the same
    collective as one extra column holding the int32 bit patterns, and the call returns ``(spectra, dims)`` with ``dims``
    an int32 tensor [num_points] in point order.

    Point p = (rank p % world, its step p // world), so the gathered ``[world, per, n]`` block is put into point order by one
    transposed copy on the device -- no per-rank host loop."""
    rank, world = rank_world()
    grouped = dist.is_available() and dist.is_initialized()
    if dims is not None:
        dims = torch.as_tensor(dims, dtype=torch.int32).reshape(-1)
        if dims.numel() != local.shape[0]:
            raise ValueError(f"gather_spectra: {dims.numel()} dims for {local.shape[0]} local spectra")
    if not grouped:
        return local if dims is None else (local, dims.to(device))
    per = (num_points + world - 1) // world
    width = n_sv + (1 if dims is not None else 0)
    send = torch.zeros(per, width, dtype=torch.float32, device=device)
    if local.numel():
        send[: local.shape[0], :n_sv] = local
        if dims is not None:
            send[: local.shape[0], n_sv] = dims.to(device).view(torch.float32)
    recv = torch.empty(world * per, width, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(recv, send)
    out = recv.view(world, per, width).transpose(0, 1).reshape(per * world, width)[:num_points]
    if dims is None:
        return out.contiguous()
    return out[:, :n_sv].contiguous(), out[:, n_sv].contiguous().view(torch.int32)


def my_rows(total_rows, rank, world):
    """Contiguous row range [lo, hi) of one point's score matrix owned by ``rank`` (sizes differ by at most one)."""
    base, extra = divmod(total_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_reduce_sum(tensor):
    """In-place sum over the default process group; identity when there is none."""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def all_reduce_sum_async(tensor):
    """Start an in-place sum over the default process group; returns a handle for ``wait`` (None without a group).

    On RCCL the collective runs on the communicator's own stream after the work already queued on the current stream
    (``tensor`` is complete when it starts), so the caller can go on queueing compute: the two overlap."""
    if dist.is_available() and dist.is_initialized():
        return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, async_op=True)
    return None


def wait(work):
    """Make the current stream (GPU) / the caller (CPU) wait for an asynchronous collective."""
    if work is not None:
        work.wait()
