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
import socket
import subprocess
import sys
import threading

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
        rdzv = os.environ.get("IDIFF_RDZV_FILE")     # set by launch_local_ranks only: no port to race for
        if rdzv:
            kwargs["init_method"] = "file://" + rdzv
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def launched():
    """True when a launcher (torch.distributed.run, or ``launch_local_ranks`` below) set this process's rank."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def visible_devices():
    """Device ordinals this process can open, WITHOUT initialising the GPU (``device_count`` only enumerates on this image;
    ``is_available`` / any other ``torch.cuda`` call would make the process unfit to start rank processes)."""
    return torch.cuda.device_count()


def shares_card():
    """IDIFF_DIST_BACKEND=gloo on a GPU box: the rehearsal mode in which several ranks share the visible card(s)."""
    return os.environ.get("IDIFF_DIST_BACKEND") == "gloo"


def device_ordinal(local_rank):
    """The device of this rank: its LOCAL_RANK -- wrapped over the visible cards only in the card-sharing rehearsal mode."""
    n = visible_devices()
    return local_rank % n if (shares_card() and n) else local_rank


def check_world(requested, world, need_devices=True):
    """A world that is not the one asked for is an error, never a warning: a scaling run whose ``--gpus 8`` leg silently ran
    one rank would report a one-GPU number under an eight-GPU label.  With ``need_devices`` this rank's device ordinal
    (LOCAL_RANK) must exist -- except under IDIFF_DIST_BACKEND=gloo, the rehearsal mode in which several ranks share one card."""
    if world != requested:
        raise SystemExit(f"error: --gpus {requested} but the launcher started WORLD_SIZE {world} rank(s)")
    local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    if need_devices and not shares_card() and visible_devices() <= local_rank:
        raise SystemExit(f"error: {int(os.environ.get('LOCAL_WORLD_SIZE', world))} devices needed, {visible_devices()} visible "
                         f"(LOCAL_RANK {local_rank} has no device)")


def launch_local_ranks(script, argv, n, need_devices=True, timeout=None):
    """Start ``n`` fresh rank processes of ``script`` on this node (one per GPU) and relay rank 0's stdout.

    Called by a parent that has NOT touched the GPU (no HIP call, no ``torch.cuda.is_available()``): the children are new
    interpreters (``subprocess.Popen`` -- never ``os.exec*``, never a fork of an initialised process) with RANK / LOCAL_RANK /
    WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, i.e. exactly what ``torch.distributed.run`` would have
    set, so the child takes the launcher path of ``init_from_env``.  Rank 0's stdout is passed through line by line (the one
    JSON line of bench.py), the other ranks' stdout goes to stderr with a rank prefix.  Returns the first non-zero exit code
    (the remaining ranks are terminated then), else 0."""
    if n < 1:
        raise SystemExit(f"error: --gpus {n}")
    if need_devices and visible_devices() < (1 if shares_card() else n):
        raise SystemExit(f"error: {n} devices needed, {visible_devices()} visible")
    with socket.socket() as s:                       # a free port on the loop-back interface (kept in the environment for tools
        s.bind(("127.0.0.1", 0))                      # that read MASTER_PORT; the ranks themselves meet through a file, below)
        port = s.getsockname()[1]
    # The ranks of a self-launched job rendezvous through a FILE store in a directory of this launch (init_from_env): a port picked
    # by bind-close-reuse can be taken by another job of the node between the close and rank 0's listen
    import tempfile
    rdzv_dir = tempfile.mkdtemp(prefix="idiff_rdzv_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), IDIFF_SELF_LAUNCHED="1",
                   IDIFF_RDZV_FILE=os.path.join(rdzv_dir, "store"))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=env, stdout=subprocess.PIPE, text=True))

    def relay(r, pipe):
        for ln in pipe:
            if r == 0:
                sys.stdout.write(ln)
                sys.stdout.flush()
            else:
                sys.stderr.write(f"[rank {r}] {ln}")
        pipe.close()

    threads = [threading.Thread(target=relay, args=(r, p.stdout), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    rc, pending = 0, set(range(n))
    import time
    t_end = None if timeout is None else time.monotonic() + timeout
    while pending and rc == 0:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    rc = code
                    print(f"error: rank {r} exited with code {code}", file=sys.stderr)
                    break
        if t_end is not None and time.monotonic() > t_end:
            rc = 124
            print(f"error: ranks {sorted(pending)} still running after {timeout} s", file=sys.stderr)
        if pending and rc == 0:
            time.sleep(0.05)
    for r in pending:                                  # a rank died: its peers would wait in a collective for ever
        if procs[r].poll() is None:
            procs[r].terminate()
    for r in pending:
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()                            # the exact PID this function started, nothing else
            procs[r].wait()
    for t in threads:
        t.join(timeout=5)
    import shutil
    shutil.rmtree(rdzv_dir, ignore_errors=True)
    return rc


def rank_devices(device):
    """``[(rank, "cuda:3"), ...]`` of every rank (all-gathered), so the bench line shows which ordinal each rank drove."""
    me = str(device)
    if not (dist.is_available() and dist.is_initialized()):
        return [me]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, me)
    return out


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
