"""ID driver: score matrix -> centred spectrum, per data point (drop-in for /root/reference/dim_reduction.py).

``get_manifold_dimension(config, name=None, return_svd=False)`` keeps the reference's contract (:116-215):
same config keys, same number of processed points (``idx + 1 >= num_datapoints`` stops one short, :159-164),
same per-point row count ``M = (num_batches - 1) * B + extra`` with ``ambient_dim = prod(x.shape[1:])`` of ONE
un-batched sample (:166-171), same output ``{'singular_values': [[...], ...]}`` returned or pickled to
``<log_path>/<log_name>/svd/<name>.pkl`` (:206-211).

What is different, by design for MI355X:
* nothing leaves the GPU inside the loop: the reference moves every score batch to the host (:183) and runs a
  full CPU SVD with U and V (:197); here rows are written straight into the device-resident S [M, D] and the
  spectrum comes from the fp64 Gram -> tridiagonal -> bisection kernels (csrc/spectrum.hip);
* rows that the reference computes and throws away (the tail of the last batch, :185-188) are not computed;
* score evaluations run ``inflight`` rows at a time (default: as many of the point's rows as fit the
  ``dim_estimation.inflight_rows`` budget) instead of B -- every row is an independent sample, GroupNorm is
  per-sample and the model is in eval mode, so the batch boundary is not observable;
* noise comes from an in-kernel Philox stream keyed by ``seed + 1000003 * (point_index + 1)`` and indexed by the
  element's position in the point's noise matrix (csrc/rng.hip), fused with the perturbation, so results depend
  neither on how points are distributed over GPUs nor on the launch-set size; points are sharded round-robin over the ranks of the process group and
  the spectra are combined by one all-gather (parallel.py).
"""
import math
import os
import pickle
from pathlib import Path

import torch

from . import _lib, parallel
from .lightning_data_modules.utils import create_lightning_datamodule
from .lightning_modules.utils import create_lightning_module
from .models import utils as mutils
from .plot_utils import estimate_dim


def batching(sample_shape, batchsize):
    """(num_batches, extra_in_last_batch, rows_in_S) of dim_reduction.py:166-171."""
    ambient_dim = math.prod(sample_shape[1:])
    num_batches = (ambient_dim // batchsize + 1) * 4
    extra = ambient_dim - (ambient_dim // batchsize) * batchsize
    return num_batches, extra, (num_batches - 1) * batchsize + extra


def _num_datapoints(config):
    # dotted hasattr, as the reference does (dim_reduction.py:144-147)
    if hasattr(config, 'dim_estimation.num_datapoints'):
        return config.dim_estimation.num_datapoints
    elif hasattr(config, 'logging.svd_points'):
        return config.logging.svd_points
    raise NameError("num_datapoints: neither dim_estimation.num_datapoints nor logging.svd_points is set")


class ScoreMatrixBuilder:
    """Produces S for one point with all work on the device."""

    def __init__(self, score_fn, sde, sampling_eps, device, inflight_rows=None, concurrent_sets=None):
        self.score_fn, self.sde, self.eps, self.device = score_fn, sde, sampling_eps, device
        self.inflight_rows = inflight_rows
        # OPT-IN (IDIFF_CONCURRENT_SETS=2 / concurrent_sets=2): consecutive launch sets of a point go to two worker streams,
        # so the tail of every launch and the small-map layers of one forward fill up with the other forward's work
        # (two 2240-row NCSN++ forwards 476.6 -> 468 ms in round 3, scripts/two_stream_probe.py).  Same kernels, same bits.  Off by
        # default -- and since the convolutions moved to the one-workgroup-per-CU pair kernel it LOSES: round 5 re-measured
        # bench.py --concurrent-sets 2 at 10.4-14.2k evals/s (erratic: how the two streams' launches interleave) against 16.1k in
        # sequence (profiles/HISTORY_r05.md).  Kept for the parity test of the stream logic only.
        if concurrent_sets is None:
            concurrent_sets = int(os.environ.get("IDIFF_CONCURRENT_SETS", "1"))
        self.concurrent_sets = max(1, int(concurrent_sets))
        self._workers = None
        self._warmed = False

    def rows_per_launch(self, rows, sample_numel, vector=False):
        if self.inflight_rows:
            return max(1, min(rows, int(self.inflight_rows)))
        # default: all rows of a vector point (the k-sphere MLPs); for images 2240 rows of 32x32x3 (measured best of
        # 1120 / 2240 / 4480 on the nf=128 NCSN++: whole numbers of workgroups per CU at every level) scaled by the sample size
        if vector and rows <= 65536:
            return rows
        return min(rows, max(128, (2240 * 3072) // sample_numel))

    def build(self, x, batchsize, t=None, noise=None, generator=None, seed=None, row_range=None, safe=False):
        """x: one sample on the device; returns S [M, D] fp32 (rows in the reference's order), or only the rows
        ``row_range = (lo, hi)`` of it (row-sharded pipeline: with ``seed`` the draws of a row do not depend on who
        computes it).

        ``safe=True``: the same point with every contraction on the range-unlimited route (fp32-contraction Winograd, six-product
        GEMMs: ``IDIFF_NO_WINO43H`` / ``IDIFF_NO_PAIRS`` for this host thread's launches only).  The drivers call it ONCE for a
        point whose score matrix came back non-finite from the default fp16-pair route, before they give up on the point: the
        reference evaluates any checkpoint in fp32 (models/layerspp.py:242-274), so a range limit of ours must not turn into an error
        of the user's.  Position-keyed noise (``seed`` / ``noise``) makes the re-run the same matrix.

        Noise: ``noise`` (explicit draws, for parity tests) > ``seed`` (in-kernel Philox stream, the default of the
        drivers: independent of the launch-set size; for D % 4 != 0 a torch generator keyed by the seed, same property) >
        ``generator`` (torch.randn, consumed launch set by launch set: the caller owns reproducibility)."""
        if safe:
            with _lib.thread_option("IDIFF_NO_WINO43H", 1), _lib.thread_option("IDIFF_NO_PAIRS", 1):
                return self.build(x, batchsize, t=t, noise=noise, generator=generator, seed=seed, row_range=row_range)
        _, _, rows = batching(tuple(x.shape), batchsize)
        D = x.numel()
        t = self.eps if t is None else t
        r_lo, r_hi = (0, rows) if row_range is None else row_range
        if row_range is not None and noise is None and seed is None:
            raise RuntimeError("a row range needs position-keyed noise: pass `seed` or explicit `noise`")
        if noise is None and seed is not None and D % 4:
            # the in-kernel Philox stream writes 16-byte groups (D % 4 == 0).  Other widths draw the point's WHOLE noise
            # matrix from a generator keyed by the point seed and slice it: still a function of (seed, row, column) only,
            # whatever the launch-set size, the row range or the number of ranks
            noise = torch.randn(rows, D, device=self.device, dtype=torch.float32,
                                generator=torch.Generator(device=self.device).manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF))
        S = torch.empty(r_hi - r_lo, D, device=self.device, dtype=torch.float32)
        step = self.rows_per_launch(rows, D, vector=x.ndim == 1)
        xf = x.reshape(-1).contiguous()
        nsets = -(-(r_hi - r_lo) // step)
        if (self.concurrent_sets > 1 and nsets > 1 and torch.device(self.device).type == "cuda" and self._warmed
                and (noise is not None or seed is not None)):      # torch-generator draws stay in sequential order
            return self._build_concurrent(x, xf, S, r_lo, r_hi, step, D, t, noise, seed)
        self._warmed = True     # the first point runs on one stream: filter banks and constants are made lazily, once
        for lo in range(r_lo, r_hi, step):
            n = min(step, r_hi - lo)
            vec_t = torch.full((n,), float(t), device=self.device, dtype=torch.float32)
            mean_unit, std = self.sde.marginal_prob(torch.ones((), device=self.device), vec_t)
            coeff = None if mean_unit.ndim == 0 else mean_unit.reshape(-1).contiguous()
            batch = torch.empty(n, D, device=self.device, dtype=torch.float32)
            if noise is None and seed is not None:
                _lib.perturb_randn(xf, std.contiguous(), coeff, batch, n, D, lo, seed)
            else:
                if noise is not None:
                    z = noise[lo:lo + n].reshape(n, D).contiguous()
                else:
                    z = torch.randn(n, D, device=self.device, dtype=torch.float32, generator=generator)
                _lib.perturb(xf, z, std.contiguous(), coeff, batch, n, D)
            self._score_into(S[lo - r_lo:lo - r_lo + n], batch.view(n, *x.shape), vec_t)
        return S

    def _score_into(self, rows, batch, vec_t):
        """Scores of ``batch`` into ``rows`` (a row block of S): written there by the network's last kernel when the model takes an
        output buffer (the image models), copied otherwise."""
        if getattr(self.score_fn, "accepts_out", False):
            self.score_fn(batch, vec_t, out=rows)
        else:
            rows.copy_(self.score_fn(batch, vec_t).reshape(rows.shape))

    def _build_concurrent(self, x, xf, S, r_lo, r_hi, step, D, t, noise, seed):
        """The launch sets of one point dealt round-robin to ``concurrent_sets`` worker streams; the caller's stream waits
        for all of them before S is handed back."""
        cur = torch.cuda.current_stream()
        if self._workers is None or len(self._workers) != self.concurrent_sets:
            self._workers = [torch.cuda.Stream(device=self.device) for _ in range(self.concurrent_sets)]
        for w in self._workers:
            w.wait_stream(cur)
        for i, lo in enumerate(range(r_lo, r_hi, step)):
            n = min(step, r_hi - lo)
            with torch.cuda.stream(self._workers[i % len(self._workers)]):
                vec_t = torch.full((n,), float(t), device=self.device, dtype=torch.float32)
                mean_unit, std = self.sde.marginal_prob(torch.ones((), device=self.device), vec_t)
                coeff = None if mean_unit.ndim == 0 else mean_unit.reshape(-1).contiguous()
                batch = torch.empty(n, D, device=self.device, dtype=torch.float32)
                if noise is None:
                    _lib.perturb_randn(xf, std.contiguous(), coeff, batch, n, D, lo, seed)
                else:
                    _lib.perturb(xf, noise[lo:lo + n].reshape(n, D).contiguous(), std.contiguous(), coeff, batch, n, D)
                self._score_into(S[lo - r_lo:lo - r_lo + n], batch.view(n, *x.shape), vec_t)
        for w in self._workers:
            cur.wait_stream(w)
        return S


class SpectrumPipeline:
    """Runs the spectrum of point p on a side HIP stream while the score evaluations of point p+1 fill the
    main stream: the tridiagonalisation is a chain of ~2D short bandwidth-bound launches that leaves most CUs
    idle, the convolutions are MFMA-bound -- the two overlap almost for free.

    The spectrum of a point is ENQUEUED one ``submit`` late, i.e. after the next point's score evaluations have been
    queued: its ~6000 launches do not fit the side stream's hardware queue, so the enqueueing host thread blocks until
    the spectrum is nearly done -- and while it did that right after point p, nothing of point p+1 had been queued yet and
    the main stream sat idle for the length of a spectrum at every point boundary (72 ms gaps in the rocprofv3 trace)."""

    def __init__(self, device, overlap=True):
        self.device = device
        # A stream of another priority class gets a hardware queue of its own.  A default-priority stream shares the
        # runtime's small pool of queues with every other stream of the process: once an RCCL communicator exists (its
        # streams were made first) this one landed on the main stream's queue and the spectrum ran BETWEEN the score
        # evaluations instead of beside them (+35 ms per point under torch.distributed.run, same kernels, same durations).
        prio = int(os.environ.get("IDIFF_SIDE_STREAM_PRIORITY", "-1"))
        self.side = torch.cuda.Stream(device=device, priority=prio) if overlap else None
        self.pending = []         # [sv, S or None, pinned failure flag, event after the flag copy, rebuild]: S is held until the flag is read
        self.deferred = None      # (S, ready event) of the last submitted point, not yet enqueued
        self.resolved = 0         # spectra that needed a fallback form of the eigensolver (fail-soft, see _reap)
        self.rebuilt = 0          # points whose score matrix was non-finite on the default route and was built again on the safe one

    def _launch(self, S, ready, rebuild=None):
        stream = self.side if self.side is not None else torch.cuda.current_stream()
        if self.side is not None:
            self.side.wait_event(ready)                  # S is complete on the producing stream
        with torch.cuda.stream(stream):
            sv = _lib.spectrum(S, full=True)
            # the eigensolver reports trouble as NaN; one flag per spectrum goes to pinned host memory so that the check
            # costs no synchronisation: it is read once the event behind the copy has completed
            flag = torch.empty(1, dtype=torch.bool, pin_memory=True)
            flag.copy_(torch.isnan(sv).any().reshape(1), non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        if self.side is not None:
            S.record_stream(self.side)                   # keep S alive until the side stream is done with it
        self.pending.append([sv, S, flag, done, rebuild])

    def _reap(self, block):
        """Reads the failure flags of the spectra that have completed (all of them with ``block``).  A flagged spectrum is
        solved again from its S -- still held for exactly this -- with the eigensolver's fallback forms
        (``_lib.resolve_failed_spectrum``: wavefront chase, then the one-stage sweep) in this process; only a matrix
        that defeats all three, or holds non-finite scores, raises."""
        for entry in self.pending:
            if entry[1] is None:
                continue
            if block:
                entry[3].synchronize()
            elif not entry[3].query():
                break                                    # in stream order: later ones are not done either
            if bool(entry[2][0]):
                stream = self.side if self.side is not None else torch.cuda.current_stream()
                S = entry[1]
                if entry[4] is not None and not bool(torch.isfinite(S).all()):
                    # not the eigensolver: the score matrix itself is non-finite.  One re-run of the point on the range-unlimited
                    # route (ScoreMatrixBuilder.build(safe=True)), on the caller's stream like every build, before that becomes an error
                    warn_non_finite_point()
                    S = entry[4]()
                    self.rebuilt += 1
                    torch.cuda.current_stream().synchronize()   # the rare path: S (and any filter bank packed for it) is complete
                    with torch.cuda.stream(stream):
                        entry[0] = _lib.spectrum(S, full=True)
                with torch.cuda.stream(stream):
                    if bool(torch.isnan(entry[0]).any()):
                        entry[0] = _lib.resolve_failed_spectrum(S, full=True,
                                                                log=None if parallel.rank_world()[0] == 0 else (lambda msg: None))
                        self.resolved += 1
            entry[1] = entry[4] = None

    def submit(self, S, rebuild=None):
        """``rebuild``: a callable returning this point's S again on the safe route (see ``_reap``); without it a non-finite S raises."""
        self._reap(block=False)
        if self.side is None:
            self._launch(S, None, rebuild)
            return
        ready = torch.cuda.Event()
        ready.record()
        previous, self.deferred = self.deferred, (S, ready, rebuild)
        if previous is not None:
            self._launch(*previous)

    def results(self):
        """All submitted spectra, in order, checked; joins the side stream into the current one."""
        if self.side is not None and self.deferred is not None:
            self._launch(*self.deferred)
            self.deferred = None
        self._reap(block=True)
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        out, self.pending = [e[0] for e in self.pending], []
        return out


def warn_non_finite_point():
    import warnings
    if parallel.rank_world()[0] == 0:
        warnings.warn("id-diff_amd: a point's score matrix came back non-finite from the default (fp16-pair) route; building the "
                      "point again with every contraction on the fp32 route (IDIFF_NO_WINO43H / IDIFF_NO_PAIRS for this re-run)")


def row_sharded_spectrum(S_local, total_rows, ops=None, block_rows=None, rebuild=None):
    """Singular values (descending, fp32, [D]) of the column-centred [total_rows, D] matrix whose rows are spread over
    the ranks of the default process group (this rank holds ``S_local``); every rank returns the full spectrum.

    SURVEY.md 8(f) rank 2 / DESIGN.md 6: two collectives -- all-reduce of the fp64 column sums [D], all-reduce of the
    fp64 centred Gram [D, D] -- then the eigensolve, run redundantly on every rank (cheaper than shipping the result).
    The Gram (75 MB at D = 3072, 1.2 GB at D = 12288: bandwidth-bound on the xGMI ring) goes in blocks of ``block_rows``
    rows of its upper triangle, each packed from its first diagonal tile onwards (0.6 GB in total at D = 12288): the
    all-reduce of block b is asynchronous and runs while the matrix cores compute block b + 1; the lower triangle is
    mirrored once at the end.  With one rank this is the same arithmetic as
    ``_lib.spectrum``.  ``ops`` = (column_sums, gram_rows, symmetrize, sym_eigvals) defaults to the HIP stages; the CPU
    process-group test injects plain-torch stand-ins to check the reduction logic without a GPU.  ``rebuild``: a callable that returns
    this rank's rows again on the safe route; used once, on every rank, when any rank's Gram comes out non-finite."""
    col_sums, gram_rows, symmetrize, eigvals = ops if ops is not None else (
        _lib.column_sums, _lib.centered_gram_rows, _lib.symmetrize_upper, _lib.sym_eigvals)
    D = S_local.shape[1]
    if total_rows < D:
        raise RuntimeError(f"row_sharded_spectrum: needs total_rows >= D (got {total_rows} x {D})")
    sums = parallel.all_reduce_sum(col_sums(S_local))
    mean = sums / float(total_rows)
    if block_rows is None:
        block_rows = max(64, ((D // 8 + 63) // 64) * 64)            # ~8 blocks: enough to hide all but the first Gram block
    grouped = parallel.is_grouped()

    def reduced_gram():
        G = torch.zeros(D, D, dtype=torch.float64, device=S_local.device)
        pending = []
        for r0 in range(0, D, block_rows):
            r1 = min(D, r0 + block_rows)
            gram_rows(S_local, mean, G, r0, r1)
            if not grouped:
                continue
            # only the columns from the block's first diagonal tile onwards carry data (the rest is mirrored afterwards):
            # reduce that trapezoid, packed, not whole rows -- about half the bytes of a bandwidth-bound collective
            c0 = (r0 // 128) * 128
            staging = G[r0:r1, c0:].contiguous()
            pending.append((parallel.all_reduce_sum_async(staging), staging, r0, r1, c0))
        for work, staging, r0, r1, c0 in pending:
            parallel.wait(work)
            G[r0:r1, c0:].copy_(staging)
        return symmetrize(G)

    eig = eigvals(reduced_gram())                                     # the eigensolver overwrites its input
    if ops is None:
        # fail soft.  The failure flag is reduced (MAX) over the ranks so that EVERY rank takes the same branch even when
        # only one device's eigensolve failed (a stalled chase is a property of one device, not of G); no copy of G is
        # kept for this rare path -- it is rebuilt (1.2 GB of fp64 at D = 12288 otherwise sat beside G on every call).
        failed = torch.isnan(eig).any().to(torch.int32).reshape(1)
        if grouped:
            torch.distributed.all_reduce(failed, op=torch.distributed.ReduceOp.MAX)
        if bool(failed.item()):
            if rebuild is not None:
                bad = (~torch.isfinite(S_local).all()).to(torch.int32).reshape(1)
                if grouped:
                    torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX)
                if bool(bad.item()):                         # the same branch on every rank: the flag is reduced
                    warn_non_finite_point()
                    return row_sharded_spectrum(rebuild(), total_rows, ops=ops, block_rows=block_rows, rebuild=None)
            eig = _resolve_failed_eigvals(reduced_gram, eigvals, grouped)
    return eig.clamp_min(0.0).sqrt().flip(0).to(torch.float32)


def _resolve_failed_eigvals(make_gram, eigvals, grouped=False):
    """The slower eigensolver forms in turn, on a freshly built Gram matrix each (the solver overwrites it); with a process
    group every rank runs the same form and the per-form failure flag is reduced, so all ranks return the same spectrum."""
    import warnings
    G = make_gram()
    if not bool(torch.isfinite(G).all()):
        raise RuntimeError("the Gram matrix holds non-finite values (NaN / inf score vectors): no spectrum exists")
    for i, (name, what) in enumerate(_lib._FALLBACKS):
        with _lib.thread_option(name, 1):                  # this thread's launches only
            eig = eigvals(G if i == 0 else make_gram())
        failed = torch.isnan(eig).any().to(torch.int32).reshape(1)
        if grouped:
            torch.distributed.all_reduce(failed, op=torch.distributed.ReduceOp.MAX)
        if not bool(failed.item()):
            if parallel.rank_world()[0] == 0:
                warnings.warn(f"id-diff_amd: the two-stage eigensolver reported a failure; re-solved with {what} ({name})")
            return eig
    raise RuntimeError("the eigensolver reported a failure (NaN eigenvalues) in all of its three forms")


def checked_spectra(spectra):
    """Host copy of gathered spectra.  The eigensolver poisons its output with NaN instead of returning a wrong spectrum
    (band-reduction residual above tolerance, a stalled chase: include/idiff_hip.h, idiff_symtridiag_f64); this is where
    the values first reach the host, so this is where that turns into an exception."""
    host = spectra.cpu()
    if bool(torch.isnan(host).any()):
        raise RuntimeError("NaN singular values reached the host: a spectrum that did not come through SpectrumPipeline / "
                           "row_sharded_spectrum (which re-solve a failed eigensolve in-process) reported a failure")
    return host


def build_many(builder, xs, batchsize, seeds):
    """S [P, M, D] for P small (vector) points with ONE score_fn call over all P*M rows: the k-sphere workload is
    launch-bound one point at a time (M = 1501 rows of a 7-layer MLP), so points are batched (BASELINE config 2)."""
    x0 = xs[0]
    _, _, rows = batching(tuple(x0.shape), batchsize)
    D, P, dev = x0.numel(), len(xs), builder.device
    vec_t = torch.full((rows,), float(builder.eps), device=dev, dtype=torch.float32)
    mean_unit, std = builder.sde.marginal_prob(torch.ones((), device=dev), vec_t)
    coeff = None if mean_unit.ndim == 0 else mean_unit.reshape(-1).contiguous()
    std = std.contiguous()
    batch = torch.empty(P, rows, D, device=dev, dtype=torch.float32)
    for i, (x, seed) in enumerate(zip(xs, seeds)):
        if D % 4 == 0:
            _lib.perturb_randn(x.reshape(-1).contiguous(), std, coeff, batch[i], rows, D, 0, seed)
        else:
            z = torch.randn(rows, D, device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(seed))
            _lib.perturb(x.reshape(-1).contiguous(), z, std, coeff, batch[i], rows, D)
    t_all = torch.full((P * rows,), float(builder.eps), device=dev, dtype=torch.float32)
    score = builder.score_fn(batch.view(P * rows, *x0.shape), t_all)
    return score.reshape(P, rows, D)


def setup_model(config):
    """Steps :123-141 of the reference: data module, module + checkpoint, SDE, device, score_fn."""
    DataModule = create_lightning_datamodule(config)
    DataModule.setup()
    pl_module = create_lightning_module(config)
    pl_module = pl_module.load_from_checkpoint(config.model.checkpoint_path)
    pl_module.configure_sde(config)
    device = torch.device(config.device)
    if device.type != "cuda":
        raise RuntimeError(f"config.device is {device}: id-diff_amd runs the manifold_dimension path on the "
                           "MI355X only (the CPU restatement lives in oracle/ and is test infrastructure)")
    pl_module = pl_module.to(device)
    pl_module.eval()
    score_fn = mutils.get_score_fn(pl_module.sde, pl_module.score_model, conditional=False, train=False, continuous=True)
    return DataModule, pl_module, score_fn, device


def collect_points(loader, num_datapoints):
    """The points the reference's double loop would visit, in order (dim_reduction.py:154-164)."""
    pts, idx = [], 0
    for orig_batch in loader:
        if isinstance(orig_batch, (list, tuple)):
            orig_batch = orig_batch[0]
        batchsize = orig_batch.size(0)
        if idx + 1 >= num_datapoints:
            break
        for x in orig_batch:
            if idx + 1 >= num_datapoints:
                break
            pts.append((x, batchsize))
            idx += 1
    return pts


def get_manifold_dimension(config, name=None, return_svd=False, return_dims=False):
    """Drop-in for dim_reduction.py:116-215.  ``return_dims=True`` (not in the reference; needs ``return_svd``) also
    returns the integer ID of every point, computed by the reference's rule on the rank that owns the point and gathered
    as int32 in the same collective as the spectra."""
    log_path, log_name = config.logging.log_path, config.logging.log_name
    save_path = os.path.join(log_path, log_name, 'svd')
    rank, world = parallel.rank_world()
    if rank == 0 and not return_svd:
        Path(save_path).mkdir(parents=True, exist_ok=True)

    seed = int(config.get('seed', 42))
    torch.manual_seed(seed)  # same data split / loader order on every rank
    DataModule, pl_module, score_fn, device = setup_model(config)
    num_datapoints = _num_datapoints(config)
    points = collect_points(DataModule.train_dataloader(), num_datapoints)
    if not points:                       # num_datapoints <= 1: the reference's loop body never runs (:159-164)
        info = {'singular_values': []}
        if return_svd:
            return info
        if rank == 0:
            with open(os.path.join(save_path, f'{name}.pkl'), 'wb') as f:
                pickle.dump(info, f)
        return None
    builder = ScoreMatrixBuilder(score_fn, pl_module.sde, pl_module.sampling_eps, device,
                                 config.get('dim_estimation.inflight_rows', None))
    def point_seed(p):
        return seed + 1000003 * (p + 1)

    if str(config.get('dim_estimation.shard', 'points')) == 'rows' and world > 1:
        # one point at a time on ALL ranks, each computing a slice of its rows (latency of a single large point)
        spectra = []
        with torch.no_grad():
            for p, (x, batchsize) in enumerate(points):
                rows = batching(tuple(x.shape), batchsize)[2]
                kw = dict(seed=point_seed(p), row_range=parallel.my_rows(rows, rank, world))
                S_local = builder.build(x.to(device), batchsize, **kw)
                spectra.append(row_sharded_spectrum(S_local, rows, rebuild=lambda x=x, b=batchsize, kw=kw: builder.build(x.to(device), b, safe=True, **kw)))
        info = {'singular_values': [s.tolist() for s in checked_spectra(torch.stack(spectra))]}
        if return_svd:
            return info
        if rank == 0:
            with open(os.path.join(save_path, f'{name}.pkl'), 'wb') as f:
                pickle.dump(info, f)
        return None

    mine = parallel.my_points(len(points), rank, world)
    n_sv = None
    pipe = SpectrumPipeline(device, overlap=bool(config.get('dim_estimation.overlap_spectrum', True)))

    with torch.no_grad():
        small = bool(mine) and points[mine[0]][0].numel() <= 4096 and len({points[p][1] for p in mine}) == 1
        if small:
            # vector data: many points per launch set, batched spectrum kernel (one workgroup per matrix for D <= 128)
            rows = batching(tuple(points[mine[0]][0].shape), points[mine[0]][1])[2]
            group = max(1, min(len(mine), int(config.get('dim_estimation.points_per_launch', max(1, 131072 // rows)))))
            for lo in range(0, len(mine), group):
                ids = mine[lo:lo + group]
                S = build_many(builder, [points[p][0].to(device) for p in ids], points[ids[0]][1],
                               [point_seed(p) for p in ids])
                pipe.submit(S)
            local = [sv for block in pipe.results() for sv in block]
        else:
            for p in mine:
                x, batchsize = points[p]
                pipe.submit(builder.build(x.to(device), batchsize, seed=point_seed(p)),
                            rebuild=lambda x=x, b=batchsize, sd=point_seed(p): builder.build(x.to(device), b, seed=sd, safe=True))
            local = pipe.results()
    n_sv = points[0][0].numel()              # fixed-width rows for the exchange (a rank without points takes part too)
    local = torch.stack(local) if local else torch.empty(0, n_sv, device=device)
    # torch.linalg.svd returns min(M, D) values per point (dim_reduction.py:197): a short loader batch gives a shorter list
    keep = [min(batching(tuple(x.shape), b)[2], x.numel()) for x, b in points]
    # (the rule needs three singular values; -1 marks a point that has fewer)
    my_dims = [estimate_dim(sv[:keep[p]].tolist()) if keep[p] >= 3 else -1 for sv, p in zip(checked_spectra(local), mine)]
    spectra, dims = parallel.gather_spectra(local, len(points), n_sv, device, dims=my_dims)
    info = {'singular_values': [s[:k].tolist() for s, k in zip(checked_spectra(spectra), keep)]}
    if return_svd:
        return (info, dims.tolist()) if return_dims else info
    if rank == 0:
        with open(os.path.join(save_path, f'{name}.pkl'), 'wb') as f:
            pickle.dump(info, f)


def collect_labelled_points(loader, num_datapoints, label=1):
    """The (x, y, loader batch size) triples the reference's conditional loop visits (dim_reduction.py:49-60): items
    whose label equals 1, ``num_datapoints - 1`` of them."""
    pts, idx = [], 0
    for orig_batch, orig_labels in loader:
        batchsize = orig_batch.size(0)
        if idx + 1 >= num_datapoints:
            break
        for x, y in zip(orig_batch, orig_labels):
            if y.item() != label:
                continue
            if idx + 1 >= num_datapoints:
                break
            pts.append((x, y.item(), batchsize))
            idx += 1
    return pts


def conditional_spectra(builder, loader, num_datapoints, seed=42, levels=None, noise=None, overlap=True):
    """Spectra of the label-1 validation points at the noise levels ``linspace(sampling_eps, 0.3, 12)``
    (dim_reduction.py:39-101).  Returns one dict per level: ``{'level', 't', 'images', 'singular_values', 'labels'}``.

    MI355X specifics: per level the points are sharded round-robin over the ranks and their spectra combined by one
    all-gather (parallel.py); the spectrum of point p runs on the side stream under the score evaluations of point
    p+1 (``SpectrumPipeline``); nothing is synchronised with the host until a level is complete.  The Philox stream
    of (level, point) is keyed by ``seed + 1000003 * (point + 1) + 7919 * level``: levels do not share draws.
    ``levels`` restricts the sweep to some of the 12 indices and ``noise(level, point) -> [rows, *x.shape]`` supplies
    explicit draws (parity tests); the drop-in entry point below uses neither."""
    rank, world = parallel.rank_world()
    device = builder.device
    times = torch.linspace(builder.eps, 0.3, 12)
    out = []
    for level, t_slice in enumerate(times):
        if levels is not None and level not in levels:
            continue
        points = collect_labelled_points(loader, num_datapoints)
        mine = parallel.my_points(len(points), rank, world)
        pipe = SpectrumPipeline(device, overlap=overlap)
        with torch.no_grad():
            for p in mine:
                x, _, batchsize = points[p]
                z = None if noise is None else noise(level, p).to(device)
                kw = dict(t=float(t_slice), noise=z, seed=seed + 1000003 * (p + 1) + 7919 * level)
                pipe.submit(builder.build(x.to(device), batchsize, **kw),
                            rebuild=lambda x=x, b=batchsize, kw=kw: builder.build(x.to(device), b, safe=True, **kw))
            local = pipe.results()
        if points:
            n_sv = points[0][0].numel()
            local = torch.stack(local) if local else torch.empty(0, n_sv, device=device)
            spectra = checked_spectra(parallel.gather_spectra(local, len(points), n_sv, device))
            spectra = [s[:min(batching(tuple(x.shape), b)[2], x.numel())] for s, (x, _, b) in zip(spectra, points)]
        else:
            spectra = torch.empty(0, 0)
        out.append({'level': level, 't': float(t_slice),
                    'images': torch.stack([x.permute(1, 2, 0) for x, _, _ in points]).numpy() if points else [],
                    'singular_values': [s.tolist() for s in spectra], 'labels': [y for _, y, _ in points]})
    return out


def get_conditional_manifold_dimension(config, name=None):
    """dim_reduction.py:12-114: the same estimator at 12 noise levels linspace(sampling_eps, 0.3, 12) on the
    label==1 points of the validation loader; writes images.pkl / labels_svd.pkl / labels.pkl per level
    (rank 0 only when run under torch.distributed)."""
    log_path, log_name = config.logging.log_path, config.logging.log_name
    config.data.return_labels = True
    seed = int(config.get('seed', 42))
    torch.manual_seed(seed)
    DataModule, pl_module, score_fn, device = setup_model(config)
    num_datapoints = config.get('dim_estimation.num_datapoints', 26)
    builder = ScoreMatrixBuilder(score_fn, pl_module.sde, pl_module.sampling_eps, device,
                                 config.get('dim_estimation.inflight_rows', None))
    loader = DataModule.val_dataloader()
    results = conditional_spectra(builder, loader, num_datapoints, seed=seed,
                                  overlap=bool(config.get('dim_estimation.overlap_spectrum', True)))
    if parallel.rank_world()[0] != 0:
        return
    for lv in results:
        t_save_path = os.path.join(log_path, log_name, 'svd', '%.3f' % lv['t'])
        Path(t_save_path).mkdir(parents=True, exist_ok=True)
        for fname, payload in (('images.pkl', {'images': lv['images']}),
                               ('labels_svd.pkl', {'singular_values': lv['singular_values']}),
                               ('labels.pkl', {'labels': lv['labels']})):
            with open(os.path.join(t_save_path, fname), 'wb') as f:
                pickle.dump(payload, f)
