"""world_size-2 test of the sharding + all-gather exchange on CPU (gloo); the GPU path uses the same code on RCCL."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, num_points, n_sv, q):
    sys.path.insert(0, ROOT)
    import id_diff_amd  # noqa: F401
    from id_diff_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    mine = parallel.my_points(num_points, r, w)
    local = torch.stack([torch.arange(n_sv, dtype=torch.float32) + 100.0 * p for p in mine]) if mine \
        else torch.empty(0, n_sv)
    out = parallel.gather_spectra(local, num_points, n_sv, torch.device("cpu"))
    # the same exchange with the per-point integer IDs riding along as int32 bit patterns
    out2, dims = parallel.gather_spectra(local, num_points, n_sv, torch.device("cpu"), dims=[7 * p + 3 for p in mine])
    assert torch.equal(out, out2) and dims.dtype == torch.int32 and dims.tolist() == [7 * p + 3 for p in range(num_points)]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, num_points, n_sv=7):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_points, n_sv, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = torch.stack([torch.arange(n_sv, dtype=torch.float32) + 100.0 * p for p in range(num_points)])
    for r in range(world):
        assert torch.equal(results[r], expect), (r, results[r])


def test_gather_two_ranks_even():
    _run(2, 4)


def test_gather_two_ranks_ragged_and_idle_rank():
    _run(2, 3)   # rank 1 owns one point fewer
    _run(2, 1)   # rank 1 owns nothing but still joins the collective


def test_gather_eight_ranks_like_the_scaling_run():
    """The driver's scaling run is 1 / 2 / 4 / 8 ranks of one node; no 8-GPU box exists in the build loop, so the 8-rank
    exchange (round-robin ownership, ragged tail, idle ranks, IDs riding in the same collective) is rehearsed on gloo."""
    _run(8, 19)  # ranks 0-2 own three points, ranks 3-7 two
    _run(8, 5)   # ranks 5-7 own nothing


# ---- row-sharded single point: the two all-reduces, with plain-torch stand-ins for the HIP stages
def _torch_ops():
    col_sums = lambda S: S.double().sum(0)

    def gram_rows(S, mean, G, r0, r1):            # upper-triangle rows [r0, r1), nothing left of the block's diagonal tile
        c = S.double() - mean
        G[r0:r1, r0:] = c[:, r0:r1].T @ c[:, r0:]
        return G

    def symmetrize(G):
        G.copy_(torch.triu(G) + torch.triu(G, 1).T)
        return G

    eigvals = lambda G: torch.linalg.eigvalsh(G)
    return col_sums, gram_rows, symmetrize, eigvals


def _rows_worker(rank, world, port, M, D, q):
    sys.path.insert(0, ROOT)
    import id_diff_amd  # noqa: F401
    from id_diff_amd import dim_reduction, parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env(backend="gloo")
    S = torch.randn(M, D, generator=torch.Generator().manual_seed(5)) + 3.0      # every rank draws the same matrix
    lo, hi = parallel.my_rows(M, rank, world)
    sv = dim_reduction.row_sharded_spectrum(S[lo:hi].contiguous(), M, ops=_torch_ops(), block_rows=5)   # 3 blocks, async
    q.put((rank, (lo, hi), sv))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_spectrum_two_and_three_ranks():
    M, D = 37, 12
    S = torch.randn(M, D, generator=torch.Generator().manual_seed(5)) + 3.0
    ref = torch.linalg.svdvals((S - S.mean(0)).double()).float()
    for world in (2, 3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 31500 + (os.getpid() % 2000) + world
        procs = [ctx.Process(target=_rows_worker, args=(r, world, port, M, D, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=120) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        ranges = sorted(g[1] for g in got)
        assert ranges[0][0] == 0 and ranges[-1][1] == M and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        for _, _, sv in got:
            torch.testing.assert_close(sv, ref, rtol=1e-5, atol=1e-6)


# ---- bench.py's own control flow (barrier / timed region / exchange / JSON line) at world size 2, stand-in workload
def _bench_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bench

    class FakeWork:
        """12 'rows' per point, spectra of 5 values that encode (rank, point) so the exchange can be checked."""
        rows, D, B = 12, 5, 4

        def __init__(self, args, rank, dev):
            self.rank, self.out, self.cfg, self.last_S = rank, [], None, None

        def point(self, i):
            self.out.append(torch.tensor([50., 40., 30., 2., 1.]) + 0.01 * i + 0.001 * self.rank)

        def collect(self):
            out, self.out = torch.stack(self.out), []
            return out

    line = bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--device", "cpu"], workload_factory=FakeWork)
    q.put((rank, line))


def test_bench_main_eight_ranks_gloo():
    """bench.main at the world size of the driver's largest scaling leg (CPU / gloo, stand-in workload): max-over-ranks
    timing, one JSON line from rank 0, value = 8 ranks x steps x rows / time, every rank's IDs in the line."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bench_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(8))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(got[r] is None for r in range(1, 8))
    line = got[0]
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["process_group"] == "gloo"
    assert line["value"] == pytest.approx(8 * 3 * 12 / (line["ms_per_step"] * 3e-3), rel=1e-6)
    assert line["id_estimates_all_ranks"] == [2] * 24


def test_bench_main_two_ranks_gloo(capfd):
    """The driver launches bench.py with one rank per GPU; here its main() runs at world size 2 on CPU/gloo with a
    stand-in workload: rank 0 prints the one JSON line, `value` counts both ranks' points, the spectra went through
    parallel.gather_spectra (round-robin layout) and the process group is torn down."""
    import json
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 27500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] is None
    line = got[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["process_group"] == "gloo" and line["config"]["rows_per_point"] == 12
    assert line["value"] == pytest.approx(2 * 3 * 12 / (line["ms_per_step"] * 3e-3), rel=1e-6)
    assert line["id_estimates"] == [2, 2, 2]          # rank 0's three timed points: the cliff 30 -> 2 leaves two small values
    assert line["id_estimates_all_ranks"] == [2] * 6  # every rank's IDs, computed where the spectrum lives, one collective
    printed = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert len(printed) == 1 and json.loads(printed[0])["metric"] == line["metric"]


# ---- `python bench.py --gpus N` as a PLAIN process: the file is its own launcher (no torch.distributed.run around it)
def _plain(args, env_extra=None, timeout=300):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, cwd=ROOT, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_plain_process_starts_its_own_two_ranks():
    """No launcher, no RANK in the environment: `python bench.py --gpus 2 --device cpu` must start two rank processes itself
    and print ONE line with n_gpus 2 (round 3 printed n_gpus 1 from a single rank without an error)."""
    import json
    r = _plain(["--gpus", "2", "--device", "cpu", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["config"]["process_group"] == "gloo"
    assert line["config"]["rank_devices"] == ["cpu", "cpu"] and "self-launched" in line["config"]["launched_by"]
    assert line["id_estimates_all_ranks"] == [2] * 6 and "rehearsal" in line["data"]
    assert line["value"] == pytest.approx(2 * 3 * 12 / (line["ms_per_step"] * 3e-3), rel=1e-6)


def test_bench_world_size_mismatch_is_an_error():
    """A launcher that started another world than --gpus asks for is an error on every rank, not a warning."""
    r = _plain(["--gpus", "2", "--device", "cpu", "--steps", "1", "--warmup", "0"],
               {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29411"})
    assert r.returncode != 0 and "--gpus 2 but the launcher started WORLD_SIZE 1" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_more_gpus_than_visible_without_touching_one():
    """`--gpus 9` on any box of this pool (at most 8 devices; none here): loud error before any rank process starts."""
    r = _plain(["--gpus", "9", "--steps", "1"])
    assert r.returncode != 0 and "9 devices needed" in r.stderr and "visible" in r.stderr
    assert not r.stdout.strip()


def test_a_dead_rank_ends_the_launch_with_its_code(tmp_path):
    """launch_local_ranks: a rank that exits non-zero ends the job (its peers, waiting in a collective, are terminated) and
    the launcher returns that code."""
    sys.path.insert(0, ROOT)
    import id_diff_amd  # noqa: F401
    from id_diff_amd import parallel
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "print('hello from', r, os.environ['WORLD_SIZE'], os.environ['MASTER_PORT'], flush=True)\n"
                      "if r == 1:\n    sys.exit(7)\n"
                      "time.sleep(60)\n")
    import time
    t0 = time.monotonic()
    rc = parallel.launch_local_ranks(str(script), [], 3, need_devices=False)
    assert rc == 7 and time.monotonic() - t0 < 30
