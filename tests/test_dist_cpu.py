"""The N>1 path on CPU: two gloo ranks shard a page list, each processes its block (the CPU oracle stands
in for the device here — it is the checker, and this test is about the sharding/collective plumbing),
and the union equals the single-process result byte for byte."""
import hashlib
import os
import socket

import numpy as np
import pytest


def test_page_range_partitions_exactly():
    from prlib_amd.dist import page_range

    for n in (0, 1, 7, 256, 1024, 1000):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                seen += list(page_range(n, world, r))
            assert seen == list(range(n))
            sizes = [len(page_range(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        page_range(10, 2, 2)


def test_c_abi_page_range_is_the_same_split(prl):
    """prl_hip_page_range (what prl_hip_binarize_batch_host shards devices with) == prlib_amd.dist.page_range."""
    import ctypes as C

    from prlib_amd import _capi
    from prlib_amd.dist import page_range

    L = _capi.lib()
    for n in (0, 1, 7, 256, 1000, 1024):
        for parts in (1, 2, 3, 4, 8):
            for r in range(parts):
                first, count = C.c_int(-1), C.c_int(-1)
                assert L.prl_hip_page_range(n, parts, r, C.byref(first), C.byref(count)) == 0
                pr = page_range(n, parts, r)
                assert (first.value, count.value) == (pr.start, len(pr))
    first, count = C.c_int(0), C.c_int(0)
    assert L.prl_hip_page_range(10, 2, 2, C.byref(first), C.byref(count)) == _capi.PRL_ERR_BAD_ARG
    assert L.prl_hip_page_range(10, 0, 0, C.byref(first), C.byref(count)) == _capi.PRL_ERR_BAD_ARG


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pages, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from oracle import capi as oc
    from prlib_amd import dist as pd, synth

    w, r, _ = pd.init("gloo")
    assert (w, r) == (world, rank)
    p = oc.make_params(oc.SAUVOLA, 15, 0.34, 0)
    mine = pd.page_range(n_pages, world, rank)
    digest = hashlib.sha256()
    for i in mine:
        digest.update(oc.binarize(synth.page_numpy(40, 56, index=i), p).tobytes())
    pd.barrier()
    t = pd.max_over_ranks(float(rank + 1))
    total = pd.sum_over_ranks(float(len(mine)))
    assert t == float(world) and total == float(n_pages)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{mine.start} {mine.stop} {digest.hexdigest()}")
    pd.finish()
    assert not dist.is_initialized()


def test_two_gloo_ranks_equal_one_process(tmp_path):
    import torch.multiprocessing as mp

    from oracle import capi as oc
    from prlib_amd import synth
    from prlib_amd.dist import page_range

    n_pages, world = 7, 2
    mp.spawn(_worker, args=(world, _free_port(), n_pages, str(tmp_path)), nprocs=world, join=True)
    p = oc.make_params(oc.SAUVOLA, 15, 0.34, 0)
    for r in range(world):
        start, stop, got = open(tmp_path / f"rank{r}.txt").read().split()
        rng = page_range(n_pages, world, r)
        assert (int(start), int(stop)) == (rng.start, rng.stop)
        d = hashlib.sha256()
        for i in rng:
            d.update(oc.binarize(synth.page_numpy(40, 56, index=i), p).tobytes())
        assert d.hexdigest() == got


def test_bench_launcher_path_two_ranks_dryrun():
    """bench.py under `python -m torch.distributed.run --nproc-per-node 2` (gloo on CPU): rendezvous on 127.0.0.1,
    page sharding, barrier, max/sum reductions and the single JSON line from rank 0."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PRL_BENCH_DRYRUN="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--pages", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d == {"dryrun": True, "n_gpus": 2, "pages_total": 10, "max_rank_plus_1": 2.0, "first_block": [0, 5], "blocks": [[0, 5], [5, 10]],
                 "scaling": "weak", "strong_ride_along": True}   # N > 1: the weak line also carries the strong-scaling measurement


def _run_bench(argv, extra_env=None):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PRL_BENCH_DRYRUN="1", **(extra_env or {}))
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True,
                          timeout=300, env=env, cwd=root)


def test_bench_gpus_flag_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (child torch.distributed.run, before
    anything touches a GPU) - round 1's --gpus was parsed and ignored."""
    import json

    r = _run_bench(["--gpus", "2", "--pages", "7", "--scaling", "strong"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d == {"dryrun": True, "n_gpus": 2, "pages_total": 7, "max_rank_plus_1": 2.0, "first_block": [0, 4], "blocks": [[0, 4], [4, 7]],
                 "scaling": "strong", "strong_ride_along": False}


@pytest.mark.parametrize("scaling,pages,total", [("weak", 3, 24), ("strong", 61, 61)])
def test_bench_gpus_8_dryrun_partitions_the_page_list(scaling, pages, total):
    """The shape of the driver's first 8-GPU command, `python bench.py --gpus 8`, as eight gloo ranks on the CPU: rendezvous,
    barrier, max / sum over the ranks, one JSON line from rank 0 - and the eight contiguous page blocks partition the list
    (weak: 8 x pages-per-GPU; strong: a 61-page list that does not divide by 8, block sizes differ by at most one)."""
    import json

    r = _run_bench(["--gpus", "8", "--pages", str(pages), "--scaling", scaling, "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["pages_total"] == total and d["max_rank_plus_1"] == 8.0
    blocks = d["blocks"]
    assert len(blocks) == 8 and blocks[0][0] == 0 and blocks[-1][1] == total
    assert all(blocks[i][1] == blocks[i + 1][0] for i in range(7))             # contiguous, in rank order, no gap, no overlap
    sizes = [b - a for a, b in blocks]
    assert sum(sizes) == total and max(sizes) - min(sizes) <= 1


def test_bench_refuses_a_world_that_is_not_gpus():
    r = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2" in (r.stdout + r.stderr)
