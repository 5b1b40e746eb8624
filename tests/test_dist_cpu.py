"""CPU, world_size 2 over gloo: the clip-sharding + mask all-gather path used by bench.py at N > 1."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tce_rvos_amd.dist import run_sharded, shard_range
    clips = list(range(n_total))
    seen = []

    def fake_forward(c):  # stands for model(...)["pred_masks"][0]: a function of the clip only
        seen.append(c)
        return torch.full((2, 3, 4, 5), float(c)) + torch.arange(5, dtype=torch.float32)

    out = run_sharded(fake_forward, clips)
    lo, hi = shard_range(n_total, rank, world)
    ok = seen == list(range(lo, hi)) and tuple(out.shape) == (n_total, 2, 3, 4, 5)
    ok = ok and all(torch.equal(out[c], torch.full((2, 3, 4, 5), float(c)) + torch.arange(5, dtype=torch.float32))
                    for c in range(n_total))
    # the overlapped form bench.py uses: step i's gather is completed only after step i+1 has been launched
    from tce_rvos_amd.dist import gather_clip_masks_async
    pend, got = None, []
    for step in range(3):
        local = torch.stack([fake_forward(c) + 100.0 * step for c in range(lo, hi)], 0) if hi > lo else \
            torch.zeros((0, 2, 3, 4, 5))
        nxt = gather_clip_masks_async(local, n_total)
        if pend is not None:
            got.append(pend.wait())
        pend = nxt
    got.append(pend.wait())
    for step, g in enumerate(got):
        ok = ok and tuple(g.shape) == (n_total, 2, 3, 4, 5) and all(
            torch.equal(g[c], torch.full((2, 3, 4, 5), float(c) + 100.0 * step) + torch.arange(5, dtype=torch.float32))
            for c in range(n_total))
    q.put((rank, ok, hi - lo))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 5, 1])   # 1: rank 1's block is empty (n_total < world)
def test_clip_sharding_and_gather_world2(n_total):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sorted(n for _, _, n in res) == sorted([n_total // 2, n_total - n_total // 2])


def test_shard_range_partitions_exactly():
    from tce_rvos_amd.dist import shard_range
    for n in (0, 1, 7, 64):
        for w in (1, 2, 4, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
