"""The N > 1 data-parallel path on CPU: two gloo ranks shard a batch, pad to the GLOBAL Tp and collate
their (stand-in) waveforms with the single all-gather of emojivoice_amd.dist."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from emojivoice_amd import dist as D

    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    B = 6
    lengths = torch.tensor([40, 13, 77, 5, 61, 130])           # global batch; rank 1 owns the longest utterance
    lo, hi = D.shard_bounds(B, r, w)
    tp = D.global_padded_length(int(lengths[lo:hi].max()))
    # stand-in for the vocoder output of this shard: value encodes (global row, sample index)
    wav = torch.stack([torch.arange(tp * 4, dtype=torch.float32) + 1000.0 * i for i in range(lo, hi)]).unsqueeze(1)
    full = D.all_gather_waveforms(wav)
    D.barrier()
    q.put((rank, tp, tuple(full.shape), float(full[:, 0, 0].sum()), float(full[-1, 0, -1])))
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tp, shape, s0, last in res:
        assert tp == 132                                  # max(130) rounded up to a multiple of 4, identical on both ranks
        assert shape == (6, 1, 132 * 4)                   # rows in global order
        assert s0 == sum(1000.0 * i for i in range(6))
        assert last == 5000.0 + 132 * 4 - 1
