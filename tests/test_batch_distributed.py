"""N>1 path of the batched-file mode on CPU: world_size 2, gloo, 127.0.0.1 rendezvous.  The
engine is replaced by a test double built on the oracle (tests may use it; the product has no
CPU codec) - what is under test is the sharding, the static offsets and the gather."""
import os
import socket
import sys

import numpy as np
import pytest

from aad_amd.batch import BatchCodec, partition_lpt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_lpt_is_balanced_and_deterministic():
    costs = [9920] * 10 + [100, 50000, 3, 777, 31000]
    a = partition_lpt(costs, 4)
    assert a == partition_lpt(list(costs), 4)
    assert sorted(i for b in a for i in b) == list(range(len(costs)))
    loads = [sum(costs[i] for i in b) for b in a]
    assert max(loads) - min(loads) <= max(costs)
    assert partition_lpt([5, 5, 5, 5], 2) == [[0, 2], [1, 3]]
    assert partition_lpt([], 3) == [[], [], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lengths, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_binding as ob
    from aad_amd.synth import synth_pcm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        codec = BatchCodec(lambda pcms: [ob.encode(p, 4, 1024) for p in pcms], rank, world, dist)
        table = codec.broadcast_table(lengths if rank == 0 else [], root=0)
        assert list(table) == list(lengths)
        out = codec.encode_sharded(table, lambda i: synth_pcm(1, int(table[i]), 2, seed=1234, first_stream=i)[0],
                                   lambda n: ob.encoded_size(n, 2, 4, 1024), root=0)
        if rank == 0:
            q.put([__import__("hashlib").sha256(b).hexdigest() for b in out])
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    import torch.multiprocessing as mp
    import hashlib
    import oracle_binding as ob
    from aad_amd.synth import synth_pcm
    lengths = [992, 5000, 3, 9920, 1500, 992, 20000]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [hashlib.sha256(ob.encode(synth_pcm(1, n, 2, seed=1234, first_stream=i)[0], 4, 1024)).hexdigest()
            for i, n in enumerate(lengths)]
    assert got == want
