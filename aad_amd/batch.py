"""Batched-file mode across the GPUs of one node (BASELINE config 5).

Streams are independent, so the codec itself needs no collective: every rank encodes (or
decodes) its own shard with its own engine.  RCCL (torch.distributed backend "nccl" on ROCm)
is used only around the data path, and only when a single rank owns the files:
  * the job table (file lengths + parameters) is broadcast from the root,
  * the encoded images are gathered to the root, which writes them out in job order.
Image sizes follow from the lengths (AADHip_CalculateEncodedSize), so every rank can compute all
offsets without an exchange.  With gloo the same code runs on CPU tensors (used by the tests,
with a test double in place of the engine - the product has no CPU codec).
"""
import numpy as np


def partition_lpt(costs, world):
    """Longest-processing-time-first assignment of jobs to `world` bins.  Deterministic: ties
    break on the job index, so every rank derives the same shards from the same table.
    Returns a list (per rank) of job indices in ascending order."""
    order = sorted(range(len(costs)), key=lambda i: (-int(costs[i]), i))
    load = [0] * world
    bins = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        bins[r].append(i)
        load[r] += int(costs[i])
    return [sorted(b) for b in bins]


class BatchCodec:
    """encode_fn(list of int16 [samples, channels] arrays) -> list of bytes, for one parameter set.
    In production encode_fn is Engine.encode_host bound to an AADEncodeParameter."""

    def __init__(self, encode_fn, rank=0, world=1, dist=None, device="cpu"):
        self.encode_fn, self.rank, self.world, self.dist, self.device = encode_fn, rank, world, dist, device

    def broadcast_table(self, lengths, root=0):
        """lengths: samples per file on the root (ignored elsewhere) -> int64 array on every rank"""
        import torch
        if self.world == 1:
            return np.asarray(lengths, dtype=np.int64)
        n = torch.tensor([len(lengths) if self.rank == root else 0], dtype=torch.int64, device=self.device)
        self.dist.broadcast(n, src=root)
        t = torch.zeros(int(n.item()), dtype=torch.int64, device=self.device)
        if self.rank == root:
            t.copy_(torch.as_tensor(np.asarray(lengths, dtype=np.int64)))
        self.dist.broadcast(t, src=root)
        return t.cpu().numpy()

    def encode_sharded(self, lengths, load_pcm, image_size, root=0):
        """Encode all files of the table.  load_pcm(i) returns file i's PCM on the rank that owns
        it; image_size(n) is the encoded size for n samples.  Returns the list of images in job
        order on the root, None elsewhere."""
        import torch
        lengths = np.asarray(lengths, dtype=np.int64)
        shards = partition_lpt(lengths, self.world)
        mine = shards[self.rank]
        images = self.encode_fn([load_pcm(i) for i in mine]) if mine else []
        sizes = [int(image_size(int(lengths[i]))) for i in range(len(lengths))]
        for i, img in zip(mine, images):
            assert len(img) == sizes[i], "image size differs from the format arithmetic"
        if self.world == 1:
            return images
        # one padded byte row per rank; sizes are static so no size exchange is needed
        shard_bytes = [sum(sizes[i] for i in s) for s in shards]
        row = max(shard_bytes + [1])
        send = torch.zeros(row, dtype=torch.uint8, device=self.device)
        if images:
            blob = np.frombuffer(b"".join(images), dtype=np.uint8)
            send[: len(blob)] = torch.as_tensor(blob.copy()).to(self.device)
        recv = [torch.zeros(row, dtype=torch.uint8, device=self.device) for _ in range(self.world)] \
            if self.rank == root else None
        self.dist.gather(send, recv, dst=root)
        if self.rank != root:
            return None
        out = [None] * len(lengths)
        for r, s in enumerate(shards):
            buf = recv[r].cpu().numpy()
            pos = 0
            for i in s:
                out[i] = bytes(buf[pos:pos + sizes[i]])
                pos += sizes[i]
        return out
