"""Batched-file mode across the GPUs of one node (BASELINE config 5).

Streams are independent, so the codec itself needs no collective: every rank encodes (or
decodes) its own shard with its own engine.  RCCL (torch.distributed backend "nccl" on ROCm)
is used only around the data path, and only when a single rank owns the files:
  * the job table (file lengths) is broadcast from the root,
  * the encoded images are gathered to the root, which writes them out in job order.
Image sizes follow from the lengths (AADHip_CalculateEncodedSize), so every rank can compute all
offsets without an exchange.

Two shard encoders:
  encode_sharded_device  the product path: the shard's PCM is resident in HBM, the engine's plan
                         writes the images STRAIGHT INTO the rank's gather row (a device tensor),
                         dist.gather moves rows device-to-device (xGMI under RCCL) and the root
                         makes ONE device-to-host copy of what it received;
  encode_sharded         host-memory shards through a caller-supplied encode_fn
                         (Engine.encode_host bound to a parameter in production; the CPU tests
                         pass an oracle-based double - the product has no CPU codec).
With the gloo backend the same code runs on CPU tensors (rows are moved to the host first).
"""
import numpy as np


def partition_lpt(costs, world):
    """Longest-processing-time-first assignment of jobs to `world` bins.  Deterministic: ties
    break on the job index, so every rank derives the same shards from the same table.
    Returns a list (per rank) of job indices in ascending order.
    Every rank runs this on the whole table before it encodes anything: for 10 000 files it has to cost
    microseconds, not the 62 ms of a Python sort with a key function and a linear scan of the bins per job."""
    import heapq
    costs = np.asarray(costs, dtype=np.int64).reshape(-1)
    n = int(costs.size)
    if n == 0:
        return [[] for _ in range(world)]
    if int(costs.min()) == int(costs.max()):
        # equal costs: the greedy rule deals the jobs round-robin (job i to bin i mod world)
        return [list(range(r, n, world)) for r in range(world)]
    order = np.lexsort((np.arange(n), -costs))  # cost descending, index ascending
    heap = [(0, k) for k in range(world)]  # (load, bin): the least loaded bin, the lowest index among equals
    bins = [[] for _ in range(world)]
    cost = costs.tolist()
    for i in order.tolist():
        load, r = heap[0]
        bins[r].append(i)
        heapq.heapreplace(heap, (load + cost[i], r))
    return [sorted(b) for b in bins]


def _round_up(v, a):
    return (v + a - 1) // a * a


class BatchCodec:
    def __init__(self, encode_fn=None, rank=0, world=1, dist=None, device="cpu", engine=None, force_collectives=False):
        """encode_fn(list of int16 [samples, channels] arrays) -> list of bytes (host shards);
        engine: an aad_amd.Engine (device shards).  device: where collective tensors live -
        "cuda:N" with the nccl backend, "cpu" with gloo.
        force_collectives: a world of one normally skips the broadcast and the gather (nothing to
        exchange); with this flag it runs them anyway under its one-rank process group - the only way
        to execute the RCCL calls (device-tensor broadcast / gather) on a one-GPU box."""
        self.encode_fn, self.rank, self.world, self.dist, self.device, self.engine = encode_fn, rank, world, dist, device, engine
        self.force_collectives = bool(force_collectives)
        if self.force_collectives and dist is None:
            raise ValueError("force_collectives needs an initialised torch.distributed module")

    def _alone(self):
        """no peer to talk to, and nobody asked for the collectives to run regardless"""
        return self.world == 1 and not self.force_collectives

    # ---- collectives ----------------------------------------------------------------------
    def _collective_device(self):
        return self.device if self.dist is not None and self.dist.get_backend() == "nccl" else "cpu"

    def broadcast_table(self, lengths, root=0):
        """lengths: samples per file on the root (ignored elsewhere) -> int64 array on every rank"""
        import torch
        if self._alone():
            return np.asarray(lengths, dtype=np.int64)
        dev = self._collective_device()
        n = torch.tensor([len(lengths) if self.rank == root else 0], dtype=torch.int64, device=dev)
        self.dist.broadcast(n, src=root)
        t = torch.zeros(int(n.item()), dtype=torch.int64, device=dev)
        if self.rank == root:
            t.copy_(torch.as_tensor(np.asarray(lengths, dtype=np.int64)))
        self.dist.broadcast(t, src=root)
        return t.cpu().numpy()

    def _gather_rows(self, send, root):
        """send: uint8 tensor, the same length on every rank -> list of uint8 numpy rows on the root
        (None elsewhere).  The rows travel as they are (device tensors under nccl)."""
        import torch
        dev = self._collective_device()
        if str(send.device) != str(torch.device(dev)):
            send = send.to(dev)
        # the root receives into the rows of ONE block (no stacking copy afterwards) ...
        block = torch.empty((self.world, send.numel()), dtype=send.dtype, device=send.device) if self.rank == root else None
        self.dist.gather(send, list(block.unbind(0)) if block is not None else None, dst=root)
        if self.rank != root:
            return None
        return self._to_host(block)  # ... and brings it down with one device-to-host copy

    @staticmethod
    def _to_host(t):
        """device tensor -> numpy array through a pinned buffer (torch keeps pinned blocks cached): 12.8 MB come down in
        0.24 ms instead of 1.1 ms into pageable memory (tools/experiments/d2h_probe.py)"""
        import torch
        if t.device.type != "cuda":
            return t.numpy()
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        host.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return host.numpy()

    @staticmethod
    def _row_layout(shards, sizes):
        """16-byte aligned offsets of every image inside its rank's row (an int64 array indexed by job), and the common
        row length"""
        sizes16 = (np.asarray(sizes, dtype=np.int64) + 15) // 16 * 16
        offsets, row = np.zeros(len(sizes16), dtype=np.int64), 1
        for s in shards:
            if len(s) == 0:
                continue
            idx = np.asarray(s, dtype=np.int64)
            ends = np.cumsum(sizes16[idx])
            offsets[idx] = ends - sizes16[idx]
            row = max(row, int(ends[-1]))
        return offsets, row

    def _sizes(self, param, lengths):
        """encoded size of every job, one library call per DISTINCT length"""
        uniq, inverse = np.unique(lengths, return_inverse=True)
        per = np.asarray([self.engine.encoded_size(param, int(n)) for n in uniq], dtype=np.int64)
        return per[inverse]

    def _unpack(self, rows, shards, sizes, offsets):
        out = [None] * len(sizes)
        for r, s in enumerate(shards):
            row = rows[r]
            for i in s:
                o = int(offsets[i])
                out[i] = row[o:o + int(sizes[i])].tobytes()
        return out

    # ---- device-resident shards (the product path) -----------------------------------------------
    def encode_sharded_device(self, param, lengths, shard_pcm, root=0, return_rows=False):
        """Encode all files of the table.  shard_pcm(indices) returns this rank's PCM in HBM: an
        int16 cuda tensor [len(indices), samples, channels] (equal lengths) or a list of
        [samples_i, channels] tensors.  Returns the images in job order on the root (or, with
        return_rows, the raw gathered rows plus the layout - no per-file Python work), None elsewhere."""
        import torch
        from .capi import STREAM_DESC_DTYPE
        eng = self.engine
        lengths = np.asarray(lengths, dtype=np.int64)
        shards = partition_lpt(lengths, self.world)
        sizes = self._sizes(param, lengths)
        offsets, row = self._row_layout(shards, sizes)
        mine = shards[self.rank]
        send = torch.zeros(row, dtype=torch.uint8, device="cuda:%d" % eng.device)
        if mine:
            pcm = shard_pcm(mine)
            ch = param.num_channels
            d = np.zeros(len(mine), dtype=STREAM_DESC_DTYPE)
            if isinstance(pcm, (list, tuple)):
                starts = np.cumsum([0] + [_round_up(int(p.shape[0]) * ch, 8) for p in pcm])
                flat = torch.zeros(int(starts[-1]), dtype=torch.int16, device=send.device)
                for k, p in enumerate(pcm):
                    flat[int(starts[k]):int(starts[k]) + p.numel()] = p.reshape(-1)
                d["pcm_offset"] = starts[:-1]
            else:
                assert pcm.dtype == torch.int16 and pcm.is_contiguous() and pcm.shape[0] == len(mine)
                flat = pcm
                d["pcm_offset"] = np.arange(len(mine), dtype=np.uint64) * np.uint64(pcm.shape[1] * ch)
            mine_idx = np.asarray(mine, dtype=np.int64)
            d["data_offset"] = offsets[mine_idx]
            d["data_size"] = (sizes[mine_idx] + 15) // 16 * 16
            d["num_samples"] = lengths[mine_idx]
            plan = eng.encode_plan(param, d)
            try:
                plan.run(flat, send, None)
                torch.cuda.current_stream().synchronize()
            finally:
                plan.close()
        if self._alone():
            rows = self._to_host(send)[None, :]
        else:
            rows = self._gather_rows(send, root)
            if self.rank != root:
                return None
        if return_rows:
            return rows, shards, sizes, offsets
        return self._unpack(rows, shards, sizes, offsets)

    # ---- host-memory shards -------------------------------------------------------------------
    def encode_sharded(self, lengths, load_pcm, image_size, root=0):
        """Encode all files of the table.  load_pcm(i) returns file i's PCM on the rank that owns
        it; image_size(n) is the encoded size for n samples.  Returns the list of images in job
        order on the root, None elsewhere."""
        import torch
        lengths = np.asarray(lengths, dtype=np.int64)
        shards = partition_lpt(lengths, self.world)
        mine = shards[self.rank]
        images = self.encode_fn([load_pcm(i) for i in mine]) if mine else []
        uniq, inverse = np.unique(lengths, return_inverse=True)
        sizes = np.asarray([int(image_size(int(n))) for n in uniq], dtype=np.int64)[inverse]
        for i, img in zip(mine, images):
            assert len(img) == int(sizes[i]), "image size differs from the format arithmetic"
        if self._alone():
            return images
        offsets, row = self._row_layout(shards, sizes)  # sizes are static: no size exchange
        send = np.zeros(row, dtype=np.uint8)
        for i, img in zip(mine, images):
            send[int(offsets[i]):int(offsets[i]) + int(sizes[i])] = np.frombuffer(img, dtype=np.uint8)
        rows = self._gather_rows(torch.from_numpy(send), root)
        if self.rank != root:
            return None
        return self._unpack(rows, shards, sizes, offsets)
