#!/usr/bin/env python3
"""bench.py - Msamples/s of the AAD encode+decode hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches
one rank per GPU with torch.distributed.run.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]+[2], SURVEY.md section 8d "Config 2/3", primary form):
  1000 independent 48 kHz stereo 4-bit streams x 1 block (992 samples/channel) of the synthetic
  corpus (aad_amd/synth.py, seed 1234) -> encode to .aad images, then decode those images.
  A "step" = one encode pass + one decode pass over the batch, inputs resident in HBM.
  1 sample = 1 channel-sample; value = (samples encoded + samples decoded) / time, whole job.
Multi-GPU: streams are independent, so every rank runs its own batch (different seed offset),
no data-path collective: "scaling": "weak".

Extra objects on the JSON line:
  roofline      the dominant kernel (encode_streams_kernel) against the HBM roof, from HIP events
                recorded on the launch stream inside the timed region
  cpu_baseline  the compiled reference (oracle/_ref, kind "reference") or the oracle restatement
                (kind "port") timed single-threaded on this host, rank 0 / N=1 only
  saturated     the same kernels on a batch big enough to fill the chip (context for the
                roofline: every BASELINE config is lane-starved)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def measured_traffic(kernel_key, streams, samples):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/r01_hbm_traffic.json, made by tools/hbm_traffic.py from separate FETCH_SIZE /
    WRITE_SIZE runs of this same command line).  None when the file does not cover this workload."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    try:
        t = json.load(open(path))
    except Exception:
        return None
    if t.get("streams") != streams or t.get("samples_per_channel") != samples:
        return None
    k = t.get("kernels", {}).get(kernel_key)
    return k.get("hbm_bytes_per_launch") if k else None


def algorithmic_bytes_per_sample(channels, block_size, spb):
    """2 (int16 PCM) + block_size / (samples_per_block * channels)"""
    return 2.0 + block_size / float(spb * channels)


def measure(engine, torch, dist, pcm, param, steps, warmup, world, event_every=1, want_digests=False):
    """-> dict with wall ms/step (max over ranks) and mean kernel durations from HIP events."""
    streams, samples, ch = pcm.shape
    enc = engine.uniform_encode_plan(param, streams, samples)
    img = torch.zeros((streams, enc.stride), dtype=torch.uint8, device=pcm.device)
    enc.run(pcm, img, None)
    torch.cuda.synchronize()
    from aad_amd.engine import parse_header
    header = parse_header(bytes(img[0, :31].cpu().numpy()))
    dec = engine.uniform_decode_plan(header, streams, enc.stride, enc.image_size)
    out = torch.zeros((streams, samples, ch), dtype=torch.int16, device=pcm.device)

    def step(events=None):
        if events is not None:
            events[0].record()
        enc.run(pcm, img, None)
        if events is not None:
            events[1].record()
        dec.run(img, out)
        if events is not None:
            events[2].record()

    for _ in range(warmup):
        step()
    # HIP events bracket the two kernels on every `event_every`-th step of the timed region (each
    # record is a packet on the stream; bracketing every step would add ~2 % to a 140 us step)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] if k % event_every == 0 else None
           for k in range(steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        step(evs[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1: # MAX over ranks
        t = torch.tensor([dt], dtype=torch.float64, device=pcm.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timed = [e for e in evs if e is not None]
    enc_ms = sum(e[0].elapsed_time(e[1]) for e in timed) / len(timed)
    dec_ms = sum(e[1].elapsed_time(e[2]) for e in timed) / len(timed)
    ok = bool((out == pcm).float().mean() > 0.0)  # touch the result so nothing is elided
    digests = None
    if want_digests:  # what the timed steps left in HBM, for the bit-exact flag
        import hashlib
        digests = (hashlib.sha256(img[:, :enc.image_size].contiguous().cpu().numpy().tobytes()).hexdigest(),
                   hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest())
    enc.close()
    dec.close()
    return dict(wall_s=dt, enc_ms=enc_ms, dec_ms=dec_ms, header=header, touched=ok, digests=digests)


def golden_check(digests, streams, blocks, trials, rank):
    """Compare the bytes the timed steps produced with the hashes the compiled reference gave for
    the same corpus (tests/golden/manifest.json "corpora", made by tests/golden/make_golden.py).
    -> True / False, or None when the manifest does not hold this workload."""
    if digests is None or rank != 0 or blocks != 1:
        return None
    try:
        corpora = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["corpora"]
    except Exception:
        return None
    for c in corpora:
        if (c["streams"], c["samples"], c["channels"], c["bits"], c["trials"], c["seed"]) == (streams, 992, 2, 4, trials, 1234):
            return digests[0] == c["aad_concat_sha256"] and digests[1] == c["decoded_concat_sha256"]
    return None


def cpu_baseline(pcm_np, param_kw, budget_s=8.0):
    """Single-thread CPU reference on the same batch: encode all streams, decode all images."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libaad_oracle.so"], check=True)
    import oracle_binding as ob
    streams, samples, ch = pcm_np.shape
    bits, mbs, trials = param_kw["bits"], param_kw["max_block_size"], param_kw["trials"]
    if os.path.exists(ob.REF_SO):
        import aad_amd
        ref = aad_amd.LegacyCodec(aad_amd.load_library(ob.REF_SO, hip=False))
        kind = "reference"
        lib = ref.lib
        planar = [np.ascontiguousarray(pcm_np[s].T.astype(np.int32)) for s in range(streams)]
        cap = samples * ch * 2 + 4096
        outs = np.zeros((streams, cap), dtype=np.uint8)
        sizes = np.zeros(streams, dtype=np.uint32)
        param = aad_amd.make_parameter(ch, bits, mbs, 48000, False, trials)
        dec_buf = np.zeros((ch, samples), dtype=np.int32)
        from aad_amd.capi import _planar_pointers
        rows_in = [_planar_pointers(p) for p in planar]
        rows_out = _planar_pointers(dec_buf)
        u8p = C.POINTER(C.c_uint8)

        def enc_all():
            for s in range(streams):
                e = lib.AADEncoder_Create(mbs, None, 0)
                lib.AADEncoder_SetEncodeParameter(e, C.byref(param))
                sz = C.c_uint32()
                lib.AADEncoder_EncodeWhole(e, rows_in[s], samples, outs[s].ctypes.data_as(u8p), cap, C.byref(sz))
                sizes[s] = sz.value
                lib.AADEncoder_Destroy(e)

        def dec_all():
            for s in range(streams):
                d = lib.AADDecoder_Create(None, 0)
                lib.AADDecoder_DecodeWhole(d, outs[s].ctypes.data_as(u8p), int(sizes[s]), rows_out, ch, samples)
                lib.AADDecoder_Destroy(d)
    else:
        kind = "port"
        o = ob.lib()
        flat = np.ascontiguousarray(pcm_np)
        stride = ob.encoded_size(samples, ch, bits, mbs)
        outs = np.zeros((streams, stride), dtype=np.uint8)
        dec_buf = np.zeros((streams, samples, ch), dtype=np.int16)

        def enc_all():
            assert o.aado_encode_batch(flat.ctypes.data, streams, samples, ch, 48000, bits, mbs, 0, trials,
                                       outs.ctypes.data, stride) == 0

        def dec_all():
            assert o.aado_decode_batch(outs.ctypes.data, streams, stride, stride, dec_buf.ctypes.data, samples) == 0

    best_e = best_d = 1e9
    t_start, reps = time.perf_counter(), 0
    while reps < 3 or (time.perf_counter() - t_start < budget_s and reps < 200):
        t0 = time.perf_counter()
        enc_all()
        t1 = time.perf_counter()
        dec_all()
        t2 = time.perf_counter()
        best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
        reps += 1
    n = streams * samples * ch
    # all-cores figure (SURVEY.md section 8d "for honesty"): the oracle's batch entry points, one
    # C call per thread over a contiguous slice of the streams (ctypes drops the GIL for the whole
    # call; per-stream calls into the reference library would be Python-bound), kind "port"
    from concurrent.futures import ThreadPoolExecutor
    o = ob.lib()
    threads = max(1, min(os.cpu_count() or 1, 16))
    flat = np.ascontiguousarray(pcm_np)
    stride = ob.encoded_size(samples, ch, bits, mbs)
    p_out = np.zeros((streams, stride), dtype=np.uint8)
    p_dec = np.zeros((streams, samples, ch), dtype=np.int16)
    bounds = [(t * streams // threads, (t + 1) * streams // threads) for t in range(threads)]

    def port_slice(lohi):
        lo, hi = lohi
        if hi > lo:
            o.aado_encode_batch(flat[lo:hi].ctypes.data, hi - lo, samples, ch, 48000, bits, mbs, 0, trials,
                                p_out[lo:hi].ctypes.data, stride)
            o.aado_decode_batch(p_out[lo:hi].ctypes.data, hi - lo, stride, stride, p_dec[lo:hi].ctypes.data, samples)

    with ThreadPoolExecutor(threads) as pool:
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            list(pool.map(port_slice, bounds))
            best = min(best, time.perf_counter() - t0)
    all_cores = dict(value=round(2 * n / best / 1e6, 3), unit="Msamples/s", cores=threads, kind="port")
    return dict(value=round(2 * n / (best_e + best_d) / 1e6, 3), unit="Msamples/s", cores=1, kind=kind, all_cores=all_cores,
                sample="%d stereo streams x %d samples/ch (the full step batch), encode+decode, best of %d passes, 1 thread"
                       % (streams, samples, reps),
                encode_msps=round(n / best_e / 1e6, 3), decode_msps=round(n / best_d / 1e6, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=1000)
    ap.add_argument("--blocks", type=int, default=1, help="blocks per stream")
    ap.add_argument("--trials", type=int, default=0, help="num_encode_trials (reference CLI default is 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-saturated", action="store_true")
    ap.add_argument("--saturated-streams", type=int, default=262144)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--event-every", type=int, default=8, help="bracket the kernels with HIP events on every n-th timed step")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the AAD engine has no CPU path")
    local = local % torch.cuda.device_count()  # (a 1-GPU box can rehearse N>1 with --backend gloo)
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":  # RCCL on ROCm
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    from aad_amd.capi import make_parameter
    from aad_amd.engine import Engine
    from aad_amd.synth import synth_pcm

    ch, bits, mbs = 2, 4, 1024
    param = make_parameter(ch, bits, mbs, 48000, False, args.trials)
    engine = Engine(local)
    spb = 992
    samples = spb * args.blocks
    pcm_np = synth_pcm(args.streams, samples, ch, seed=1234, first_stream=rank * args.streams)
    pcm = torch.from_numpy(pcm_np).cuda()

    torch.cuda.synchronize()
    # everything timed below is launched on the engine's stream, and the HIP events are recorded on it
    torch.cuda.set_stream(engine.stream)
    m = measure(engine, torch, dist, pcm, param, args.steps, args.warmup, world, args.event_every, want_digests=True)
    hd = m["header"]
    n_step = args.streams * samples * ch  # channel-samples per direction per rank
    value = 2.0 * n_step * world * args.steps / m["wall_s"] / 1e6
    bps = algorithmic_bytes_per_sample(ch, hd.block_size, hd.num_samples_per_block)
    enc_gbs = n_step * bps / (m["enc_ms"] * 1e-3) / 1e9
    dec_gbs = n_step * bps / (m["dec_ms"] * 1e-3) / 1e9

    line = {
        "metric": "Msamples/s encode+decode, 48 kHz stereo 4-bit",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(m["wall_s"] / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]+[2]: %d independent 48 kHz stereo 4-bit streams x %d block(s) "
                        "(%d samples/ch) per GPU, encode then decode, device-resident" % (args.streams, args.blocks, samples),
            "streams_per_gpu": args.streams, "samples_per_channel": samples, "channels": ch, "bits_per_sample": bits,
            "max_block_size": mbs, "num_encode_trials": args.trials, "lane_mapping": os.environ.get("AAD_HIP_MAPPING", "auto (quad for this batch size: four lanes per recurrence; decode with the step-index walk as a parallel scan)"),
            "lanes_encode": args.streams * ch,
            "lanes_decode": args.streams * args.blocks * ch,
            "value_counts": "samples encoded + samples decoded",
        },
        "bit_exact_vs_reference_golden": golden_check(m["digests"], args.streams, args.blocks, args.trials, rank),
        "encode_msps": round(n_step * world / (m["enc_ms"] * 1e-3) / 1e6, 3),
        "decode_msps": round(n_step * world / (m["dec_ms"] * 1e-3) / 1e6, 3),
        "roofline": {
            "kernel": "aad::encode_streams_kernel<4>",
            "bound": "hbm", "achieved": round(enc_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(enc_gbs / HBM_PEAK_GBS, 6), "traffic": measured_traffic("encode", args.streams, samples),
            "algorithmic_bytes_per_launch": int(round(n_step * bps)),
            "bytes_per_sample": round(bps, 4), "samples_per_launch": n_step,
            "kernel_ms": round(m["enc_ms"], 5), "hip_events": "on every %d-th step of the timed region" % args.event_every,
            # what actually bounds a lane-starved launch (DESIGN.md "Kernels"): one wave per SIMD, one
            # instruction slot per ~4 cycles.  35.0 slots per sample is the 4-bit stereo quad encoder's
            # chunk loop counted in this build's ISA (560 per 16 samples, s_nop / s_waitcnt included).
            "issue_bound": {"slots_per_sample": 35.0, "samples_per_recurrence": samples - 4,
                            "achieved_Mslots_per_s_per_wave": round(35.0 * (samples - 4) / (m["enc_ms"] * 1e-3) / 1e6, 1),
                            "peak_Mslots_per_s_per_wave": 600.0,
                            "frac": round(35.0 * (samples - 4) / (m["enc_ms"] * 1e-3) / 600e6, 4),
                            "note": "peak = 2.4 GHz / 4 cycles per wave64 instruction on a 16-lane SIMD; kernel_ms includes launch and prologue"},
            "decode_kernel": {"kernel": "aad::decode_split_kernel<4> (quad batches) / aad::decode_blocks_kernel<4> (dense)", "achieved": round(dec_gbs, 3),
                              "frac": round(dec_gbs / HBM_PEAK_GBS, 6), "kernel_ms": round(m["dec_ms"], 5)},
        },
    }

    if not args.no_saturated and world == 1:
        big_streams = args.saturated_streams  # 262144 stereo streams = 8192 waves = 8 per SIMD (dense mapping)
        reps = -(-big_streams // args.streams)
        big = pcm.repeat((reps, 1, 1))[:big_streams].contiguous()
        ms = measure(engine, torch, dist, big, param, 5, 1, 1)
        del big
        nb = big_streams * samples * ch
        line["saturated"] = {
            "workload": "%d stereo streams x %d samples/ch per launch (the step batch tiled)" % (big_streams, samples),
            "encode_msps": round(nb / (ms["enc_ms"] * 1e-3) / 1e6, 1), "decode_msps": round(nb / (ms["dec_ms"] * 1e-3) / 1e6, 1),
            "encode_gbs": round(nb * bps / (ms["enc_ms"] * 1e-3) / 1e9, 2), "decode_gbs": round(nb * bps / (ms["dec_ms"] * 1e-3) / 1e9, 2),
            "encode_frac": round(nb * bps / (ms["enc_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "decode_frac": round(nb * bps / (ms["dec_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(pcm_np, dict(bits=bits, max_block_size=mbs, trials=args.trials))
    engine.close()
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
