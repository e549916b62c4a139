#!/usr/bin/env python3
"""bench.py - Msamples/s of the AAD encode+decode hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches
one rank per GPU with torch.distributed.run.  Rank 0 prints ONE JSON line.
`--gpus N` means N: under a launcher it must equal WORLD_SIZE (anything else exits 2, no line); run plainly
(`python bench.py --gpus 4`, no WORLD_SIZE) the script starts its own N ranks as a child process - before it has
touched the GPU itself - and leaves with the child's exit code; under RCCL it refuses more ranks than the node has GPUs
(`--backend gloo` rehearses N > 1 on one GPU).

Headline workload (BASELINE.json configs[1]+[2], SURVEY.md section 8d "Config 2/3", primary form):
  1000 independent 48 kHz stereo 4-bit streams x 1 block (992 samples/channel) of the synthetic
  corpus (aad_amd/synth.py, seed 1234) -> encode to .aad images, then decode those images.
  A "step" = one encode pass + one decode pass over the batch, inputs resident in HBM.
  1 sample = 1 channel-sample; value = (samples encoded + samples decoded) / time, whole job.
  Round 4: TWO steps in flight (`--in-flight 2`, the default): a 1000-stream step puts 125 waves per kernel on a chip of 1024 SIMDs,
  so step k runs on pipeline k mod 2 - two encode and two decode contexts on four streams, one hardware queue each; every step still
  encodes the whole batch and decodes what it encoded.  `one_pipeline` (= `--in-flight 1`, rounds 2-3's headline) and `serial` stay
  on the line.
Multi-GPU: streams are independent, so every rank runs its own batch (different corpus streams),
no data-path collective: "scaling": "weak".

Extra objects on the JSON line (N = 1 unless stated):
  roofline      the dominant kernel (encode_streams_kernel) against the HBM roof, from HIP events
                carried by the kernels' own dispatch packets (AADHip_ContextSignalNextRun) inside the timed region
  one_pipeline  rounds 2-3's headline: ONE step pipeline (the encode of step k+1 beside the decode of step k), same K
  serial        the step with nothing overlapped
  trials2       the same batch with num_encode_trials = 2, the reference CLI's default (src/main.c:45-47)
  end_to_end    PCIe-inclusive figures for the same batch: pinned buffers + device plans, and the
                host-memory C-ABI (AADHip_EncodeBatch / AADHip_DecodeBatch, pageable caller buffers)
  configs       the other BASELINE shapes: cfg2(ii) 1000 x 16 blocks, cfg2(iii) 1 x 1000 blocks,
                cfg4 8-channel 3-/2-bit x 10 000, cfg5's per-GPU shard 1250 files x 10 blocks - each
                with kernel ms, Msamples/s, HBM-roof fraction and a bit-exact flag against the hashes
                the compiled reference produced for the same corpus (tests/golden/manifest.json)
  saturated     the same kernels on a batch big enough to fill the chip (rank 0, any N)
  cpu_baseline  the compiled reference (oracle/_ref, kind "reference") or the oracle restatement
                (kind "port") on ONE pinned host core, same batch, through a C loop (rank 0, any N)
  config5       (N > 1 only) BASELINE config 5's batched-file mode: RCCL broadcast of the job table,
                1250 files per rank encoded device-resident, RCCL gather of the images to rank 0,
                which checks them against the reference's hashes (aad_amd/batch.py)
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
# VALU issue peak.  tools/microbench/ubench_rate.hip (profiles/r03_microbench_valu_rate.txt) measured what a SIMD retires per
# wave64 vector instruction with 1, 2, 4 and 8 waves on it: 2 cycles (the guide's figure) for v_add/sub_u32, v_and/or/xor_b32,
# v_lshrrev/ashrrev, v_mov_b32, v_add/mul_f32 once two waves share the SIMD - and 4 cycles, whatever the wave count, for
# everything these kernels' recurrences are made of: integer multiplies (24- and 32-bit), v_mad_i64_i32 / v_mad_u64_u32, every
# three-operand VOP3 form (v_and_or, v_lshl_or, v_add3, v_med3, v_perm, v_bfe, v_mad_i32_i24), v_min/max, v_lshlrev, conversions,
# DPP and SDWA forms, v_cmp, f64 and packed adds (v_fma_f32: 3.7).  The encoder's own mix ran at 3.80, the decoder's at 4.00 cycles
# per instruction at 8 waves per SIMD; ONE wave alone is offered a slot every 4.1-4.6 cycles.  256 CUs x 4 SIMDs, 2.4 GHz nominal
# (a chip-filling launch of these kernels holds ~1.85 GHz, GRBM_GUI_ACTIVE / duration: fractions against the nominal clock
# are the conservative ones).
SIMDS, CLOCK_GHZ, FULL_RATE_CYCLES, LONE_WAVE_CYCLES_PER_INST = 1024, 2.4, 2.0, 4.0
MIX_CYCLES = {"encode": 3.80, "decode": 4.00}  # cycles per instruction of the kernels' instruction mix (ubench_rate: encoder_mix, decoder_mix)
STAMP_FILE = os.path.join(ROOT, "profiles", "r04_pmc_stamp.json")


def kernel_source_digest():
    """SHA-256 over the device and host sources of libaad_hip.so (aad_amd/csrc), the stamp that ties
    a committed PMC measurement to the kernels it was taken from."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "aad_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".c")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def pmc_stamp(workload, role, streams, samples):
    """Counters per launch of the `role` ("encode" / "decode") kernel of `workload` ("headline" / "saturated") from
    the committed rocprofv3 --pmc passes (tools/collect_profiles.sh -> tools/stamp_pmc.py).  -> (dict or None, note).
    None when the file does not cover this workload OR was taken from different kernel sources than the ones this
    run was built from: a stale number is worse than none."""
    try:
        t = json.load(open(STAMP_FILE))
    except Exception:
        return None, "no committed PMC measurement"
    w = t.get("workloads", {}).get(workload)
    if not w or w.get("streams") != streams or w.get("samples_per_channel") != samples:
        return None, "the committed PMC measurement is for another workload"
    if t.get("kernel_source_sha256") != kernel_source_digest():
        return None, "kernel sources changed since the committed PMC passes (%s): re-run tools/collect_profiles.sh" % os.path.basename(STAMP_FILE)
    k = w.get("kernels", {}).get(role)
    return (k, "rocprofv3 --pmc, separate passes, %s" % os.path.basename(STAMP_FILE)) if k else (None, "kernel missing from the PMC file")


def traffic_fields(stamp, note, algorithmic):
    """roofline.traffic = FETCH_SIZE x 2 + WRITE_SIZE (the guide's gfx950 correction for 16-B-per-lane streams), the
    raw counter sum beside it.  A corrected figure below 0.95 x the algorithmic bytes cannot be right (every
    algorithmic byte has to cross the memory interface at least once): it is withheld, not printed."""
    if not stamp or "hbm_bytes_per_launch" not in stamp:
        return {"traffic": None, "traffic_raw": None, "traffic_source": note}
    t, raw = stamp["hbm_bytes_per_launch"], stamp["hbm_bytes_per_launch_raw"]
    if t < 0.95 * algorithmic:
        return {"traffic": None, "traffic_raw": raw, "traffic_withheld": t,
                "traffic_source": note + "; corrected traffic below 0.95 x algorithmic bytes: withheld"}
    return {"traffic": t, "traffic_raw": raw, "traffic_over_algorithmic": round(t / algorithmic, 3),
            "traffic_read_x2": stamp.get("read_bytes_x2"), "traffic_write": stamp.get("write_bytes"),
            "traffic_source": note + ": FETCH_SIZE x 2 + WRITE_SIZE (MI355X_MICROARCH.md, HBM: FETCH_SIZE tallies 64 B per "
                                     "128-B request); traffic_raw = FETCH_SIZE + WRITE_SIZE as counted"}


def valu_fields(stamp, note, kernel_ms, samples_per_recurrence, role):
    """The bound that actually applies (SURVEY.md section 8d "state both"): wave-instructions per second against what the
    chip's 1024 SIMDs can issue of THIS instruction mix (MIX_CYCLES), and against the full-rate figure.  Instruction counts
    from the stamped PMC pass (SQ_INSTS_VALU, SQ_WAVES), the duration is THIS run's (HIP events)."""
    if not stamp or "SQ_INSTS_VALU" not in stamp or not stamp.get("SQ_WAVES"):
        return {"frac": None, "source": note}
    insts, waves = stamp["SQ_INSTS_VALU"], stamp["SQ_WAVES"]
    rate = insts / (kernel_ms * 1e-3) / 1e9  # G wave-instructions / s
    mix = MIX_CYCLES[role]
    peak_mix, peak_full = SIMDS * CLOCK_GHZ / mix, SIMDS * CLOCK_GHZ / FULL_RATE_CYCLES
    occupied = min(waves, SIMDS)
    v = {"insts_per_launch": int(insts), "waves": int(waves),
         # wave-instructions a recurrence's wave issues per sample of the chain it walks (whole kernel: headers, tails,
         # loads and stores included)
         "insts_per_sample": round(insts / waves / samples_per_recurrence, 2),
         "insts_per_wave": round(insts / waves, 1),
         "wave_insts_per_s": round(rate * 1e9, 1), "peak_wave_insts_per_s": round(peak_mix * 1e9, 1),
         "cycles_per_inst_assumed": mix, "clock_ghz_assumed": CLOCK_GHZ,
         "frac": round(rate / peak_mix, 5),
         "peak_full_rate_wave_insts_per_s": peak_full * 1e9, "frac_of_full_rate": round(rate / peak_full, 5),
         "peak_note": "peak = 1024 SIMDs x 2.4 GHz / %.2f cycles: what a SIMD retires of this kernel's instruction mix (multiplies, "
                      "64-bit multiply-adds, VOP3, DPP, SDWA, conversions: 4 cycles each on gfx950; only add/sub/logic/right-shift/mov "
                      "go at the full rate of 2) - profiles/r03_microbench_valu_rate.txt" % mix,
         "source": note + " (SQ_INSTS_VALU, SQ_WAVES); duration from this run"}
    if "SQ_ACTIVE_INST_VALU" in stamp and stamp.get("GRBM_GUI_ACTIVE"):
        # clock-independent: quad-cycles in which a SIMD issued VALU work / SIMD-cycles of the profiled launch
        v["valu_active_frac_pmc"] = round(stamp["SQ_ACTIVE_INST_VALU"] * 4.0 / SIMDS / (stamp["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if stamp.get("GRBM_GUI_ACTIVE") and stamp.get("duration_ns") and waves > SIMDS:
        # the shader clock the chip HELD during the profiled launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs; only meaningful
        # when every XCD is busy for the whole launch): a chip-filling launch of these kernels runs at ~1.85 GHz, not at 2.4
        clk = stamp["GRBM_GUI_ACTIVE"] / 8.0 / stamp["duration_ns"]
        if 0.5 < clk <= CLOCK_GHZ * 1.02:
            v["clock_ghz_measured"] = round(clk, 3)
            v["frac_at_measured_clock"] = round(rate / (SIMDS * clk / mix), 5)
    if waves <= SIMDS:
        # fewer waves than SIMDs: each wave is alone on its SIMD and is offered an issue slot every ~4 cycles
        lone_peak = occupied * CLOCK_GHZ / LONE_WAVE_CYCLES_PER_INST
        v["occupied_simds"] = int(occupied)
        v["lone_wave_issue_frac"] = round(rate / lone_peak, 4)
        v["lone_wave_note"] = ("%d waves on %d SIMDs: a wave alone on its SIMD issues at most one instruction per ~%.0f cycles, so "
                               "this launch can use at most %.3f of the chip's VALU peak; lone_wave_issue_frac is the share of "
                               "THOSE slots it fills" % (waves, SIMDS, LONE_WAVE_CYCLES_PER_INST, lone_peak / peak_mix))
    return v


def algorithmic_bytes_per_sample(channels, block_size, spb):
    """2 (int16 PCM) + block_size / (samples_per_block * channels)"""
    return 2.0 + block_size / float(spb * channels)


def measure(engine, torch, dist, pcm, param, steps, warmup, world, event_every=1, keep=False, decode_engine=None,
            repeats=1, min_timed_s=0.0, max_repeats=1, collective=False, more_pipelines=()):
    """-> dict with wall ms/step (max over ranks) and mean kernel durations from HIP events.

    decode_engine: a second context (its own stream) -> the step is PIPELINED: the encode of step k+1
    runs while step k decodes - each of the two kernels fills half the chip's CUs on this batch - on two
    streams ordered by events, the images double-buffered so that an encode never overwrites what a
    decode still reads.  Every step still encodes the whole batch and decodes exactly what it encoded.
    more_pipelines: further (encode engine, decode engine) pairs -> SEVERAL STEPS IN FLIGHT: step k runs on pipeline
    k mod (1 + len(more_pipelines)), each pipeline with its own contexts, streams, image ring and output buffer, so that
    consecutive steps overlap in full (two encodes and two decodes at work at any time with one more pipeline)."""
    from aad_amd.engine import EncodeDecodePipeline, parse_header
    streams, samples, ch = pcm.shape
    out = torch.zeros((streams, samples, ch), dtype=torch.int16, device=pcm.device)
    pipe = enc = dec = None
    pipes, outs = [], [out]
    if decode_engine:
        pipe = EncodeDecodePipeline(engine, decode_engine, param, streams, samples, ring=32)  # one cross-stream wait per 16 encodes
        pipes = [pipe] + [EncodeDecodePipeline(e_, d_, param, streams, samples, ring=32) for e_, d_ in more_pipelines]
        outs = [out] + [torch.zeros_like(out) for _ in more_pipelines]
        header, image_size = pipe.header, pipe.enc.image_size
        last, turn = [None], [0]

        def step(events=None):
            p_ = turn[0] % len(pipes)
            turn[0] += 1
            last[0] = pipes[p_].step(pcm, outs[p_], events)
    else:
        enc = engine.uniform_encode_plan(param, streams, samples)
        img = torch.zeros((streams, enc.stride), dtype=torch.uint8, device=pcm.device)
        enc.run(pcm, img, None)
        torch.cuda.synchronize()
        header, image_size = parse_header(bytes(img[0, :31].cpu().numpy())), enc.image_size
        dec = engine.uniform_decode_plan(header, streams, enc.stride, enc.image_size)
        last = [img]

        def step(events=None):
            if events is not None:
                engine.signal_next(events[1], start=events[0])
            enc.run(pcm, img, None)
            if events is not None:
                engine.signal_next(events[3], start=events[2])
            dec.run(img, out)

    for _ in range(warmup):
        step()
    # HIP events time the two kernels on every `event_every`-th step of the timed region.  They ride on the kernels' own
    # dispatch packets (AADHip_ContextSignalNextRun: start / stop event of hipExtLaunchKernelGGL): the elapsed time of a pair
    # is the kernel's duration as rocprofv3 reports it, and the queue sees no packet more than on an untimed step
    from aad_amd.engine import HipEvent
    evs = [[HipEvent(timing=True) for _ in range(4)] if k % event_every == 0 else None for k in range(steps)]
    timed = [e for e in evs if e is not None]
    group = world > 1 or collective  # barriers and the MAX over ranks run whenever a process group is up
    regions, enc_sum, dec_sum, n_ev = [], 0.0, 0.0, 0
    # The K-step region is timed `repeats` times over (at least; more while less than `min_timed_s` of
    # timed work has accumulated, up to `max_repeats`): one region of the headline batch is ~1.5 ms at the
    # driver's K = 20, and a single such window moved by 6 % between two runs of the same build.
    # Every region is exactly K steps between barrier + synchronize on both sides; a rank's time runs from the common start
    # (behind the opening barrier) to the completion of its own last kernel (the closing synchronize), the region's time is
    # the MAX over ranks of those, and the closing barrier follows - its own latency (a collective launch and a device
    # synchronize, ~0.1 ms against a 1.4 ms region at K = 20) is not part of the K steps and would show up as a scaling
    # loss between N = 1, which runs no collective, and N > 1.  The line reports the median region.  The stop rule uses
    # the MAX-over-ranks times, so every rank runs the same count.
    while len(regions) < repeats or (sum(regions) < min_timed_s and len(regions) < max_repeats):
        torch.cuda.synchronize()
        if group:
            dist.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            step(evs[k])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0  # this rank's K steps, from the common start to its own last kernel
        if group:
            dist.barrier()  # the closing bracket; the region's time is the MAX over ranks below, not this collective's latency on top
        if group:  # MAX over ranks
            t = torch.tensor([dt], dtype=torch.float64, device=pcm.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        regions.append(dt)
        enc_sum += sum(e[0].elapsed_ms(e[1]) for e in timed)
        dec_sum += sum(e[2].elapsed_ms(e[3]) for e in timed)
        n_ev += len(timed)
    ordered = sorted(regions)
    dt = ordered[len(ordered) // 2] if len(ordered) % 2 else 0.5 * (ordered[len(ordered) // 2 - 1] + ordered[len(ordered) // 2])
    enc_ms, dec_ms = enc_sum / n_ev, dec_sum / n_ev
    ok = bool((out == pcm).float().mean() > 0.0)  # touch the result so nothing is elided
    res = dict(wall_s=dt, wall_min_s=ordered[0], wall_max_s=ordered[-1], regions=len(regions), timed_s=sum(regions),
               enc_ms=enc_ms, dec_ms=dec_ms, header=header, touched=ok, image_size=image_size)
    if pipes:
        out = outs[(turn[0] - 1) % len(pipes)]  # the LAST step's output
        res["outputs_identical"] = all(bool(torch.equal(o, outs[0])) for o in outs[1:])  # every pipeline decoded the same batch
        res["steps_in_flight"] = len(pipes)
    if keep:  # what the LAST timed step left in HBM, for the bit-exact flags
        res["img"] = last[0][:, :image_size].contiguous().cpu().numpy()
        res["out"] = out.cpu().numpy()
    if pipes:
        for p_ in pipes:
            p_.close()
    else:
        enc.close()
        dec.close()
    return res


# ---- golden checks against the hashes the compiled reference produced (tests/golden/manifest.json) ----

_MANIFEST = None


def manifest():
    global _MANIFEST
    if _MANIFEST is None:
        try:
            _MANIFEST = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
        except Exception:
            _MANIFEST = {}
    return _MANIFEST


def golden_check(m, streams, samples, ch, bits, trials, seed=1234):
    """-> True / False, or None when the manifest does not hold this corpus."""
    if "img" not in m:
        return None
    for c in manifest().get("corpora", []):
        if (c["streams"], c["samples"], c["channels"], c["bits"], c["trials"], c["seed"]) == (streams, samples, ch, bits, trials, seed):
            return (hashlib.sha256(m["img"].tobytes()).hexdigest() == c["aad_concat_sha256"] and
                    hashlib.sha256(m["out"].tobytes()).hexdigest() == c["decoded_concat_sha256"])
    return None


def golden_check_eight_channel(m, streams, samples, bits, pcm_np, seed=1234):
    """8-channel segments (the reference stops at 2 channels): every (segment, channel) re-framed as
    the mono image the reference produced for that channel; the decode must reproduce what the
    engine's own stereo/mono-pinned arithmetic implies, checked here as decode(encode) consistency
    with the per-channel hashes only on the encode side."""
    import numpy as np
    from aad_amd.reframe import channels_as_mono_images
    if "img" not in m:
        return None
    for c in manifest().get("eight_channel_corpora", []):
        if (c["streams"], c["samples"], c["bits"], c["seed"]) == (streams, samples, bits, seed):
            mono = channels_as_mono_images(m["img"], 8, bits, c["block_size"], c["mono_block_size"])
            return hashlib.sha256(np.ascontiguousarray(mono).tobytes()).hexdigest() == c["mono_images_concat_sha256"]
    return None


# ---- CPU baseline -------------------------------------------------------------------------------

def pin_to_one_core():
    """Pin this process to ONE of the cores it may run on ("single thread
    pinned to one core") -> (previous affinity set, chosen core) or (None, None) where unsupported."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        core = allowed[len(allowed) // 2]
        os.sched_setaffinity(0, {core})
        return set(allowed), core
    except (AttributeError, OSError):
        return None, None


def cpu_baseline(pcm_np, bits, mbs, trials_list, budget_s=6.0):
    """Single-thread CPU path on the same batch, one pinned core: encode all streams, decode all
    images, through ONE C call per direction (oracle/ref_batch.c around the compiled reference's
    API, or the oracle's batch entry points).  -> dict keyed by trials."""
    import numpy as np
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libaad_oracle.so"], check=True)
    import oracle_binding as ob
    streams, samples, ch = pcm_np.shape
    n = streams * samples * ch
    o = ob.lib()
    flat = np.ascontiguousarray(pcm_np)
    stride = ob.encoded_size(samples, ch, bits, mbs)
    ref = None
    if os.path.exists(ob.REF_SO):
        try:
            ref = C.CDLL(ob.REF_SO)
            ref.refbatch_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_void_p, C.c_size_t, C.c_void_p]
            ref.refbatch_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
            ref.refbatch_planar_from_pcm.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        except (OSError, AttributeError):
            ref = None
    outs = np.zeros((streams, stride), dtype=np.uint8)
    if ref is not None:
        kind = "reference"
        planar = np.zeros((streams, ch, samples), dtype=np.int32)
        ref.refbatch_planar_from_pcm(flat.ctypes.data, streams, samples, ch, planar.ctypes.data)
        sizes = np.zeros(streams, dtype=np.uint32)
        dec_planar = np.zeros_like(planar)

        def enc_all(trials):
            assert ref.refbatch_encode(planar.ctypes.data, streams, samples, ch, bits, mbs, trials, outs.ctypes.data, stride,
                                       sizes.ctypes.data) == 0

        def dec_all():
            assert ref.refbatch_decode(outs.ctypes.data, streams, stride, sizes.ctypes.data, samples, ch, dec_planar.ctypes.data) == 0
    else:
        kind = "port"
        dec_buf = np.zeros((streams, samples, ch), dtype=np.int16)

        def enc_all(trials):
            assert o.aado_encode_batch(flat.ctypes.data, streams, samples, ch, 48000, bits, mbs, 0, trials, outs.ctypes.data, stride) == 0

        def dec_all():
            assert o.aado_decode_batch(outs.ctypes.data, streams, stride, stride, dec_buf.ctypes.data, samples) == 0

    previous, core = pin_to_one_core()
    result = {}
    try:
        for trials in trials_list:
            best_e = best_d = 1e9
            t_start, reps = time.perf_counter(), 0
            while reps < 3 or (time.perf_counter() - t_start < budget_s / len(trials_list) and reps < 200):
                t0 = time.perf_counter()
                enc_all(trials)
                t1 = time.perf_counter()
                dec_all()
                t2 = time.perf_counter()
                best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
                reps += 1
            result[trials] = dict(value=round(2 * n / (best_e + best_d) / 1e6, 3), unit="Msamples/s", cores=1, kind=kind,
                                  pinned_core=core,
                                  sample="%d stereo streams x %d samples/ch (the full step batch), trials %d, encode+decode, "
                                         "one C call per direction, best of %d passes, 1 thread pinned with sched_setaffinity"
                                         % (streams, samples, trials, reps),
                                  encode_msps=round(n / best_e / 1e6, 3), decode_msps=round(n / best_d / 1e6, 3))
    finally:
        if previous:
            os.sched_setaffinity(0, previous)
    # all-cores figure (SURVEY.md section 8d "for honesty"): the oracle's batch entry points, one C call
    # per thread over a contiguous slice of the streams (ctypes drops the GIL for the call), kind "port"
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))
    p_out = np.zeros((streams, stride), dtype=np.uint8)
    p_dec = np.zeros((streams, samples, ch), dtype=np.int16)
    bounds = [(t * streams // threads, (t + 1) * streams // threads) for t in range(threads)]

    def port_slice(lohi):
        lo, hi = lohi
        if hi > lo:
            o.aado_encode_batch(flat[lo:hi].ctypes.data, hi - lo, samples, ch, 48000, bits, mbs, 0, trials_list[0],
                                p_out[lo:hi].ctypes.data, stride)
            o.aado_decode_batch(p_out[lo:hi].ctypes.data, hi - lo, stride, stride, p_dec[lo:hi].ctypes.data, samples)

    with ThreadPoolExecutor(threads) as pool:
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            list(pool.map(port_slice, bounds))
            best = min(best, time.perf_counter() - t0)
    result[trials_list[0]]["all_cores"] = dict(value=round(2 * n / best / 1e6, 3), unit="Msamples/s", cores=threads, kind="port")
    return result


# ---- PCIe-inclusive figures -------------------------------------------------------------------------

def end_to_end(engine, torch, pcm_np, param, reps=20):
    """(a) pinned host buffers -> H2D -> kernel -> D2H, asynchronous on the engine's stream with device
    plans (what a caller that owns pinned memory gets); (b) the host-memory C-ABI with pageable
    caller buffers, the call pattern of the reference CLI (src/main.c:182-198, :91-106) for a batch:
    AADHip_EncodeBatch / AADHip_DecodeBatch, timed around the C call only (pointer tables and
    output buffers are built beforehand, as a C caller would hold them)."""
    import numpy as np
    from aad_amd.engine import parse_header
    streams, samples, ch = pcm_np.shape
    n = streams * samples * ch
    h_pcm = torch.from_numpy(pcm_np).pin_memory()
    enc = engine.uniform_encode_plan(param, streams, samples)
    d_pcm = torch.empty_like(h_pcm, device="cuda")
    d_img = torch.zeros((streams, enc.stride), dtype=torch.uint8, device="cuda")
    h_img = torch.empty((streams, enc.stride), dtype=torch.uint8).pin_memory()
    h_out = torch.empty_like(h_pcm).pin_memory()
    d_out = torch.empty_like(d_pcm)
    d_pcm.copy_(h_pcm, non_blocking=True)
    enc.run(d_pcm, d_img, None)
    torch.cuda.synchronize()
    hd = parse_header(bytes(d_img[0, :31].cpu().numpy()))
    dec = engine.uniform_decode_plan(hd, streams, enc.stride, enc.image_size)
    dec.run(d_img, d_out)
    torch.cuda.synchronize()
    # every repetition has events of its own and the host waits once, at the end: a wait per repetition
    # lets the copy engines go idle, and the first copy after that pays ~0.4 ms of wake-up
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
    for ev in evs:
        ev[0].record()
        d_pcm.copy_(h_pcm, non_blocking=True)
        enc.run(d_pcm, d_img, None)
        h_img.copy_(d_img, non_blocking=True)
        ev[1].record()
        d_img.copy_(h_img, non_blocking=True)
        dec.run(d_img, d_out)
        h_out.copy_(d_out, non_blocking=True)
        ev[2].record()
    torch.cuda.synchronize()
    te = sum(ev[0].elapsed_time(ev[1]) for ev in evs[1:])
    td = sum(ev[1].elapsed_time(ev[2]) for ev in evs[1:])
    reps -= 1  # the first repetition starts from an idle device
    enc.close()
    dec.close()
    pinned = dict(encode_ms=round(te / reps, 4), decode_ms=round(td / reps, 4),
                  encode_msps=round(n / (te / reps) / 1e3, 1), decode_msps=round(n / (td / reps) / 1e3, 1),
                  path="pinned host -> H2D -> kernel -> D2H -> pinned host, HIP events per repetition, "
                       "repetitions queued back to back (one host wait at the end)")

    # (b) host-memory C-ABI, pageable buffers
    lib, ctx = engine.lib, engine._ctx
    size = engine.encoded_size(param, samples)
    rows = [np.ascontiguousarray(pcm_np[i]) for i in range(streams)]
    imgs = [np.zeros(size, dtype=np.uint8) for _ in range(streams)]
    decs = [np.zeros((samples, ch), dtype=np.int16) for _ in range(streams)]
    nsamp = np.full(streams, samples, dtype=np.uint32)
    caps = np.full(streams, size, dtype=np.uint64)
    sizes = np.zeros(streams, dtype=np.uint64)
    got = np.zeros(streams, dtype=np.uint32)
    pp = (C.c_void_p * streams)(*[r.ctypes.data for r in rows])
    ip = (C.c_void_p * streams)(*[r.ctypes.data for r in imgs])
    dp = (C.c_void_p * streams)(*[r.ctypes.data for r in decs])
    t_enc, t_dec = [], []
    for k in range(reps + 2):
        t0 = time.perf_counter()
        rc1 = lib.AADHip_EncodeBatch(ctx, C.byref(param), streams, pp, nsamp.ctypes.data, ip, caps.ctypes.data, sizes.ctypes.data, None)
        t1 = time.perf_counter()
        rc2 = lib.AADHip_DecodeBatch(ctx, streams, ip, sizes.ctypes.data, dp, nsamp.ctypes.data, got.ctypes.data)
        t2 = time.perf_counter()
        assert rc1 == 0 and rc2 == 0, (rc1, rc2, engine.last_error())
        if k >= 2:  # the first calls size the staging blocks
            t_enc.append(t1 - t0)
            t_dec.append(t2 - t1)
    me, md = sorted(t_enc)[len(t_enc) // 2], sorted(t_dec)[len(t_dec) // 2]
    same = all(np.array_equal(decs[i], h_out[i].numpy()) for i in range(0, streams, 97))
    host = dict(encode_ms=round(me * 1e3, 4), decode_ms=round(md * 1e3, 4), encode_msps=round(n / me / 1e6, 1),
                decode_msps=round(n / md / 1e6, 1), same_output_as_device_plans=bool(same),
                path="AADHip_EncodeBatch / AADHip_DecodeBatch: pageable caller buffers -> one pinned block -> H2D -> kernel "
                     "-> D2H -> caller buffers, median of %d synchronous calls" % reps)
    return dict(workload="%d stereo 4-bit streams x %d samples/ch, trials %d" % (streams, samples, param.num_encode_trials),
                pinned_device_plans=pinned, host_memory_api=host,
                note="PCIe-inclusive: never the headline `value` (inputs are resident in HBM there)")


def config_counters(stamp_key, streams, samples, algorithmic, enc_ms, dec_ms, spb):
    """Counter evidence of a `configs[]` row from the committed rocprofv3 passes (tools/collect_config_profiles.sh): per kernel the
    HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE), its ratio to the algorithmic bytes, the VMEM instruction counts and the
    wave-instructions per sample.  Same stamp rule as the headline: withheld when the kernel sources have changed since."""
    out = {}
    for role, kms, per_rec in (("encode", enc_ms, samples), ("decode", dec_ms, min(spb, samples))):
        st, note = pmc_stamp(stamp_key, role, streams, samples) if stamp_key else (None, "no PMC passes for this row")
        row = dict({"kernel": (st or {}).get("kernel")}, **traffic_fields(st, note, algorithmic))
        if st and st.get("SQ_WAVES"):
            v = valu_fields(st, note, kms, per_rec, role)
            row["valu"] = {k: v.get(k) for k in ("insts_per_sample", "insts_per_wave", "waves", "frac", "frac_of_full_rate", "lone_wave_issue_frac") if k in v}
            row["vmem_rd_insts"], row["vmem_wr_insts"] = st.get("SQ_INSTS_VMEM_RD"), st.get("SQ_INSTS_VMEM_WR")
        out[role] = row
    return out


def config_entry(engine, torch, dist, name, streams, samples, ch, bits, trials, steps, golden, stamp_key=None):
    """One of the other BASELINE shapes: kernel time (HIP events), Msamples/s, HBM-roof fraction, the counter traffic of the
    committed PMC passes and the bit-exact flag against the compiled reference's hashes for the same corpus."""
    from aad_amd.capi import make_parameter
    from aad_amd.synth import synth_pcm
    pcm_np = synth_pcm(streams, samples, ch, seed=1234)
    pcm = torch.from_numpy(pcm_np).cuda()
    param = make_parameter(ch, bits, 1024, 48000, False, trials)
    m = measure(engine, torch, dist, pcm, param, steps, 1, 1, 1, keep=True)
    hd = m["header"]
    n = streams * samples * ch
    bps = algorithmic_bytes_per_sample(ch, hd.block_size, hd.num_samples_per_block)
    if golden == "eight":
        flag = golden_check_eight_channel(m, streams, samples, bits, pcm_np)
    else:
        flag = golden_check(m, streams, samples, ch, bits, trials)
    del pcm
    return dict(config=name, streams=streams, samples_per_channel=samples, channels=ch, bits=bits, trials=trials,
                encode_ms=round(m["enc_ms"], 4), decode_ms=round(m["dec_ms"], 4),
                encode_msps=round(n / m["enc_ms"] / 1e3, 1), decode_msps=round(n / m["dec_ms"] / 1e3, 1),
                bytes_per_sample=round(bps, 4), algorithmic_bytes_per_launch=int(round(n * bps)),
                encode_frac=round(n * bps / (m["enc_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                decode_frac=round(n * bps / (m["dec_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                counters=config_counters(stamp_key, streams, samples, int(round(n * bps)), m["enc_ms"], m["dec_ms"], hd.num_samples_per_block),
                bit_exact_vs_reference_golden=flag)


def config5_batched_files(engine, torch, dist, rank, world, files_per_rank=1250, blocks=10, force_collectives=False):
    """BASELINE config 5: one rank owns the job table; RCCL broadcast of the
    table, every rank encodes its files device-resident (its own reader: the corpus streams it was
    dealt), RCCL gather of the image rows to rank 0, which hashes them in job order against the
    compiled reference's hashes.  Timed between barriers, max over ranks."""
    from aad_amd.batch import BatchCodec
    from aad_amd.capi import make_parameter
    from aad_amd.synth import synth_pcm
    param = make_parameter(2, 4, 1024, 48000, False, 0)
    samples = 992 * blocks
    total = files_per_rank * world
    codec = BatchCodec(rank=rank, world=world, dist=dist, device="cuda:%d" % engine.device, engine=engine,
                       force_collectives=force_collectives)

    def shard_pcm(indices):  # this rank's "files": corpus streams `indices`
        parts = [synth_pcm(1, samples, 2, seed=1234, first_stream=i)[0] for i in indices]
        import numpy as np
        return torch.from_numpy(np.stack(parts)).cuda()

    table = codec.broadcast_table([samples] * total if rank == 0 else [], root=0)
    pcm_cache = {}

    def cached(indices):
        key = tuple(indices)
        if key not in pcm_cache:
            pcm_cache[key] = shard_pcm(indices)
        return pcm_cache[key]

    codec.encode_sharded_device(param, table, cached, root=0, return_rows=True)  # warm-up: plans, RCCL channels
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    res = codec.encode_sharded_device(param, table, cached, root=0, return_rows=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0  # this rank's share (the root's includes the gather and the copy to the host); MAX over ranks below
    dist.barrier()
    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank != 0:
        return None
    rows, shards, sizes, offsets = res
    h = hashlib.sha256()
    for i in range(total):  # job order
        r = i % world  # equal lengths: partition_lpt deals round-robin
        assert shards[r][i // world] == i
        h.update(rows[r][offsets[i]:offsets[i] + sizes[i]].tobytes())
    flag = None
    for c in manifest().get("file_corpora", []):
        if c["name"] == "cfg5_stereo4_10000x10blk_t0":
            flag = c.get("aad_prefix_sha256", {}).get(str(total))
            flag = (h.hexdigest() == flag) if flag else None
    n = total * samples * 2
    return dict(workload="%d stereo 4-bit files x %d blocks (%d per rank): RCCL broadcast of the job table, device-resident "
                         "encode per rank, RCCL gather of the images to rank 0" % (total, blocks, files_per_rank),
                backend=dist.get_backend(), files=total, seconds=round(dt, 6), encode_msps=round(n / dt / 1e6, 1),
                gathered_bytes=int(sum(sizes)), bit_exact_vs_reference_golden=flag,
                includes="table broadcast excluded; shard encode + gather + device-to-host copy on rank 0 included")


def launch_command(argv, gpus, port):
    """the command `python bench.py --gpus N ...` turns into when it is run plainly (no WORLD_SIZE in the environment):
    the driver's own launch line - one rank per GPU of ONE node under torch.distributed.run, rendezvous on 127.0.0.1"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(argv, gpus):
    """Start the N ranks as a CHILD process and leave with its exit code.  Runs before anything in this process has
    touched the GPU (torch is not even imported yet): a process that has initialised HIP must not exec or fork GPU
    work.  Rank 0 of the child writes the one JSON line to the stdout it inherits from here."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    return subprocess.run(launch_command(argv, gpus, port), env=env).returncode


def resolve_world(gpus, environ):
    """--gpus N against the launcher's environment -> ("run", rank, local_rank, world) or ("launch", N) or ("error", text).
    `--gpus N` MEANS N: a line with n_gpus != N is never printed."""
    if gpus < 1:
        return ("error", "--gpus must be at least 1")
    if "WORLD_SIZE" not in environ:
        return ("run", 0, 0, 1) if gpus == 1 else ("launch", gpus)
    try:
        world, rank, local = int(environ["WORLD_SIZE"]), int(environ.get("RANK", "0")), int(environ.get("LOCAL_RANK", "0"))
    except ValueError:
        return ("error", "WORLD_SIZE / RANK / LOCAL_RANK are not integers")
    if world != gpus:
        return ("error", "--gpus %d but the launcher started WORLD_SIZE=%d ranks: pass --gpus %d (or run `python bench.py --gpus %d` "
                         "plainly and let it start its own ranks)" % (gpus, world, world, gpus))
    if not 0 <= rank < world:
        return ("error", "RANK=%d outside WORLD_SIZE=%d" % (rank, world))
    return ("run", rank, local, world)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node; run plainly with N > 1 the script starts its own N ranks "
                                                         "(torch.distributed.run), under a launcher it must equal WORLD_SIZE")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=1000)
    ap.add_argument("--blocks", type=int, default=1, help="blocks per stream")
    ap.add_argument("--trials", type=int, default=0, help="num_encode_trials of the headline (reference CLI default is 2: see `trials2`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-saturated", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip trials2 / end_to_end / configs (profiling runs)")
    ap.add_argument("--saturated-streams", type=int, default=262144)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--serial", action="store_true", help="do not overlap the encode of step k+1 with the decode of step k")
    ap.add_argument("--in-flight", type=int, default=2, choices=(1, 2),
                    help="steps in flight: 2 (default) = two encode + decode pipelines on four streams, stepped in turn; 1 = one pipeline (rounds 2-3)")
    ap.add_argument("--event-every", type=int, default=8, help="bracket the kernels with HIP events on every n-th timed step")
    ap.add_argument("--repeats", type=int, default=9, help="time the K-step region at least this many times; the line reports the median region")
    ap.add_argument("--min-timed-ms", type=float, default=50.0, help="...and until this much timed work has accumulated (at most 101 regions)")
    ap.add_argument("--no-config5", action="store_true",
                    help="N = 1: skip the batched-file leg under a one-rank RCCL group (BASELINE config 5's code path with the collectives forced)")
    args = ap.parse_args()

    plan = resolve_world(args.gpus, os.environ)
    if plan[0] == "error":
        sys.stderr.write("bench.py: %s\n" % plan[1])
        raise SystemExit(2)
    if plan[0] == "launch":
        raise SystemExit(launch_ranks(sys.argv[1:], plan[1]))
    _, rank, local, world = plan

    # Rank 0 owes the driver ONE line on stdout.  RCCL prints a version banner on stdout when its first communicator
    # comes up (seen on the GPU box: five lines in front of the JSON), so file descriptor 1 is pointed at stderr for
    # the whole run - native libraries included - and the line goes out through a saved duplicate at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    if world > 1 and args.backend == "nccl" and torch.cuda.device_count() < world:  # counting devices does not initialise HIP
        raise SystemExit("bench.py: --gpus %d under RCCL needs %d GPUs, this node shows %d (RCCL refuses two ranks on one device; "
                         "--backend gloo rehearses N > 1 on fewer)" % (world, world, torch.cuda.device_count()))
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
    # ALL the streams this run steps its pipelines on are made here, before the first host-to-device copy: the HIP runtime has four
    # hardware queues and binds a stream to one when the stream is made (a later stream gets whichever queue is least used), and a
    # pageable copy in between takes a queue for the runtime's own use - with the order engine, copy, other engines two of the four
    # pipeline streams ended up on ONE queue and two steps in flight measured 60 000 instead of 87 000-101 000 Msamples/s, every time
    # (tools/experiments/inflight_fresh.py ORDER=bench).  Each of the two kernels fills about half of the chip's CUs on this batch
    # (125 workgroups), so the step is pipelined over two contexts: encode of step k+1 while step k decodes (see measure()).
    # --serial runs the two launches of a step strictly one after the other, as rounds 1 and 2a did; the line carries that figure
    # too (`serial`).
    engine = Engine(local)
    decode_engine = None if args.serial else Engine(local, stream=torch.cuda.Stream(local))  # a stream of its own, not torch's current one
    # the second pipeline (--in-flight 2): two more contexts on two more streams - four streams = the runtime's four hardware queues
    second_pair = None
    if not args.serial and args.in_flight == 2:
        second_pair = (Engine(local, stream=torch.cuda.Stream(local)), Engine(local, stream=torch.cuda.Stream(local)))
    more = (second_pair,) if second_pair else ()
    spb = 992
    samples = spb * args.blocks
    pcm_np = synth_pcm(args.streams, samples, ch, seed=1234, first_stream=rank * args.streams)
    pcm = torch.from_numpy(pcm_np).cuda()

    torch.cuda.synchronize()
    # everything timed below is launched on the engine's stream, and the HIP events are recorded on it
    torch.cuda.set_stream(engine.stream)
    m = measure(engine, torch, dist, pcm, param, args.steps, args.warmup, world, args.event_every, keep=(rank == 0),
                decode_engine=decode_engine, repeats=args.repeats, min_timed_s=args.min_timed_ms * 1e-3, max_repeats=max(args.repeats, 101),
                more_pipelines=more)
    hd = m["header"]
    n_step = args.streams * samples * ch  # channel-samples per direction per rank
    value = 2.0 * n_step * world * args.steps / m["wall_s"] / 1e6  # the MEDIAN K-step region
    bps = algorithmic_bytes_per_sample(ch, hd.block_size, hd.num_samples_per_block)
    enc_gbs = n_step * bps / (m["enc_ms"] * 1e-3) / 1e9
    dec_gbs = n_step * bps / (m["dec_ms"] * 1e-3) / 1e9
    stamp_e, note_e = pmc_stamp("headline", "encode", args.streams, samples)
    stamp_d, note_d = pmc_stamp("headline", "decode", args.streams, samples)
    algorithmic = int(round(n_step * bps))
    quad_note = "auto (quad for this batch size: four lanes per recurrence; decode with the step-index walk as a parallel scan)"

    line = {
        "metric": "Msamples/s encode+decode, 48 kHz stereo 4-bit",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(m["wall_s"] / args.steps * 1e3, 5),
        "value_min": round(2.0 * n_step * world * args.steps / m["wall_max_s"] / 1e6, 3),
        "value_max": round(2.0 * n_step * world * args.steps / m["wall_min_s"] / 1e6, 3),
        "timed_regions": {"count": m["regions"], "steps_each": args.steps, "timed_s": round(m["timed_s"], 6),
                          "rule": "the K-step region (barrier + synchronize on both sides) is timed at least %d times and until "
                                  "%.0f ms of timed work; value / ms_per_step are the MEDIAN region, value_min / value_max the "
                                  "slowest / fastest" % (args.repeats, args.min_timed_ms)},
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "io_dtype": "int16 PCM / packed 4-bit codes (the recurrence itself is int32 with wraparound)",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]+[2]: %d independent 48 kHz stereo 4-bit streams x %d block(s) "
                        "(%d samples/ch) per GPU, encode then decode, device-resident" % (args.streams, args.blocks, samples),
            "streams_per_gpu": args.streams, "samples_per_channel": samples, "channels": ch, "bits_per_sample": bits,
            "max_block_size": mbs, "num_encode_trials": args.trials,
            "lane_mapping": os.environ.get("AAD_HIP_MAPPING", quad_note),
            "lanes_encode": args.streams * ch,
            "lanes_decode": args.streams * args.blocks * ch,
            "value_counts": "samples encoded + samples decoded",
            "steps_in_flight": m.get("steps_in_flight", 1),
            "pipeline": ("serial: the decode of step k ends before the encode of step k+1 starts" if args.serial else
                         ("one pipeline - two contexts on two streams: the encode of step k+1 overlaps the decode of step k (events order "
                          "them, images double-buffered); every step encodes the whole batch and decodes what it encoded") if not more else
                         ("two steps in flight - two such pipelines (two encode and two decode contexts on four streams, one hardware queue "
                          "each), step k on pipeline k mod 2: a 1000-stream step puts 125 waves per kernel on a chip of 1024 SIMDs, so two "
                          "encodes and two decodes run side by side.  Every step encodes the whole batch and decodes exactly what it encoded, "
                          "into a ring of image buffers and an output buffer of its pipeline's own; `ms_per_step` = region / K is therefore "
                          "SHORTER than a step's own kernels (`roofline.kernel_ms`: a step takes as long as before, two of them share the "
                          "time) - `one_pipeline` carries rounds 2-3's figure, `serial` the strictly sequential one")),
        },
        "outputs_identical_across_pipelines": m.get("outputs_identical"),
        "bit_exact_vs_reference_golden": golden_check(m, args.streams, samples, ch, bits, args.trials) if rank == 0 else None,
        "encode_msps": round(n_step * world / (m["enc_ms"] * 1e-3) / 1e6, 3),
        "decode_msps": round(n_step * world / (m["dec_ms"] * 1e-3) / 1e6, 3),
        "roofline": dict(
            {"kernel": (stamp_e or {}).get("kernel", "aad::encode_streams_kernel<4, 2, ...>"),
             "bound": "hbm", "achieved": round(enc_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(enc_gbs / HBM_PEAK_GBS, 6)},
            **traffic_fields(stamp_e, note_e, algorithmic),
            **{"algorithmic_bytes_per_launch": algorithmic,
               "bytes_per_sample": round(bps, 4), "samples_per_launch": n_step,
               "kernel_ms": round(m["enc_ms"], 5),
               "kernel_ms_note": "HIP events carried by the encode kernel's own dispatch packet (hipExtLaunchKernelGGL start / stop events via "
                                 "AADHip_ContextSignalNextRun: no packet around the kernel), on every %d-th step of the timed regions; "
                                 "PIPELINED run: the other kernels in flight run beside it (`serial.encode_kernel_ms` is the same "
                                 "kernel with the chip to itself)" % args.event_every if not args.serial else
                                 "HIP events carried by the kernel's own dispatch packet, every %d-th step" % args.event_every,
               "valu": valu_fields(stamp_e, note_e, m["enc_ms"], samples, "encode"),
               "decode_kernel": dict({"kernel": (stamp_d or {}).get("kernel", "aad::decode_split_kernel<4> (quad batches) / aad::decode_blocks_kernel<4> (dense)"),
                                      "achieved": round(dec_gbs, 3), "frac": round(dec_gbs / HBM_PEAK_GBS, 6), "kernel_ms": round(m["dec_ms"], 5),
                                      "valu": valu_fields(stamp_d, note_d, m["dec_ms"], hd.num_samples_per_block, "decode")},
                                     **traffic_fields(stamp_d, note_d, algorithmic)),
               "note": "a 2000-recurrence batch is bound by per-wave instruction issue, not by bandwidth (DESIGN.md \"Kernels\"): `valu` "
                       "carries that bound; `saturated` shows the same kernels on a chip-filling batch"}),
    }

    if not args.serial and world == 1 and args.steps < 100:
        # A short region (the driver passes K = 20) carries the pipeline's fill and drain - the first encode has no decode
        # beside it, the last decode runs behind the last encode: ~50 us per region.  The same pipeline over 200-step regions:
        ml = measure(engine, torch, dist, pcm, param, 200, min(args.warmup, 5), world, args.event_every, decode_engine=decode_engine,
                     repeats=5, max_repeats=5, more_pipelines=more)
        line["steady_state"] = {"value": round(2.0 * n_step * 200 / ml["wall_s"] / 1e6, 3), "unit": "Msamples/s", "steps": 200,
                                "ms_per_step": round(ml["wall_s"] / 200 * 1e3, 5),
                                "note": "not the headline: the same pipelined step timed over regions of 200 steps (median of 5), where the "
                                        "fill and drain of the pipeline (~60-100 us per region) weigh a tenth of what they do at K = %d" % args.steps}
    if more:  # rounds 2-3's headline: ONE pipeline (the encode of step k+1 beside the decode of step k), same K
        mo = measure(engine, torch, dist, pcm, param, args.steps, min(args.warmup, 5), world, args.event_every, decode_engine=decode_engine,
                     repeats=5, max_repeats=5)
        line["one_pipeline"] = {"value": round(2.0 * n_step * world * args.steps / mo["wall_s"] / 1e6, 3), "unit": "Msamples/s", "steps": args.steps,
                                "ms_per_step": round(mo["wall_s"] / args.steps * 1e3, 5),
                                "encode_kernel_ms": round(mo["enc_ms"], 5), "decode_kernel_ms": round(mo["dec_ms"], 5),
                                "note": "one step pipeline (two contexts, two streams): what `value` was in rounds 2 and 3; `--in-flight 1` makes it the headline again"}
    if not args.serial:  # the same K steps with nothing overlapped, for reference
        ks = max(10, args.steps // 4)
        ms_ = measure(engine, torch, dist, pcm, param, ks, min(args.warmup, 3), world, args.event_every, repeats=5, max_repeats=5)
        line["serial"] = {"value": round(2.0 * n_step * world * ks / ms_["wall_s"] / 1e6, 3), "unit": "Msamples/s",
                          "steps": ks, "ms_per_step": round(ms_["wall_s"] / ks * 1e3, 5),
                          "encode_kernel_ms": round(ms_["enc_ms"], 5), "decode_kernel_ms": round(ms_["dec_ms"], 5),
                          "note": "one context, the decode of step k ends before the encode of step k+1 starts"}

    extras = world == 1 and rank == 0 and not args.no_extras

    def guarded(key, leg):
        """An auxiliary leg must not take the headline down: whatever it raises goes on the line under its key, and the run goes on
        (the headline above has been measured; a leg that failed inside a collective of an N > 1 run fails on every rank alike)."""
        try:
            v = leg()
            if v is not None:
                line[key] = v
        except Exception as e:  # noqa: BLE001 - reported, not raised
            line[key] = {"error": repr(e)[:400]}

    def trials2_leg():
        # the reference CLI's default operating point: the same batch with the trial search (src/main.c:45-47)
        p2 = make_parameter(ch, bits, mbs, 48000, False, 2)
        m2 = measure(engine, torch, dist, pcm, p2, max(10, args.steps // 4), 3, 1, 1, keep=True, decode_engine=decode_engine, repeats=5, max_repeats=5,
                     more_pipelines=more)
        e2 = n_step * bps / (m2["enc_ms"] * 1e-3) / 1e9
        return {
            "workload": "the headline batch with num_encode_trials = 2 (reference CLI default, src/main.c:45-47)",
            "value": round(2.0 * n_step * max(10, args.steps // 4) / m2["wall_s"] / 1e6, 3), "unit": "Msamples/s",
            "ms_per_step": round(m2["wall_s"] / max(10, args.steps // 4) * 1e3, 5),
            "encode_ms": round(m2["enc_ms"], 5), "decode_ms": round(m2["dec_ms"], 5),
            "encode_msps": round(n_step / (m2["enc_ms"] * 1e-3) / 1e6, 1), "decode_msps": round(n_step / (m2["dec_ms"] * 1e-3) / 1e6, 1),
            "encode_frac": round(e2 / HBM_PEAK_GBS, 6),
            "bit_exact_vs_reference_golden": golden_check(m2, args.streams, samples, ch, bits, 2),
        }

    def configs_leg():
        cfgs = [
            ("cfg2(ii) 1000 stereo 4-bit streams x 16 blocks", 1000, 992 * 16, 2, 4, 0, 5, "corpus", "cfg2ii"),
            ("cfg2(iii) 1 stereo 4-bit stream x 1000 blocks (serial worst case: 2 encode recurrences)", 1, 992 * 1000, 2, 4, 0, 2, "corpus"),
            ("cfg2(iii) with trials 2", 1, 992 * 1000, 2, 4, 2, 1, "corpus"),
            ("cfg4 10000 x 8-channel 3-bit one-block segments", 10000, 292, 8, 3, 0, 10, "eight", "cfg4_3bit"),
            ("cfg4 10000 x 8-channel 2-bit one-block segments", 10000, 444, 8, 2, 0, 10, "eight", "cfg4_2bit"),
            ("cfg5 per-GPU shard: 1250 stereo 4-bit files x 10 blocks", 1250, 9920, 2, 4, 0, 5, "corpus", "cfg5_shard"),
        ]
        rows = []
        for c in cfgs:
            try:
                rows.append(config_entry(engine, torch, dist, *c))
            except Exception as e:  # noqa: BLE001 - this row only
                rows.append({"config": c[0], "error": repr(e)[:400]})
        return rows

    if extras:
        guarded("trials2", trials2_leg)
        guarded("end_to_end", lambda: end_to_end(engine, torch, pcm_np, param))
        guarded("configs", configs_leg)

    def saturated_leg():  # per-kernel figures of ONE GPU: rank 0 alone, no collective inside
        big_streams = args.saturated_streams  # 262144 stereo streams = 8192 waves = 8 per SIMD (dense mapping)
        reps = -(-big_streams // args.streams)
        big = pcm.repeat((reps, 1, 1))[:big_streams].contiguous()
        ms = measure(engine, torch, dist, big, param, 5, 1, 1)
        del big
        nb = big_streams * samples * ch
        sat = {
            "workload": "%d stereo streams x %d samples/ch per launch (the step batch tiled)" % (big_streams, samples),
            "encode_msps": round(nb / (ms["enc_ms"] * 1e-3) / 1e6, 1), "decode_msps": round(nb / (ms["dec_ms"] * 1e-3) / 1e6, 1),
            "encode_gbs": round(nb * bps / (ms["enc_ms"] * 1e-3) / 1e9, 2), "decode_gbs": round(nb * bps / (ms["dec_ms"] * 1e-3) / 1e9, 2),
            "encode_frac": round(nb * bps / (ms["enc_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "decode_frac": round(nb * bps / (ms["dec_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "algorithmic_bytes_per_launch": int(round(nb * bps)),
        }
        for role, kms, per_rec in (("encode", ms["enc_ms"], samples), ("decode", ms["dec_ms"], hd.num_samples_per_block)):
            st, note = pmc_stamp("saturated", role, big_streams, samples)
            sat[role] = dict({"kernel": (st or {}).get("kernel"), "kernel_ms": round(kms, 5)},
                                           **traffic_fields(st, note, int(round(nb * bps))),
                                           **{"valu": valu_fields(st, note, kms, per_rec, role)})

        if extras:
            # the other fast-path geometries on chip-filling batches of one-block streams (524 288 recurrences each),
            # round trip checked on the device: the decode of the encode equals its neighbour tile's
            geo = []
            for g_bits, g_ch in ((3, 2), (2, 2), (4, 1), (3, 1), (2, 1)):
                g_param = make_parameter(g_ch, g_bits, mbs, 48000, False, 0)
                g_samples = {4: 1984, 3: 2632, 2: 3960}[g_bits] // g_ch
                g_streams = 524288 // g_ch
                tile = torch.from_numpy(synth_pcm(1000, g_samples, g_ch, seed=1234)).cuda()
                g_pcm = tile.repeat((-(-g_streams // 1000), 1, 1))[:g_streams].contiguous()
                gm = measure(engine, torch, dist, g_pcm, g_param, 3, 1, 1, keep=False)
                g_hd = gm["header"]
                g_n = g_streams * g_samples * g_ch
                g_bps = algorithmic_bytes_per_sample(g_ch, g_hd.block_size, g_hd.num_samples_per_block)
                geo.append({"bits": g_bits, "channels": g_ch, "streams": g_streams, "samples_per_channel": g_samples,
                            "encode_msps": round(g_n / (gm["enc_ms"] * 1e-3) / 1e6, 1), "decode_msps": round(g_n / (gm["dec_ms"] * 1e-3) / 1e6, 1),
                            "encode_frac": round(g_n * g_bps / (gm["enc_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                            "decode_frac": round(g_n * g_bps / (gm["dec_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)})
                del g_pcm, tile
            sat["geometries"] = geo
        return sat

    if not args.no_saturated and rank == 0:
        guarded("saturated", saturated_leg)

    if world > 1 and args.blocks == 1:
        # every rank takes part; a failure inside a collective is every rank's alike and goes on the line instead of ending the run
        guarded("config5", lambda: config5_batched_files(engine, torch, dist, rank, world))
    elif world == 1 and args.blocks == 1 and not args.no_config5:
        # One GPU: the same leg under a ONE-rank RCCL process group with the collectives forced (a world of one would
        # skip them), so that the code an N > 1 run depends on - init_process_group(device_id=), broadcast and
        # gather of device tensors, the float64 all_reduce and the barriers of measure() - executes on this box
        # too.  A failure to bring RCCL up is reported on the line, it does not take the headline down.
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
            try:
                c5 = config5_batched_files(engine, torch, dist, 0, 1, force_collectives=True)
                # the N > 1 timing path of measure() (barriers, MAX all_reduce on a device tensor), a short rehearsal
                mr = measure(engine, torch, dist, pcm, param, 20, 3, 1, args.event_every, decode_engine=decode_engine,
                             repeats=3, max_repeats=3, collective=True, more_pipelines=more)
                c5["collective_timing_rehearsal"] = {
                    "value": round(2.0 * n_step * 20 / mr["wall_s"] / 1e6, 3), "unit": "Msamples/s", "steps": 20, "regions": mr["regions"],
                    "note": "the headline step timed the way an N > 1 run times it: dist.barrier() on both sides of every region and "
                            "the region time MAX-reduced over the (one) rank as a float64 device tensor through RCCL"}
                c5["world_size"] = 1
                c5["collectives"] = "forced (BatchCodec(force_collectives=True)): broadcast + gather run through RCCL with one rank"
                line["config5"] = c5
            finally:
                dist.destroy_process_group()
        except Exception as e:  # noqa: BLE001 - reported, not raised
            line["config5"] = {"error": repr(e)[:400], "backend": "nccl", "world_size": 1}

    def cpu_leg():
        trials_list = [args.trials] + ([2] if extras and args.trials != 2 else [])
        cpu = cpu_baseline(pcm_np, bits, mbs, trials_list)
        out = cpu[args.trials]
        out["gpu_over_cpu"] = round(value / cpu[args.trials]["value"], 1)
        if world > 1:
            out["gpu_over_cpu_note"] = "whole job (%d GPUs) over ONE host core" % world
        if isinstance(line.get("trials2"), dict) and "value" in line["trials2"] and 2 in cpu:
            line["trials2"]["cpu_baseline"] = cpu[2]
            line["trials2"]["gpu_over_cpu"] = round(line["trials2"]["value"] / cpu[2]["value"], 1)
        if isinstance(line.get("end_to_end"), dict) and "host_memory_api" in line["end_to_end"]:
            line["end_to_end"]["host_memory_api"]["encode_over_cpu_single_thread"] = round(
                line["end_to_end"]["host_memory_api"]["encode_msps"] / cpu[args.trials]["encode_msps"], 1)
            line["end_to_end"]["host_memory_api"]["decode_over_cpu_single_thread"] = round(
                line["end_to_end"]["host_memory_api"]["decode_msps"] / cpu[args.trials]["decode_msps"], 1)
        return out

    if rank == 0 and not args.no_cpu_baseline:  # N > 1 too: the host cores are the same box's; the other ranks wait at the barrier below
        guarded("cpu_baseline", cpu_leg)
    if second_pair:
        for e_ in second_pair:
            e_.close()
    engine.close()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.barrier()  # rank 0's CPU baseline and saturated leg end before any rank tears the group down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
