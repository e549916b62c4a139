/*
 * aad_hip.h - additive batched C-ABI of the MI355X AAD engine (SURVEY.md section 8b "New (additive) ABI").
 *
 * The reference has no batched or device-resident entry point: its only data path is one stream
 * per call through AADEncoder_EncodeWhole (src/aad_encoder.c:814-891), AADDecoder_DecodeWhole
 * (src/aad_decoder.c:478-538) and AADDecoder_DecodeBlock (src/aad_decoder.c:321-475).  A GPU is
 * only useful when many independent units are in flight (encode: stream x channel; decode:
 * block x channel), so this header exposes exactly those three loops over MANY streams:
 *
 *   AADHip_EncodePlanRun  == for each stream: the block loop of AADEncoder_EncodeWhole
 *                            (EncodeHeader + [SearchBestProcessor] + EncodeBlock per block)
 *   AADHip_DecodePlanRun  == for each stream: the block loop of AADDecoder_DecodeWhole
 *                            (DecodeBlock per block), every block decoded independently
 *
 * Plain C: pointers and sizes only.  "Device" pointers are HIP device addresses (hipMalloc or
 * any allocator sharing the HIP context, e.g. a torch tensor's data_ptr()).  The legacy
 * AADEncoder_ / AADDecoder_ symbols are implemented on top of these with a batch of one.
 *
 * Device data layout
 *   PCM   : int16, channel-interleaved frames, one contiguous run per stream.
 *   .aad  : the exact file image per stream - 31-byte header followed by the blocks - byte for
 *           byte what the reference writes for the same samples and parameters.
 */
#ifndef AAD_HIP_H_INCLUDED
#define AAD_HIP_H_INCLUDED

#include <stdint.h>
#include "aad.h"
#include "aad_encoder.h"

#define AAD_HIP_MAX_NUM_CHANNELS 8 /* container extension; the legacy API keeps AAD_MAX_NUM_CHANNELS */

struct AADHipContext;    /* device + stream + device-side tables */
struct AADHipEncodePlan; /* uploaded stream table + launch geometry for one parameter set */
struct AADHipDecodePlan;

/* One stream of a batch.  Offsets are relative to the buffers handed to ...PlanRun. */
struct AADHipStreamDesc {
  uint64_t pcm_offset;  /* index of the stream's first int16 in the PCM buffer */
  uint64_t data_offset; /* byte offset of the stream's .aad image in the data buffer */
  uint64_t data_size;   /* encode: capacity in bytes; decode: bytes present (a short last block is fine) */
  uint32_t num_samples; /* samples per channel */
  uint32_t reserved;
};

/* Predictor state of one stream x channel, carried across calls exactly like the reference's
 * struct AADEncodeProcessor (src/aad_encoder.c:10-15) inside a reused encoder handle. */
struct AADHipLaneState {
  int32_t weight[4];
  int32_t history[4];
  int32_t stepsize_index;
  int32_t quantize_error;
};

#ifdef __cplusplus
extern "C" {
#endif

/* number of usable HIP devices (0 when there is none; never fails) */
int32_t AADHip_GetDeviceCount(void);

/* Create a context on `device_index`.  `hip_stream` is a hipStream_t the caller owns, or NULL
 * to let the context create (and later destroy) its own stream.  All ...Run calls are
 * asynchronous on that stream. */
AADApiResult AADHip_ContextCreate(int32_t device_index, void *hip_stream, struct AADHipContext **context);
void AADHip_ContextDestroy(struct AADHipContext *context);
AADApiResult AADHip_ContextSynchronize(struct AADHipContext *context);
/* text of the last HIP failure seen by this context ("" if none); valid until the next call */
const char *AADHip_ContextLastError(const struct AADHipContext *context);

/* Cross-stream ordering and kernel timing without packets of their own.  `hip_start_event` / `hip_stop_event` (hipEvent_t the
 * caller owns; either may be NULL, both NULL withdraws) are recorded when the work of the NEXT AADHip_EncodePlanRun /
 * AADHip_DecodePlanRun / AADHip_ReconstructPlanRun call on this context starts / is done - one-shot: the run takes them.  An
 * encode or decode plan run is one kernel, and the events ride on that kernel's own dispatch (hipExtLaunchKernelGGL's start and
 * stop events) instead of on barrier packets around it: a hipEventRecord behind every launch of a back-to-back sequence costs
 * the queue 2.9 us per launch on MI355X, the attached event nothing (profiles/r03_microbench_event_gap.txt), and
 * hipEventElapsedTime between the two is the kernel's own duration.  Another stream waits for the stop event with
 * hipStreamWaitEvent as usual.  A run that fails leaves the events unrecorded; the host-memory calls (...Batch, the legacy API)
 * do not look at them. */
AADApiResult AADHip_ContextSignalNextRun(struct AADHipContext *context, void *hip_start_event, void *hip_stop_event);
/* NOT available when the process runs with ROC_SYSTEM_SCOPE_SIGNAL=0 (a ROCm runtime setting that gives kernel dispatches
 * device-scope completion signals): the stop event IS the kernel's completion signal, and another stream's
 * hipStreamWaitEvent on it would never return.  A context created under that setting refuses the call - AAD_APIRESULT_NG,
 * AADHip_ContextLastError says why - instead of letting the caller hang; withdrawing (both NULL) always succeeds, and
 * hipEventRecord behind the run remains the portable way.  AADHip_SignalNextRunSupported: 1, or 0 under that setting
 * (reads the environment, needs no device). */
int32_t AADHip_SignalNextRunSupported(void);

/* Launch options of a context.  Defaults: the environment variables AAD_HIP_MAPPING
 * (auto | dense | quad | quad-fused | dense-tiled), AAD_HIP_TRIAL_LANES (dual | single),
 * AAD_HIP_STAGING_THREADS (1..8) and AAD_HIP_TILE_KBYTES, read ONCE when the context is created; the library does not read them
 * again (the measurement aids of INTEGRATION.md section 4 - AAD_HIP_ENCODE_RING, ..._LDS_PAD - are not options and never change a byte).  An option holds for every later
 * ...Run / ...Batch call of the context; set it from the thread that owns the context. */
enum AADHipOption {
  AAD_HIP_OPTION_LANE_MAPPING = 0, /* enum AADHipLaneMapping */
  AAD_HIP_OPTION_TRIAL_LANES = 1,  /* enum AADHipTrialLanes */
  /* Threads that copy between the caller's buffers and the pinned staging blocks in the host-memory
   * entry points (...Batch, EncodeWhole/DecodeWhole), the caller's own included: 0 = by core count
   * (8 from thirty-two cores, 4 from eight, 2 from four), 1 = the caller alone (no helper thread is ever started),
   * up to 8.  Helpers start at the first chunk of a megabyte or more and end in ContextDestroy. */
  AAD_HIP_OPTION_STAGING_THREADS = 2,
  /* Budget, in KiB, of one tile of the host-memory entry points (input + output bytes that travel
   * together; see DESIGN.md "host-memory path"): 0 = built in (batches up to 16 MiB go as one tile,
   * larger ones in tiles of about 16 MiB).  A tile never holds less than one block of one stream.  Default from
   * AAD_HIP_TILE_KBYTES.  Results do not depend on it. */
  AAD_HIP_OPTION_TILE_KBYTES = 3,
  /* Order of the fp64 sums behind the reconstruction modes' RMSE / MSD (AADHip_Reconstruct*): 0 = a fixed tree on the
   * device, with the reference's channel-major sequential order taken per stream only when the tree's result lies so close
   * to a rounding boundary of the six decimals `aad -c` prints that the order could show (the printed line is the
   * reference's either way); 1 = always the reference's order (bit-identical doubles, one lane per stream: slow).
   * Default from AAD_HIP_COMPARE_ORDER (auto | sequential). */
  AAD_HIP_OPTION_COMPARE_ORDER = 4
};
enum AADHipLaneMapping {
  AAD_HIP_LANE_MAPPING_AUTO = 0,      /* by batch size (the default) */
  AAD_HIP_LANE_MAPPING_DENSE = 1,     /* one lane per recurrence */
  AAD_HIP_LANE_MAPPING_QUAD = 2,      /* four lanes per recurrence; decode: step-index scan on other waves */
  AAD_HIP_LANE_MAPPING_QUAD_FUSED = 3, /* four lanes per recurrence; decode: one fused kernel */
  AAD_HIP_LANE_MAPPING_DENSE_TILED = 4 /* one lane per recurrence, memory moved in whole sectors / lines through LDS where the layout allows (else: dense) */
};
enum AADHipTrialLanes {
  AAD_HIP_TRIAL_LANES_DUAL = 0,  /* trial search on the quad mapping: a second group of lanes runs the probe and encodes every candidate beside the chain
                                  * (batches of up to 5120 recurrences, where it pays; larger ones take the other layout) */
  AAD_HIP_TRIAL_LANES_SINGLE = 1 /* both strands on the same lanes */
};
AADApiResult AADHip_ContextSetOption(struct AADHipContext *context, int32_t option, int32_t value);

/* bytes of the .aad image of a stream (header + full blocks + short tail); 0 on a bad parameter.
 * Same arithmetic as the write_offset AADEncoder_EncodeWhole ends with (src/aad_encoder.c:881-889). */
uint64_t AADHip_CalculateEncodedSize(const struct AADEncodeParameter *parameter, uint32_t num_samples);

/* ---- encode ------------------------------------------------------------------------------ */

/* Validates `parameter` like AADEncoder_SetEncodeParameter + AADEncoder_EncodeHeader would
 * (INVALID_FORMAT), checks every stream's capacity (INSUFFICIENT_BUFFER) and uploads the table.
 * `streams` is a host array. */
AADApiResult AADHip_EncodePlanCreate(
    struct AADHipContext *context, const struct AADEncodeParameter *parameter,
    uint32_t num_streams, const struct AADHipStreamDesc *streams,
    struct AADHipEncodePlan **plan);
void AADHip_EncodePlanDestroy(struct AADHipEncodePlan *plan);

/* Encode every stream of the plan.  device_state: NULL for fresh encoders (zero weights, zero
 * step index), else num_streams * num_channels records read before and written after the run. */
AADApiResult AADHip_EncodePlanRun(
    struct AADHipEncodePlan *plan, const int16_t *device_pcm, uint8_t *device_data,
    struct AADHipLaneState *device_state);

/* ---- decode ------------------------------------------------------------------------------ */

/* `format` supplies channels / bits / block geometry / channel process method for the whole
 * batch (its num_samples field is ignored; each stream's count comes from `streams`).
 * Validated like AADDecoder_SetHeader (src/aad_decoder.c:173-225) with the channel limit raised
 * to AAD_HIP_MAX_NUM_CHANNELS.  has_file_header: non-zero when each image starts with the
 * 31-byte file header (DecodeWhole), zero when data_offset points at a bare block (DecodeBlock). */
AADApiResult AADHip_DecodePlanCreate(
    struct AADHipContext *context, const struct AADHeaderInfo *format, int32_t has_file_header,
    uint32_t num_streams, const struct AADHipStreamDesc *streams,
    struct AADHipDecodePlan **plan);
void AADHip_DecodePlanDestroy(struct AADHipDecodePlan *plan);

AADApiResult AADHip_DecodePlanRun(
    struct AADHipDecodePlan *plan, const uint8_t *device_data, int16_t *device_pcm);

/* ---- host-memory convenience (stage -> run -> copy back, synchronous) ---------------------- */

/* pcm[i]: num_samples[i] interleaved frames; data[i]: data_capacity[i] bytes; output_size[i]
 * (may be NULL) receives the image size.  state: NULL for fresh encoders, else a host array of
 * num_streams * num_channels records, read before and written after the run. */
AADApiResult AADHip_EncodeBatch(
    struct AADHipContext *context, const struct AADEncodeParameter *parameter,
    uint32_t num_streams, const int16_t *const *pcm, const uint32_t *num_samples,
    uint8_t *const *data, const uint64_t *data_capacity, uint64_t *output_size,
    struct AADHipLaneState *state);

/* data[i]/data_size[i]: .aad images (all of one format); pcm[i] must hold
 * pcm_capacity_frames[i] >= header.num_samples frames.  decoded_frames[i] (may be NULL) receives
 * the frames written: header.num_samples, or fewer when the image ends early - the reference's
 * block walk stops when the bytes run out (src/aad_decoder.c:514) and so does this. */
AADApiResult AADHip_DecodeBatch(
    struct AADHipContext *context, uint32_t num_streams,
    const uint8_t *const *data, const uint64_t *data_size,
    int16_t *const *pcm, const uint32_t *pcm_capacity_frames, uint32_t *decoded_frames);

/* ---- reconstruction modes: encode -> decode -> residual / statistics, all on the device ------ */

/* What the reference CLI's -r / -g / -c modes compute (src/main.c:275-503) for MANY inputs
 * without the encoded images or the reconstructed PCM ever leaving HBM.  Per stream:
 *   images  = AADEncoder_EncodeWhole(pcm)          (src/main.c:319-325)
 *   out     = AADDecoder_DecodeWhole(images)       (src/main.c:328-332)        -> `aad -r`
 *   out     = int16 wrap of pcm - out              (src/main.c:419-423)        -> `aad -g`
 *   stats   = RMSE / MSD / MaxAE as `aad -c` prints them (src/main.c:476-497)  -> `aad -c`
 */
enum AADHipReconstructOutput {
  AAD_HIP_RECONSTRUCT_DECODED = 0, /* out holds the reconstructed PCM */
  AAD_HIP_RECONSTRUCT_RESIDUAL = 1 /* out holds original minus reconstructed */
};

/* the three numbers of `aad -c`'s "RMSE:%f MSD:%f MaxAE:%f" line, per stream */
struct AADHipErrorStats {
  double rms_error;
  double mean_abs_error;
  double max_abs_error;
};

struct AADHipReconstructPlan;

/* `streams` as for AADHip_EncodePlanCreate: pcm_offset addresses BOTH the input and the output
 * PCM buffer, data_offset / data_size the scratch buffer that receives the .aad images. */
AADApiResult AADHip_ReconstructPlanCreate(
    struct AADHipContext *context, const struct AADEncodeParameter *parameter,
    uint32_t num_streams, const struct AADHipStreamDesc *streams,
    struct AADHipReconstructPlan **plan);
void AADHip_ReconstructPlanDestroy(struct AADHipReconstructPlan *plan);

/* device_stats: NULL, or num_streams records.  Fresh encoders (as the CLI creates per file). */
AADApiResult AADHip_ReconstructPlanRun(
    struct AADHipReconstructPlan *plan, const int16_t *device_pcm, uint8_t *device_data,
    int16_t *device_out, int32_t output_kind, struct AADHipErrorStats *device_stats);

/* host-memory form.  out_pcm: NULL (statistics only - nothing but 24 bytes per stream comes
 * back over PCIe) or per-stream buffers of num_samples[i] frames; stats: NULL or num_streams.
 * Any batch size: the compute runs over a WAVE of whole streams resident on the device (one ReconstructPlanRun: the encoders'
 * block chains side by side, the statistics summed per stream as the plan form does), the PCM goes up and the output comes down
 * through the context's pinned blocks in chunks of the tile budget (AAD_HIP_OPTION_TILE_KBYTES), and a batch beyond three
 * quarters of the device's free memory is run as several waves of consecutive streams.  Results do not depend on any of it. */
AADApiResult AADHip_ReconstructBatch(
    struct AADHipContext *context, const struct AADEncodeParameter *parameter,
    uint32_t num_streams, const int16_t *const *pcm, const uint32_t *num_samples,
    int32_t output_kind, int16_t *const *out_pcm, struct AADHipErrorStats *stats);

#ifdef __cplusplus
}
#endif

#endif /* AAD_HIP_H_INCLUDED */
