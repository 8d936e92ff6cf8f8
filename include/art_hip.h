/* art_hip.h — MI355X extensions to the reference C API (additive; nothing here exists in the
 * reference).  Device-pointer entry points take HIP device pointers, enqueue their work on the
 * context's stream and return without synchronising: the frame counts in the result are computed
 * on the host by replaying the reference's scalar position state machine
 * (reference resampler.c:487-537) in closed form, so they are available immediately.
 *
 * Devices.  An ordinary context lives on the HIP device that is current when it is created and makes
 * that device current around every call it serves (so `torch.cuda.set_device(LOCAL_RANK)` before
 * resampleInit gives one process per GPU).  A context created with
 * RESAMPLE_MULTITHREADED spreads its channels over several devices INSIDE the one context — the
 * reference's one-worker-per-channel fan-out (reference resampler.c:185-186, :442-470) with GPUs
 * for threads: see artamdSetDevices below.
 */
#ifndef ARTAMD_ART_HIP_H
#define ARTAMD_ART_HIP_H

#include "resampler.h"
#include "biquad.h"
#include "decimator.h"
#include <stddef.h>
#include "stretch.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime ---- */
int artamdDeviceCount (void);                       /* 0 when no usable gfx950 device / HIP runtime */
/* device memory and transfers for callers of the device-pointer entry points that bring no GPU runtime of their own (a tool in C,
 * or tools/art_gpu.py without torch): thin names over the HIP runtime; the transfers are asynchronous on `hipStream` (NULL: the
 * null stream) and return 0 on success */
void *artamdDeviceAlloc (size_t bytes);
void artamdDeviceFree (void *d_ptr);
int artamdUpload (void *d_dst, const void *h_src, size_t bytes, void *hipStream);
int artamdDownload (void *h_dst, const void *d_src, size_t bytes, void *hipStream);
int artamdDeviceZero (void *d_dst, size_t bytes, void *hipStream);
int artamdStreamSynchronize (void *hipStream);
const char *artamdVersion (void);

/* Devices that RESAMPLE_MULTITHREADED contexts created from now on spread their channels over (contiguous, balanced
 * channel slices, one shard per listed device; a device may be listed more than once).  count <= 0 restores the default:
 * the environment's ARTAMD_DEVICES="0,1,2,..." list, else every visible device.  With a single device the flag has no
 * effect unless ARTAMD_SHARDS=n forces n shards (several per device: how the sharded path is tested on a one-GPU box).
 * Every shard has its own stream, history and filter-bank replica; no sample crosses between shards.  Host-pointer calls
 * de-interleave each shard's channel slice on the way into its HBM; device-pointer calls take buffers on the device the
 * context was created on and move the slices peer-to-peer.  Returns 0, or -1 for a device that does not exist. */
int artamdSetDevices (const int *devices, int count);
int resampleHipGetDevice (Resample *cxt);            /* the context's device (sharded: where device-pointer buffers are expected) */
int resampleHipNumShards (Resample *cxt);            /* 0 for an ordinary context */
int resampleHipShardInfo (Resample *cxt, int shard, int *device, int *firstChannel, int *numChannels);   /* 0, or -1: no such shard */

/* ---- resampler ---- */
/* default: the null stream.  Work already enqueued on the previous stream is drained before the switch (history and
 * scratch buffers are shared between calls); biquadBankSetStream, decimateHipSetStream and stretchHipSetStream do the same. */
void resampleHipSetStream (Resample *cxt, void *hipStream);
void resampleHipSynchronize (Resample *cxt);
/* kernel selection for tests and comparisons: 0 = automatic, 1 = general kernel, 2 = matrix-core path wherever the ratio is
 * rational (the persistent streaming kernel for regular launches, the one-tile-per-workgroup kernel otherwise; falls back to
 * 1 elsewhere), 5 = as 2 but always the one-tile-per-workgroup f32 kernel, 6 = as 2 but never the fixed-point kernel: the f32
 * streaming kernel for regular launches (5 and 6 give the same bits; the fixed-point kernel rounds once per output and
 * differs from them in the last place), 7 = as 2 and the fixed-point kernel wherever it can run (automatically it takes
 * filters of 512 taps and more in calls of about a billion output-sample taps and more, where it is the faster one).
 * ARTAMD_KERNEL=<n> in the environment is the preference every NEW context starts with (programs that cannot make this call:
 * the reference's own art / artest binaries on this library, tests/test_gpu_pcm_default_mode.py). */
void resampleHipSetKernel (Resample *cxt, int which);
/* (6 on a fixed-ratio stream — resampleFixedRatioInit — also makes the output independent of how the input is cut into calls, bit for bit, as the
 * reference's is: every launch runs the one kernel on tiles anchored on the stream's canonical period.  Calls of at least one period of outputs;
 * device-pointer input aligned to a frame (1 - 2 channels) / 16 bytes (4 and more).  resampler.h's header; tests/test_gpu_cut_invariance.py) */
/* THE CUT-INVARIANT STREAM POLICY (round 6).  The reference's fixed-ratio output is bitwise independent of how the input is cut into calls
 * (resampler.c:323-335, 533-535: `artest -e -b256 | -b1000 | -b4096 | -b65536`, one checksum).  In the default mode this library picks a kernel per
 * CALL (general / f32 matrix cores, K split or not / fixed point), each inside the parity bar with its own last bits.  With this policy on, the
 * arithmetic is chosen per STREAM: every launch of a rational-ratio stream — big, small, shorter than one period, planar or interleaved — runs on the
 * f32 matrix-core streaming kernel, un-split, anchored on the stream's canonical period (an output sits in the same tile row, on the same K chunks
 * and flush points whichever call brought it), and the outputs only a flush can make (the stream's last T/2 x ratio) on the general kernel, whose
 * outputs never depend on the cut either: the same bits for ANY cut into calls, host or device buffers.  A launch that cannot run anchored (a
 * nearest-filter stream with a slot on a half step, a device buffer not aligned to 16 bytes / one frame, a call of several million frames whose
 * position drift exceeds the kernels' tolerance) is given to the general kernel and COUNTED: resampleHipCutInvariantFallbacks () == 0 says the
 * guarantee held for every output so far.  Equivalent: resampleHipSetKernel (cxt, 9), ARTAMD_KERNEL=9.
 * What it costs against the library's own choice (tools/bench_cut_invariant.py, profiles/r6_cut_invariant.txt): nothing is free — small calls lose the
 * general kernel's short launch, mid-sized calls of long filters the K split, big calls the fixed-point kernel — which is why it is a context
 * setting and not the default.  RESAMPLE_STRICT_ORDER (bit-exact reference order) and preference 1 (the general kernel alone) are cut-invariant too.
 * (4-byte samples.  The 8-byte build has no kept rows: there the setting is preference 2 — use RESAMPLE_STRICT_ORDER or preference 1.) */
void resampleHipSetCutInvariant (Resample *cxt, int on);
unsigned int resampleHipCutInvariantFallbacks (Resample *cxt);
/* The streaming matrix kernels keep their filter rows across the calls of a context (built once for the stream's canonical period; every later
 * launch is anchored on that period: DESIGN.md 4.1) — on by default.  Off: every launch builds its rows from its own positions and anchors its
 * tiles on its own first output, as before round 5 (comparisons of kernel forms bit for bit; ARTAMD_ROWS_CACHE=0 does it for a whole process).
 * Either way a sample is within the parity bar; the two differ in the last place of a few per cent of the samples. */
void resampleHipKeepRows (Resample *cxt, int on);
int  resampleHipLastKernel (Resample *cxt);          /* which kernel produced the bulk of the last call */
/* the matrix-core path's fixed-point kernel (regular launches, 4-byte samples): 0 = the last call did not use it, 1 = it ran,
 * 2 = it was enqueued and stood down for the f32 kernel's tile loop (an infinity or a NaN among the frames the launch reads: any finite
 * amplitude is held, the block exponents follow the channel's peak).
 * *pairsPerChunk (may be NULL): digit-pair products issued per 32-tap chunk, 5 .. 13 (5 where both upper digit planes of the rows are zero, + 4 for each that is not).  Synchronises. */
int  resampleHipLastFixedPoint (Resample *cxt, double *pairsPerChunk);
/* the form of the fixed-point kernel the last call's last launch was given to: 0 none, 1 fir_i8_stream_kernel (register-staged: 1 and 2
 * channels, ARTAMD_I8_DMA=0), 2 fir_i8_dma_kernel (LDS-DMA staging, 32-slot tiles), 3 fir_i8_slab_kernel (64 x 256 tiles, big launches).
 * All three leave the same bits. */
int  resampleHipLastFixedPointKernel (Resample *cxt);
unsigned int resampleHipLastHandedBack (Resample *cxt);   /* outputs the matrix-core kernels evaluated at their own exact position, off their slot's canonical pattern, so far */
/* HIP-event timing of the dominant FIR kernel only (events recorded on the context's stream immediately
 * before and after that kernel's launch; the fix-up and history kernels are outside the bracket).  Enable, run calls, then read: returns accumulated kernel milliseconds and the launch count
 * since timing was (re-)enabled; the read synchronises. */
void resampleHipSetTiming (Resample *cxt, int enable);
double resampleHipReadTiming (Resample *cxt, int *numLaunches);
/* milliseconds the launches covered by the last resampleHipReadTiming spent BEFORE their dominant kernel (table / staging
 * passes of the matrix-core paths — peak pass + digit-plane pass of the fixed-point kernel — and the gaps between them) */
double resampleHipReadPrepTiming (Resample *cxt);

ResampleResult resampleProcessInterleavedDevice (Resample *cxt, const artsample_t *d_input, int numInputFrames,
                                                 artsample_t *d_output, int numOutputFrames, double ratio);
ResampleResult resampleProcessAndFlushInterleavedDevice (Resample *cxt, const artsample_t *d_input, int numInputFrames,
                                                         artsample_t *d_output, int numOutputFrames, double ratio);
/* Many independent streams, one launch: results [i] and the samples in d_outputs [i] are exactly what
 * resampleProcessInterleavedDevice (cxts [i], d_inputs [i], numInputFrames [i], d_outputs [i], numOutputFrames [i], ratios [i])
 * would have produced.  Contexts whose call the general kernel runs (any ratio per context, default or EXTEND mode, an
 * ordinary call, on the stream of cxts [0]) share launches — a service with hundreds of small-block streams is
 * launch-bound one call at a time; all other calls (strict mode, endpoint extrapolation, calls large enough for the
 * matrix-core path, contexts on other streams) are simply made one by one.  A context may appear only once.  Asynchronous
 * like the single call: counts are returned at once, the samples land on the stream.  Returns 0, or -1 if a launch failed. */
int resampleProcessBatchInterleavedDevice (Resample *const *cxts, int n, const artsample_t *const *d_inputs, const int *numInputFrames,
                                           artsample_t *const *d_outputs, const int *numOutputFrames, const double *ratios,
                                           ResampleResult *results);
/* planar device buffers: channel c at d_input + c*inputPitch (in samples), likewise output; a pitch of 0 on either side means that side
 * is interleaved.  Big calls are transposed through the context's interleaved staging on the device (the matrix-core kernels read
 * interleaved frames): the same samples as the interleaved entry point's, bit for bit */
ResampleResult resampleProcessPlanarDevice (Resample *cxt, const artsample_t *d_input, long inputPitch, int numInputFrames,
                                            artsample_t *d_output, long outputPitch, int numOutputFrames, double ratio);

/* ---- host-only building blocks (no GPU needed; used by the CPU test-suite) ---- */
/* (numFilters+1) x numTaps bank exactly as resampleInit builds it (reference resampler.c:149-168, 1090-1133) */
void artamdBuildFilterBank (int numTaps, int numFilters, double lowpassRatio, int flags, artsample_t *bank);

typedef struct {
    unsigned int first_output;      /* outputs [first_output, next.first_output) belong to this segment */
    int lin_base;                   /* ring index + lin_base = index into (history ++ new input) */
    double base_offset;             /* outputOffset valid for this ring epoch */
} ArtamdSegment;

typedef struct {
    int numTaps, numFilters, flags, inputIndex, floorActive;
    double outputOffset, fixedRatio;
} ArtamdPosition;

/* Replay one process call on explicit position state: fills the result, advances *pos and writes up
 * to maxSegments ring-epoch segments.  Returns the number of segments the call needs (may exceed
 * maxSegments; then only the first maxSegments were written).  *linFloor receives the linear index
 * below which history reads as silence (INT_MIN when unrestricted). */
int artamdPlanCall (ArtamdPosition *pos, int numInputFrames, int numOutputFrames, double ratio,
                    ResampleResult *result, ArtamdSegment *segments, int maxSegments, int *linFloor);

/* ---- biquad: device-resident bank of per-channel section chains ---- */
typedef struct artamd_biquad_bank BiquadBank;
/* sections[c*numSections + s] is section s of channel c (state and coefficients are copied) */
BiquadBank *biquadBankCreate (const Biquad *sections, int numChannels, int numSections);
/* the same bank spread over the devices of artamdSetDevices () / ARTAMD_DEVICES (ARTAMD_SHARDS forces the count; at least two
 * channels per device otherwise), contiguous channel slices, one stream per device, the way a RESAMPLE_MULTITHREADED resampler
 * and a DECIMATE_MULTITHREADED decimator spread: same entry points, same bits; an ordinary bank when there is one device */
BiquadBank *biquadBankCreateMulti (const Biquad *sections, int numChannels, int numSections);
int biquadBankShardCount (BiquadBank *bank);         /* 0: an ordinary bank */
void biquadBankSetStream (BiquadBank *bank, void *hipStream);
/* in-place over interleaved device frames [numFrames][numChannels]; asynchronous */
void biquadBankApplyInterleavedDevice (BiquadBank *bank, artsample_t *d_buffer, int numFrames);
void biquadBankRead (BiquadBank *bank, Biquad *sections);      /* synchronises; copies state back */
void biquadBankFree (BiquadBank *bank);
/* Long runs of filters that forget their state within 1,024 frames are computed parallel over TIME, still bit for bit the
 * reference's recurrence: chunks start from a warm-up, every chunk boundary is verified exactly and a mismatch recomputed
 * (pcm_kernels.hip, biquad_spec_kernel).  These report how many chunks had to be recomputed so far (normally 0). */
unsigned int biquadBankRepairs (BiquadBank *bank);   /* synchronises */
unsigned int artamdBiquadRepairs (void);             /* the host-pointer calls (biquad_apply_buffer) of this process */

/* The reference's void entry points (biquad_apply_buffer / _sample, floatIntegersLE) cannot return an error and this library has no
 * CPU path: a failure there (no device, allocation, launch) is printed to stderr, counted, and leaves silence (floatIntegersLE) or the
 * unfiltered samples (biquad) behind.  A tool checks the count before it trusts what it writes; ARTAMD_ABORT_ON_ERROR=1 aborts instead. */
int artamdErrorCount (void);
const char *artamdLastError (void);                  /* NULL while the count is zero */

/* How many periods of an exact rational ratio (outputsPerPeriod = the numerator of the reduced ratio) the matrix-core kernels take at
 * a time: their tiles hold 32 consecutive outputs of one period, so short periods (2x conversions: 2 outputs) or badly fitting ones
 * are taken several at a time — 1 while the padding stays within 15 %.  Informational (the choice is the library's own: the 4-byte
 * build applies the rule with the 64 rows of its fixed-point slab kernel's tiles, which are whole 32-row tiles too — ...Rows (p, 64) —
 * unless ARTAMD_I8_SLAB=0). */
int artamdPeriodMultiple (int outputsPerPeriod);
int artamdPeriodMultipleRows (int outputsPerPeriod, int rows);

/* ---- decimator, device pointers ---- */
void decimateHipSetStream (Decimate *cxt, void *hipStream);
/* asynchronous; clipped-sample count accumulates on the device, read with decimateHipClipped() */
void decimateProcessInterleavedLEDevice (Decimate *cxt, const artsample_t *d_input, int numInputFrames, unsigned char *d_output);
long decimateHipClipped (Decimate *cxt);             /* synchronises; total clipped since init */
/* shards of a DECIMATE_MULTITHREADED context (decimator.h: the reference's one-worker-per-channel fan-out, decimator.c:92-93,
 * 119-136, with devices for threads — ordinary contexts with contiguous channel slices on the devices of artamdSetDevices ());
 * 0: an ordinary context */
int decimateHipShardCount (Decimate *cxt);
void floatIntegersLEDevice (const unsigned char *d_input, double inputGain, int inputBits, int inputBytes, int inputStride,
                            artsample_t *d_output, int numSamples, void *hipStream);

/* ---- time stretcher, device pointers (stretch.h) ---- */
void stretchHipSetStream (Stretch *cxt, void *hipStream);
/* as stretchProcess / stretchFlush with `samples` / `output` in device memory; both wait for the call to finish (the frame
 * count is the return value).  d_output must hold stretchGetOutputCapacity() frames. */
int stretchProcessDevice (Stretch *cxt, const artsample_t *d_samples, int num_samples, artsample_t *d_output, double ratio);
int stretchFlushDevice (Stretch *cxt, artsample_t *d_output);
/* The stretcher is one serial state machine per stream, so the GPU earns its keep on MANY streams: these make the call
 * above on n independent contexts in ONE launch, one workgroup per context (results identical to n separate calls; the
 * launch uses the stream of cxts[0]; a context may appear only once).  produced[i] = frames written to d_outputs[i].
 * Return 0, or -1 if the launch failed (nothing is then known about the contexts' state). */
int stretchProcessBatchDevice (Stretch *const *cxts, int n, const artsample_t *const *d_samples, const int *num_samples,
                               artsample_t *const *d_outputs, const double *ratios, int *produced);
int stretchFlushBatchDevice (Stretch *const *cxts, int n, artsample_t *const *d_outputs, int *produced);

#ifdef __cplusplus
}
#endif
#endif
