/* art_internal.h — private C ABI between the C host layer (resampler_host.c, pcm_host.c) and the
 * HIP translation units (device_rt.hip, sinc_fir.hip, pcm_kernels.hip).  Plain C types only. */
#ifndef ART_INTERNAL_H
#define ART_INTERNAL_H

#include <stddef.h>
#include <stdint.h>
#include "art_hip.h"

#if defined(PATH_WIDTH) && (PATH_WIDTH==64)
#define ART_WIDE 1          /* double-precision sample path: general + strict kernels only */
#else
#define ART_WIDE 0
#endif
typedef artsample_t art_s;   /* the sample type of this build */

#ifdef __cplusplus
extern "C" {
#endif

/* header of a context's digit-plane buffer (fixed-point matrix kernel, fir_matrix_i8.hip), zeroed when the buffer is allocated:
 * [0] stand-down flag word.  The rows' mask words follow the header. */
#define ART_I8_FLAG_BYTES 256    /* the flag word of the fixed-point kernel's buffer (and what shares its cache line) */
#define ART_I8_HEAD_BYTES 32768  /* the buffer's header, zero when allocated: the flag, then the slab kernel's arrival counters (8 XCDs x 64 tiles x 8 waves) */

#define ART_SPLIT_HEAD_BYTES 65536   /* arrival counters of the K-split kernel: 4 per tile, up to 4096 tiles */
#define ART_MAX_SEGS 192         /* ring-epoch segments per kernel launch (passed by value: 16 B each, kernel arguments stay below 4 KB) */

/* numeric modes of the FIR */
enum { ART_MODE_FAST = 0,        /* f32 FMA accumulation, any order (default) */
       ART_MODE_PRECISE = 1,     /* f64 accumulation (EXTEND_CONVOLUTION_MATH) */
       ART_MODE_STRICT = 2 };    /* reference C source order, un-fused (RESAMPLE_STRICT_ORDER) */

enum { ART_KERNEL_AUTO = 0, ART_KERNEL_GENERAL = 1, ART_KERNEL_MFMA = 2,
       /* the cut-invariant stream policy (art_hip.h, resampleHipSetCutInvariant): ONE arithmetic for every output of a rational-ratio stream however its
        * input is cut into calls — every launch, shorter than a period or not, on the f32 streaming kernel, un-split, anchored on the stream's canonical
        * period; a launch that cannot run anchored is the general kernel's (whose outputs never depend on the cut either) and is counted */
       ART_KERNEL_INVARIANT = 9 };
/* (preferences 6 and 9 both pin the f32 streaming kernel: no fixed point, no K split) */
#define ART_PREF_PINS_F32(k) ((k) == 6 || (k) == ART_KERNEL_INVARIANT)
#define ART_FIR_ROLLED 0x100             /* arthip_fir: the history roll rode along with this launch */

typedef struct {
    int count;
    int lin_floor;                       /* linear indices below this read as silence */
    unsigned int first [ART_MAX_SEGS];   /* first output (call-relative) of each segment, ascending */
    int lin_base [ART_MAX_SEGS];         /* ring index -> linear index (history ++ input) */
    double base [ART_MAX_SEGS];          /* outputOffset of the ring epoch */
} ArtSegTable;

typedef struct {
    const art_s *bank;                   /* device, (F+1) x T */
    const art_s *hist;                   /* device, H frames x C, interleaved */
    const art_s *in;                     /* device, new input frames */
    long in_pitch;                       /* 0: interleaved [frame][C]; else planar, channel c at in + c*in_pitch */
    art_s *out;
    long out_pitch;
    art_s *roll_dst;                     /* when set: the FIR launch also rolls the history (extra workgroups) */
    int roll_appended;                   /*   (frames appended by this call) into roll_dst; arthip_fir then returns k | ART_FIR_ROLLED */                      /* 0: interleaved; else planar */
    int in_frames;                       /* frames valid at `in` (reads beyond return 0) */
    int C, T, F, H;
    int stream_C;                        /* channels of the whole stream when this context is a shard of a multi-device context (0: = C): the
                                          * kernel choice is made for the stream, so that a shard and an ordinary context of the same stream
                                          * run the same kernels and produce the same bits */
    int interpolate, lowpass;            /* SUBSAMPLE_INTERPOLATE / INCLUDE_LOWPASS in effect */
    int mode;                            /* ART_MODE_* */
    double ratio;
    unsigned int n_begin, n_end;         /* call-relative output frames to produce */
    long long lin_origin;                /* stream frame (input frames appended since init / reset, less H) that linear frame 0 of this call is: two calls' linear
                                          * indices differ by the difference of theirs — the rows kept across calls place a launch in the stream's tiling with it */
    int n_skip;                          /* matrix kernels on cached rows (below): the launch's tiles start at the cached period's first slot, n_skip
                                          * slots before the launch's first output — those slots of the first period are computed and not stored */
    int segs_truncated;                  /* the launch reaches beyond the segments of its table (a call of more than ART_MAX_SEGS ring epochs
                                          * handed over whole, arthip_fir_spans_segments): only a kernel that follows the lattice from the
                                          * launch's first period may run it — arthip_fir returns -2, nothing enqueued, otherwise */
    /* periodic-phase structure for the MFMA kernel (0 = none): out frame n+period_out sits exactly
     * period_in input frames after out frame n */
    int period_out, period_in;
    /* counters of outputs the matrix kernels evaluated off their slot's canonical pattern (device memory; fix_list is only
     * tested for non-NULL: the path needs the counters) */
    unsigned int *fix_list, *fix_count;
    unsigned int fix_cap;
    /* device scratch for the MFMA path: per-launch effective rows + canonical slot positions */
    void *scratch; size_t scratch_bytes;
    /* device memory for the fixed-point matrix kernel's digit planes of one launch (arthip_fir_planes_bytes; NULL: f32 kernels) */
    void *planes; size_t planes_bytes;
    /* the fixed-point kernel's filter rows ACROSS calls (digit planes, masks, the f32 tables of its stand-by): device memory of
     * arthip_fir_rows_bytes () bytes and a zeroed host block of arthip_fir_rows_cache_bytes () bytes that describes what it holds, both owned
     * by the context (NULL: the rows are rebuilt by every launch, as before round 5).  rows_masks_out (host, optional): where the used
     * set's row masks live on the device (resampleHipLastFixedPoint reads them) */
    void *rows; size_t rows_bytes; void *rows_cache; void **rows_masks_out;
    /* device memory for the K-split streaming kernel of launches with few tiles (arthip_fir_split_bytes; NULL: unsplit): the first
     * ART_SPLIT_HEAD_BYTES are arrival counters, zero whenever no launch is in flight (zeroed by the owner when allocated) */
    void *split; size_t split_bytes;
    /* device memory for launches of a channel count the matrix kernels are not compiled for (anything but 1, 2, 4, 8, 16, 32): the
     * launch runs in groups of up to 32 channels, each copied into a buffer of the next compiled width (arthip_fir_pad_bytes; NULL: the
     * generic matrix kernel runs such a stream, several times slower) */
    void *pad; size_t pad_bytes;
    /* host, optional, 4 ints filled when the fixed-point kernel is enqueued: the launch's flag value (the first word of
     * `planes` equals it afterwards iff the kernel stood down), mask words behind the header (at planes + ART_I8_HEAD_BYTES),
     * chunks per tile, the kernel's form (1 register-staged, 2 LDS-DMA, 3 slabs) */
    int *fixed_out;
    /* optional HIP events recorded immediately before/after the dominant kernel's launch (host side only) */
    void *ev_start, *ev_stop;
} ArtFirArgs;

/* ---- device_rt.hip ---- */
int   arthip_device_count (void);
int   arthip_current_device (void);
int   arthip_set_device (int device);
void *arthip_stream_create (void);                         /* non-blocking stream on the current device */
void  arthip_stream_destroy (void *stream);
int   arthip_copy2d (void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows, void *stream);   /* any direction */
int   arthip_copy (void *dst, const void *src, size_t bytes, void *stream);                                                /* any direction */
int   arthip_copy_by_kernel (void *dst, const void *src, size_t bytes, void *stream);   /* page-locked host <-> device, small sizes */
int   arthip_copy2_by_kernel (void *dst, const void *src, size_t bytes, void *dst2, const void *src2, size_t bytes2, void *stream);
int   arthip_slice_copy (void *dst, size_t dpitch_words, const void *src, size_t spitch_words, int width_words, size_t rows, void *stream);   /* strided rows of 4-byte words, by a kernel (device / peer memory) */
void *arthip_host_alloc (size_t bytes);                    /* page-locked host memory */
void  arthip_host_free (void *p);
void *arthip_order_event_create (void);                    /* event without timing, for cross-stream ordering */
int   arthip_stream_wait_event (void *stream, void *event);
int   arthip_event_sync (void *event);
int   arthip_enable_peer (int device, int peer);            /* 1: `device` can address `peer`'s memory (or is it); 0: no route */
int   arthip_slice_copy_bytes (void *dst, size_t dpitch, const void *src, size_t spitch, int width, size_t rows, void *stream);   /* strided rows of bytes, by a kernel */

/* ---- resampler_host.c: the device list of multi-device contexts (artamdSetDevices / ARTAMD_DEVICES / ARTAMD_SHARDS) ----
 * how many shards a MULTITHREADED context of `channels` channels gets (0 or 1: an ordinary context) and on which device shard s
 * lives; devices that cannot address `home`'s memory are replaced by `home` */
#define ART_MAX_DEVICES 64
int   artamd_shard_plan (int channels, int home, int *devices_out);
void *arthip_malloc (size_t bytes);
void  arthip_free (void *p);
int   arthip_h2d (void *dst, const void *src, size_t bytes, void *stream);
int   arthip_d2h (void *dst, const void *src, size_t bytes, void *stream);
int   arthip_d2d (void *dst, const void *src, size_t bytes, void *stream);
int   arthip_zero (void *dst, size_t bytes, void *stream);
int   arthip_sync (void *stream);
const char *arthip_last_error (void);
void arthip_set_last_error (const char *text);
void *arthip_event_create (void);
void  arthip_event_destroy (void *ev);
int   arthip_event_record (void *ev, void *stream);
float arthip_event_elapsed_ms (void *start, void *stop);   /* synchronises on `stop` */

/* ---- sinc_fir.hip ---- */
/* returns the kernel actually used (ART_KERNEL_*), <0 on launch failure */
int arthip_fir (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref, void *stream);
/* bytes a->planes must hold for the fixed-point matrix kernel to run a call of this shape (C, T, H, in_frames, period) making
 * `outputs` frames; 0: the call is not for it (shape, size, kernel preference) */
size_t arthip_fir_planes_bytes (const ArtFirArgs *a, unsigned int outputs, int kernel_pref);
/* the fixed-point kernel's rows across the calls of a context (ArtFirArgs.rows / rows_cache): device bytes a call of this shape wants (0: none),
 * size of the host block that describes the device buffer (zeroed by the owner), forgetting what the buffer held (it was replaced), releasing
 * what the host block owns */
size_t arthip_fir_rows_bytes (const ArtFirArgs *a, unsigned int outputs, int kernel_pref);
size_t arthip_fir_rows_cache_bytes (void);
void   arthip_fir_rows_cache_reset (void *cache);
void   arthip_fir_rows_cache_free (void *cache);
/* bytes a->split must hold for a call of this shape making `outputs` frames to run on the K-split kernel; 0: the call is not for it */
size_t arthip_fir_split_bytes (const ArtFirArgs *a, unsigned int outputs, int kernel_pref);
/* n independent general-kernel calls (default / precise mode) in one launch per kernel variant; d_table = device scratch of
 * n * arthip_fir_batch_item_bytes () bytes (reused call after call: stream order protects it); asynchronous like arthip_fir */
size_t arthip_fir_batch_item_bytes (void);
int arthip_fir_batch_max_segments (void);                /* ring-epoch segments a batched call may have */
int arthip_fir_takes_matrix_path (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref);   /* what arthip_fir would do */
size_t arthip_fir_pad_bytes (const ArtFirArgs *a, unsigned int outputs);      /* bytes a->pad wants for a call of this shape making `outputs` frames (0: none) */
/* May a call of more segments than a table holds be ONE launch (n_begin .. n_end = the whole call, segs = its first ART_MAX_SEGS
 * segments)?  Yes where the launch runs on a streaming matrix-core kernel: those take their positions from the lattice of the
 * launch's first period, not from the table (short filters: a ring epoch is a few hundred frames, a 1M-frame call eight tables) */
int arthip_fir_spans_segments (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref);
int arthip_fir_batch (const ArtFirArgs *a, const ArtSegTable *segs, int n, void *d_table, void *stream);
/* new_hist[H][C] = last H frames of (hist ++ in[0..appended)); in may be NULL => zeros appended */
int arthip_roll_history (art_s *new_hist, const art_s *hist, const art_s *in, long in_pitch, int appended, int H, int C, void *stream);
int arthip_interleave (art_s *dst, const art_s *src_planar, long pitch, int frames, int C, void *stream);
int arthip_deinterleave (art_s *dst_planar, long pitch, const art_s *src, int frames, int C, void *stream);

/* ---- extrapolate_host.c (host, scalar) ---- */
void art_extrapolate_forward (art_s *x, int count, int extra);
void art_extrapolate_backward (const art_s *known_newest_last, int count, art_s *older_nearest_first, int extra);

/* ---- pcm_kernels.hip ---- */
typedef struct {
    int C, bits, bytes, dither_type, dither_on, shaping_on;
    int shaping_order;                   /* order of the error-feedback filter (same for every channel) */
    art_s scale;
    art_s *feedback;                     /* device [C] */
    uint32_t *gens;                      /* device [C] */
    uint32_t *gens_next;                 /* device [C]: where the fully parallel kernel leaves the generator state */
    Biquad *shapers;                     /* device [C] */
    unsigned long long *clipped;         /* device counter */
} ArtDecArgs;
/* returns 1 when the generator state was written to a->gens_next (caller swaps), 0 otherwise, <0 on error */
int arthip_decimate (const ArtDecArgs *a, const art_s *d_in, int frames, unsigned char *d_out, void *stream);
int arthip_decimate_planar (const ArtDecArgs *a, const art_s *d_in, long in_pitch, int frames, unsigned char *d_out, long out_pitch, void *stream);
int arthip_biquad_chain (Biquad *d_sections, int C, int S, art_s *d_buf, int frames, int stride, void *stream);
/* every section has order 2, S = 1 or 2, interleaved frames: hand-scheduled kernel */
int arthip_biquad_order2 (Biquad *d_sections, int C, int S, art_s *d_buf, int frames, int stride, void *stream);   /* stride >= C: values between frames */
/* bit-exact cascade, parallel over time (speculative chunks + exact verification, pcm_kernels.hip): d_in -> d_out, distinct
 * buffers; L = chunk length, W = warm-up frames per section; d_states: arthip_biquad_spec_scratch () bytes of scratch */
size_t arthip_biquad_spec_scratch (int C, int S, int frames, int L);
int arthip_biquad_spec_arm (int *d_first_bad, int C, void *stream);      /* once per scratch: C ints the calls keep at "no mismatch" */
int arthip_biquad_spec (Biquad *d_sections, int C, int S, const art_s *d_in, int in_stride, art_s *d_out, int out_stride, int frames,
                        int L, int W, void *d_states, int *d_first_bad, unsigned int *d_repairs, void *stream);
/* ---- time stretcher (stretch_kernels.hip) ---- */
typedef struct {
    art_s *ring [2][2];                  /* [stage][ping-pong] input rings, `room` values each */
    art_s *between;                      /* hand-over buffer stage 1 -> stage 2 (cascaded pair) */
    art_s *total, *score;                /* search scratch: longest + 4 values each */
    void *state;                         /* device: { int mark, fill, cur, pad; double drift; } per stage */
    int channels, room, lo, hi, quick, paired;
} ArtStretchArgs;
/* one stretchProcess (flush == 0) or stretchFlush (flush != 0) call; *d_result receives the frames written */
int arthip_stretch_call (const ArtStretchArgs *h, const art_s *d_in, int frames, art_s *d_out, double ratio, int flush,
                         int *d_result, void *stream);

/* the same call on n independent streams in ONE launch (one workgroup per stream): d_items / d_done in device memory */
typedef struct { int mark, fill, cur, pad; double drift; } ArtStretchState;          /* per stage; layout of stretch_kernels.hip */
typedef struct { ArtStretchArgs args; const art_s *in; art_s *out; double ratio; int frames, flush; } ArtStretchItem;
typedef struct { int made, pad; ArtStretchState state [2]; } ArtStretchDone;
int arthip_stretch_batch (const ArtStretchItem *d_items, ArtStretchDone *d_done, int n, void *stream);

/* a failure of an entry point that cannot return one (the reference's ABI has no error codes): printed, counted (artamdErrorCount /
 * artamdLastError), fatal under ARTAMD_ABORT_ON_ERROR=1 — pcm_host.c */
void artamd_note_failure (const char *what);

int arthip_ingest (const unsigned char *d_in, art_s gain_factor, int bits, int bytes, int stride, art_s *d_out, int n, void *stream);

#ifdef __cplusplus
}
#endif
#endif
