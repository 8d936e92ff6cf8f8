/* art_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A from-scratch restatement (plain C99, scalar) of the reference's sinc resampler,
 * biquad and decimator arithmetic, written from the behavioural description in
 * SURVEY.md Appendix A and pinned against the real reference (oracle/_ref, built by
 * oracle/Makefile from /root/reference) and the golden fixtures in tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product library (audio_resampler_amd/csrc) never links, loads or calls it.
 *
 * Every function cites the reference file:line whose behaviour it follows.
 */
#ifndef ART_ORACLE_H
#define ART_ORACLE_H

#include <stdint.h>

/* sample type: float, or double when built with -DORA_WIDE (the reference's PATH_WIDTH=64 builds, resampler.h:22-26;
 * _build/liboracle64_*.so) */
#ifdef ORA_WIDE
typedef double ora_s;
#else
typedef float ora_s;
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* flag values are ABI of the reference (resampler.h:28-38, decimator.h:29-40) */
#define ORA_INTERPOLATE     0x1
#define ORA_BLACKMAN_HARRIS 0x2
#define ORA_LOWPASS         0x4
#define ORA_MULTITHREADED   0x8
#define ORA_NO_REDUCTION    0x10
#define ORA_FIXED_RATIO     0x20
#define ORA_EXTRAPOLATE     0x40
#define ORA_PREFILL         0x80
#define ORA_PRECISE         0x100
#define ORA_FLUSHED         0x200
#define ORA_SNAP            0x400

#define ORA_DITHER_HIGHPASS 0x1
#define ORA_DITHER_FLAT     0x2
#define ORA_DITHER_LOWPASS  0x4
#define ORA_DITHER_ANY      0x7
#define ORA_SHAPE_1ST       0x100
#define ORA_SHAPE_2ND       0x200
#define ORA_SHAPE_3RD       0x400
#define ORA_SHAPE_ATH       0x800
#define ORA_SHAPE_ANY       0xF00

typedef struct { unsigned int used, generated; } OraResult;

typedef struct OraResampler {
    int channels, taps, filters, ring_len, write_pos, flags;
    double read_pos, fixed_ratio, lowpass_ratio;
    ora_s *bank;            /* (filters+1) rows x taps, contiguous */
    ora_s *ring;            /* channel c at ring + c*(ring_len+taps); each preceded by `taps` guard zeros */
    ora_s *ring_store;
} OraResampler;

OraResampler *ora_resample_init (int channels, int taps, int filters, double lowpass_ratio, int flags);
OraResampler *ora_resample_fixed_init (int channels, int taps, int max_filters, double src_rate, double dst_rate, int lowpass_freq, int flags);
void ora_resample_free (OraResampler *r);
void ora_resample_reset (OraResampler *r);
void ora_resample_advance (OraResampler *r, double delta);
double ora_resample_position (const OraResampler *r);
unsigned ora_resample_required_input (const OraResampler *r, int n_out, double ratio);
unsigned ora_resample_expected_output (const OraResampler *r, int n_in, double ratio);

/* n_in < 0 => flush (input may be NULL).  threads > 1 => one pthread per channel
 * (the reference's workers.c model: last channel on the caller, join per call). */
OraResult ora_resample_interleaved (OraResampler *r, const ora_s *in, int n_in, ora_s *out, int out_cap, double ratio, int threads);
OraResult ora_resample_planar (OraResampler *r, const ora_s *const *in, int n_in, ora_s *const *out, int out_cap, double ratio, int threads);
OraResult ora_resample_interleaved_flush (OraResampler *r, const ora_s *in, int n_in, ora_s *out, int out_cap, double ratio, int threads);

/* plain dot products, exposed for unit tests */
double ora_dot_outside_in (const ora_s *h, const ora_s *x, int taps);
double ora_dot_precise (const ora_s *h, const ora_s *x, int taps);

/* ---- biquad (biquad.c) ---- */
typedef struct { ora_s a0, a1, a2, a3, a4, b1, b2, b3, b4; } OraBiquadCoeffs;      /* 36 bytes (72 wide) */
typedef struct { ora_s a[5], b[5], x[4], y[4]; int order, index; } OraBiquad;      /* 80 bytes (152 wide) */
void ora_biquad_lowpass (OraBiquadCoeffs *c, double freq);
void ora_biquad_highpass (OraBiquadCoeffs *c, double freq);
void ora_biquad_init (OraBiquad *f, const OraBiquadCoeffs *c, double gain);
ora_s ora_biquad_sample (OraBiquad *f, ora_s in);
void ora_biquad_buffer (OraBiquad *f, ora_s *buf, int n, int stride);

/* ---- decimator (decimator.c) ---- */
typedef struct OraDecimator {
    int channels, bits, bytes, dither_type, flags;
    double gain;
    ora_s *feedback;
    uint32_t *gens;
    OraBiquad *shapers;
} OraDecimator;
OraDecimator *ora_decimate_init (int channels, int bits, int bytes, double gain, int rate, int flags);
void ora_decimate_free (OraDecimator *d);
int ora_decimate_interleaved (OraDecimator *d, const ora_s *in, int frames, unsigned char *out);
int ora_decimate_planar (OraDecimator *d, const ora_s *const *in, int frames, unsigned char *const *out);
void ora_float_integers_le (const unsigned char *in, double gain, int bits, int bytes, int stride, ora_s *out, int n);

/* ---- artest's synthetic workload (artest.c:744-798) ---- */
uint64_t ora_noise_fill (ora_s *dst, long count, uint64_t state);   /* returns next state; seed 0x3141592653589793 */
void ora_fade_in (ora_s *data, int count);
void ora_fade_out (ora_s *data, int count);
uint64_t ora_checksum_words (uint64_t c, const void *words, long nwords);   /* c = c*3 + u32, artest.c:97 */
uint64_t ora_checksum_bytes (uint64_t c, const unsigned char *bytes, long nbytes); /* artest.c:587-588 */

#ifdef __cplusplus
}
#endif
#endif
