/* dcvc_rans.h -- C ABI of the host-side entropy coder of the DCVC-HEM bitstream.
 *
 * These entry points are exactly what the reference's two pybind11 modules expose to
 * DCVC_HEM/src/entropy_models/entropy_models.py (imported lazily at :13 and :19):
 *
 *   MLCodec_rans.BufferedRansEncoder   /root/reference/DCVC_HEM/src/cpp/rans/rans_interface.cpp:85-174, :246-255
 *       .encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets) -> dcvc_rans_encoder_encode_with_indexes
 *       .flush() -> bytes                                                  -> dcvc_rans_encoder_flush
 *       .reset()                                                           -> dcvc_rans_encoder_reset
 *   MLCodec_rans.RansDecoder           rans_interface.cpp:176-244, :257-260
 *       .set_stream(bytes)                                                 -> dcvc_rans_decoder_set_stream
 *       .decode_stream(indexes, cdfs, cdfs_sizes, offsets) -> int32[n]     -> dcvc_rans_decoder_decode_stream
 *   MLCodec_CXX.pmf_to_quantized_cdf(pmf, precision)  src/cpp/ops/ops.cpp:24-91 -> dcvc_pmf_to_quantized_cdf
 *
 * Wire format (must stay bit-exact): 64-bit rANS state, 16-bit probabilities, 32-bit
 * renormalisation words written back to front, final state as two words at the head of
 * the stream; out-of-table symbols use the sentinel (last) CDF slot followed by 4-bit
 * bypass nibbles (count in unary base 15, then the nibbles LSB first).
 *
 * Unlike the reference (assert-only, undefined behaviour on bad input) every function
 * returns a status: 0 / a non-negative size on success, a negative DCVC_RANS_E_* otherwise.
 * Handles are not thread-safe; distinct handles may be used from distinct threads.
 */
#ifndef DCVC_RANS_H
#define DCVC_RANS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCVC_RANS_E_ARG (-1)      /* null pointer / negative size */
#define DCVC_RANS_E_INDEX (-2)    /* an index outside [0, n_cdfs) or a CDF size outside [2, stride] */
#define DCVC_RANS_E_SPACE (-3)    /* output buffer too small (flush) */
#define DCVC_RANS_E_STREAM (-4)   /* decoder ran past the end of the stream / no stream set */

typedef struct dcvc_rans_encoder dcvc_rans_encoder;
typedef struct dcvc_rans_decoder dcvc_rans_decoder;

dcvc_rans_encoder *dcvc_rans_encoder_create(void);
void dcvc_rans_encoder_destroy(dcvc_rans_encoder *e);
int dcvc_rans_encoder_reset(dcvc_rans_encoder *e);
/* Appends n symbols.  cdfs is (n_cdfs, cdf_stride) int32 row-major; cdf_sizes[i] counts the
 * valid entries of row i (symbols + sentinel + 1); offsets[i] is subtracted from a symbol. */
int dcvc_rans_encoder_encode_with_indexes(dcvc_rans_encoder *e, const int32_t *symbols, const int32_t *indexes,
                                          int64_t n, const int32_t *cdfs, int32_t n_cdfs, int32_t cdf_stride,
                                          const int32_t *cdf_sizes, const int32_t *offsets);
/* Upper bound in bytes of what flush will produce for the symbols buffered so far. */
int64_t dcvc_rans_encoder_flush_bound(const dcvc_rans_encoder *e);
/* Encodes everything buffered (in reverse) into out; returns the byte count.  Like the
 * reference's flush() it leaves the buffer empty. */
int64_t dcvc_rans_encoder_flush(dcvc_rans_encoder *e, uint8_t *out, int64_t cap);

dcvc_rans_decoder *dcvc_rans_decoder_create(void);
void dcvc_rans_decoder_destroy(dcvc_rans_decoder *d);
int dcvc_rans_decoder_set_stream(dcvc_rans_decoder *d, const uint8_t *bytes, int64_t n);
/* Decodes n symbols, advancing the cursor shared by successive calls on one stream. */
int dcvc_rans_decoder_decode_stream(dcvc_rans_decoder *d, const int32_t *indexes, int64_t n, const int32_t *cdfs,
                                    int32_t n_cdfs, int32_t cdf_stride, const int32_t *cdf_sizes,
                                    const int32_t *offsets, int32_t *out);

/* cdf must have room for n + 1 entries. */
int dcvc_pmf_to_quantized_cdf(const float *pmf, int32_t n, int32_t precision, uint32_t *cdf);

#ifdef __cplusplus
}
#endif
#endif
