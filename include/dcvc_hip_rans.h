/* dcvc_hip_rans.h -- OPT-IN device-side entropy coding for the DCVC-HEM path.
 *
 * NOT the reference's wire format.  The reference's bitstream is one sequential rANS64 stream
 * (DCVC_HEM/src/cpp/rans/rans_interface.cpp:85-244), which only a single host thread can
 * produce or consume; include/dcvc_rans.h implements that format bit-exactly and stays the
 * default.  This header adds a lane-interleaved variant of the SAME coder (same 16-bit CDF
 * tables, same sentinel + 4-bit bypass escape, same 64-bit state / 32-bit renormalisation) that
 * runs on the GPU so that symbol planes never travel to the host:
 *
 *   section := u32 n_symbols, u32 lanes, u32 words[lanes], lane streams (u32 words) back to back
 *   lane j codes symbols j, j + lanes, j + 2*lanes, ... ; its stream is exactly what
 *   dcvc_rans_encoder_flush would emit for those symbols alone (final state first).
 *   A picture payload is the 4-byte magic "DGR1" followed by its sections in the reference's
 *   plane order (mv_z, mv_y step 0, mv_y step 1, z, y step 0, y step 1; video_model.py:333-340).
 *
 * Entry points work on one section at a device-resident cursor (in u32 words) into a payload
 * buffer: the encoder appends a section behind *cursor_in, the decoder consumes the one there; both
 * write the new position to *cursor_out (callers ping-pong two locations).  All
 * pointers are device pointers; launches are stream-ordered; *status (device int32) is OR-ed
 * with DCVC_DRANS_BAD_* bits instead of faulting on bad input.
 */
#ifndef DCVC_HIP_RANS_H
#define DCVC_HIP_RANS_H

#include <stdint.h>

#include "dcvc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DCVC_DRANS_MAGIC 0x31524744u /* "DGR1" little endian */
#define DCVC_DRANS_MAX_LANES 8192
#define DCVC_DRANS_BAD_INDEX 1   /* CDF row out of range */
#define DCVC_DRANS_BAD_SPACE 2   /* payload or lane scratch too small */
#define DCVC_DRANS_BAD_STREAM 4  /* decoder ran out of words / inconsistent section header */

/* The lane count both sides use for a plane of n symbols unless they agree otherwise: one lane per
 * 512 symbols, a multiple of 64, in [64, 1024].  Per lane the format spends 12 bytes and a lane is
 * strictly serial (~1 us per symbol on gfx950: ~250 dependent instructions of 64-bit integer work),
 * so `lanes` trades payload size against latency: 512 symbols per lane costs 0.19 bit per symbol and
 * ~0.4 ms per plane, 64 per lane 1.5 bit per symbol and ~60 us. */
int32_t dcvc_drans_default_lanes(int64_t n);
/* words of lane scratch dcvc_drans_encode needs for n symbols on `lanes` lanes */
int64_t dcvc_drans_scratch_words(int64_t n, int32_t lanes);

/* HOST helper: lut (n_cdfs x 256 bytes) for dcvc_drans_decode: lut[row][b] = the bin that holds
 * cumulative count b << 8 (the decoder's search starts there).  Upload it once per table. */
int dcvc_drans_build_lut(const int32_t *cdfs, int32_t n_cdfs, int32_t cdf_stride, const int32_t *cdf_sizes,
                         uint8_t *lut);

/* sym: n symbols in coding order.  idx: n CDF rows, or NULL for "row = channel" planes laid out
 * (N, C, H, W): row = (i / chan_hw) % chan_c.  Tables as in dcvc_rans_encoder_encode_with_indexes
 * (at most 32768 entries; they are staged in LDS).  Appends one section at payload[*cursor_in]
 * (payload_words = capacity) and writes the position behind it to *cursor_out (a different
 * location: the lanes are spread over several workgroups, all of which read *cursor_in). */
int dcvc_drans_encode(const int32_t *sym, const int32_t *idx, int32_t chan_hw, int32_t chan_c, int64_t n,
                      const int32_t *cdfs, int32_t n_cdfs, int32_t cdf_stride, const int32_t *cdf_sizes,
                      const int32_t *offsets, int32_t lanes, uint32_t *scratch, int64_t scratch_words,
                      uint32_t *payload, int64_t payload_words, const int32_t *cursor_in, int32_t *cursor_out,
                      int32_t *status, void *stream);

/* Decodes the section at payload[*cursor_in] into out (n symbols; n, lanes and the tables must match
 * what the encoder was given: a mismatch with the section header sets DCVC_DRANS_BAD_STREAM) and
 * writes the position behind it to *cursor_out.  n_cdfs <= 256, cdf_stride <= 256. */
int dcvc_drans_decode(const uint32_t *payload, int64_t payload_words, const int32_t *cursor_in, int32_t *cursor_out,
                      const int32_t *idx, int32_t chan_hw, int32_t chan_c, int64_t n, const int32_t *cdfs,
                      int32_t n_cdfs, int32_t cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets,
                      const uint8_t *lut, int32_t lanes, int32_t *out, int32_t *status, void *stream);

#ifdef __cplusplus
}
#endif
#endif
