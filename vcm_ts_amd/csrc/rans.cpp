// rans.cpp -- host entropy coder behind include/dcvc_rans.h.
//
// Restates the wire format of /root/reference/DCVC_HEM/src/cpp/rans/rans_interface.cpp
// (:85-244) on the published 64-bit rANS of ryg_rans' rans64.h (absent from the reference
// tree): state in [2^31, 2^63), 16-bit frequencies, 32-bit renormalisation.
// Design differs from the reference where the format allows: symbols are buffered as one
// packed 32-bit record (bypass nibbles flagged by a zero frequency), the decoder finds the
// symbol by binary search instead of a linear scan, and every index is validated.
#include <cstdint>
#include <cstring>
#include <cmath>
#include <new>
#include <vector>

#include "dcvc_rans.h"

namespace {
constexpr uint64_t kLower = 1ull << 31;
constexpr uint32_t kProbBits = 16;
constexpr uint32_t kNibbleBits = 4;
constexpr int32_t kNibbleMax = (1 << kNibbleBits) - 1;

inline uint32_t pack(uint32_t start, uint32_t freq) { return start | (freq << 16); }  // freq 0 => bypass nibble
}  // namespace

struct dcvc_rans_encoder {
    std::vector<uint32_t> rec;
};

struct dcvc_rans_decoder {
    std::vector<uint32_t> words;
    size_t pos = 0;
    uint64_t state = 0;
    bool ready = false;
};

extern "C" dcvc_rans_encoder *dcvc_rans_encoder_create(void) { return new (std::nothrow) dcvc_rans_encoder(); }
extern "C" void dcvc_rans_encoder_destroy(dcvc_rans_encoder *e) { delete e; }
extern "C" int dcvc_rans_encoder_reset(dcvc_rans_encoder *e) {
    if (!e) return DCVC_RANS_E_ARG;
    e->rec.clear();
    return 0;
}

extern "C" int dcvc_rans_encoder_encode_with_indexes(dcvc_rans_encoder *e, const int32_t *symbols,
                                                     const int32_t *indexes, int64_t n, const int32_t *cdfs,
                                                     int32_t n_cdfs, int32_t cdf_stride, const int32_t *cdf_sizes,
                                                     const int32_t *offsets) {
    if (!e || n < 0 || (n > 0 && (!symbols || !indexes)) || !cdfs || !cdf_sizes || !offsets || n_cdfs <= 0 ||
        cdf_stride < 2)
        return DCVC_RANS_E_ARG;
    for (int32_t i = 0; i < n_cdfs; ++i)
        if (cdf_sizes[i] < 2 || cdf_sizes[i] > cdf_stride) return DCVC_RANS_E_INDEX;
    const size_t mark = e->rec.size();
    e->rec.reserve(mark + (size_t)n + 16);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t row = indexes[i];
        if ((uint32_t)row >= (uint32_t)n_cdfs) {
            e->rec.resize(mark);
            return DCVC_RANS_E_INDEX;
        }
        const int32_t *cdf = cdfs + (size_t)row * cdf_stride;
        const int32_t sentinel = cdf_sizes[row] - 2;
        int32_t v = symbols[i] - offsets[row];
        uint32_t raw = 0;
        bool escape = false;
        if (v < 0) {
            raw = (uint32_t)(-2 * (int64_t)v - 1);
            escape = true;
        } else if (v >= sentinel) {
            raw = (uint32_t)(2 * ((int64_t)v - sentinel));
            escape = true;
        }
        if (escape) v = sentinel;
        e->rec.push_back(pack((uint32_t)cdf[v] & 0xFFFFu, (uint32_t)(cdf[v + 1] - cdf[v]) & 0xFFFFu));
        if (escape) {
            int32_t nib = 0;
            while (nib < 8 && (raw >> (nib * kNibbleBits)) != 0) ++nib;
            int32_t cnt = nib;
            while (cnt >= kNibbleMax) {
                e->rec.push_back(pack(kNibbleMax, 0));
                cnt -= kNibbleMax;
            }
            e->rec.push_back(pack((uint32_t)cnt, 0));
            for (int32_t j = 0; j < nib; ++j) e->rec.push_back(pack((raw >> (j * kNibbleBits)) & kNibbleMax, 0));
        }
    }
    return 0;
}

extern "C" int64_t dcvc_rans_encoder_flush_bound(const dcvc_rans_encoder *e) {
    return e ? (int64_t)(e->rec.size() + 2) * 4 : DCVC_RANS_E_ARG;
}

extern "C" int64_t dcvc_rans_encoder_flush(dcvc_rans_encoder *e, uint8_t *out, int64_t cap) {
    if (!e || !out || cap < 0) return DCVC_RANS_E_ARG;
    const size_t words = e->rec.size() + 2;
    std::vector<uint32_t> buf(words);
    uint32_t *p = buf.data() + words;
    uint64_t x = kLower;
    for (size_t i = e->rec.size(); i-- > 0;) {
        const uint32_t r = e->rec[i];
        const uint32_t start = r & 0xFFFFu, freq = r >> 16;
        if (freq) {
            const uint64_t lim = ((kLower >> kProbBits) << 32) * freq;
            if (x >= lim) {
                *--p = (uint32_t)x;
                x >>= 32;
            }
            x = ((x / freq) << kProbBits) + (x % freq) + start;
        } else {  // raw nibble: frequency 2^(16-4) on a 16-bit scale, as the in-tree PutBits does
            const uint64_t lim = ((kLower >> 16) << 32) * (uint64_t)(1u << (16 - kNibbleBits));
            if (x >= lim) {
                *--p = (uint32_t)x;
                x >>= 32;
            }
            x = (x << kNibbleBits) | start;
        }
    }
    *--p = (uint32_t)(x >> 32);
    *--p = (uint32_t)x;
    const int64_t nbytes = (int64_t)(buf.data() + words - p) * 4;
    if (nbytes > cap) return DCVC_RANS_E_SPACE;
    std::memcpy(out, p, (size_t)nbytes);
    e->rec.clear();
    return nbytes;
}

extern "C" dcvc_rans_decoder *dcvc_rans_decoder_create(void) { return new (std::nothrow) dcvc_rans_decoder(); }
extern "C" void dcvc_rans_decoder_destroy(dcvc_rans_decoder *d) { delete d; }

extern "C" int dcvc_rans_decoder_set_stream(dcvc_rans_decoder *d, const uint8_t *bytes, int64_t n) {
    if (!d || !bytes || n < 8) return DCVC_RANS_E_ARG;
    d->words.assign((size_t)(n + 3) / 4, 0u);
    std::memcpy(d->words.data(), bytes, (size_t)n);
    d->state = (uint64_t)d->words[0] | ((uint64_t)d->words[1] << 32);
    d->pos = 2;
    d->ready = true;
    return 0;
}

namespace {
inline bool refill(dcvc_rans_decoder *d, uint64_t &x) {
    if (x < kLower) {
        if (d->pos >= d->words.size()) return false;
        x = (x << 32) | d->words[d->pos++];
    }
    return true;
}
inline bool take_nibble(dcvc_rans_decoder *d, uint64_t &x, int32_t &v) {
    v = (int32_t)(x & ((1u << kNibbleBits) - 1));
    x >>= kNibbleBits;
    return refill(d, x);
}
}  // namespace

extern "C" int dcvc_rans_decoder_decode_stream(dcvc_rans_decoder *d, const int32_t *indexes, int64_t n,
                                               const int32_t *cdfs, int32_t n_cdfs, int32_t cdf_stride,
                                               const int32_t *cdf_sizes, const int32_t *offsets, int32_t *out) {
    if (!d || n < 0 || (n > 0 && (!indexes || !out)) || !cdfs || !cdf_sizes || !offsets || n_cdfs <= 0) return DCVC_RANS_E_ARG;
    if (!d->ready) return DCVC_RANS_E_STREAM;
    for (int32_t i = 0; i < n_cdfs; ++i)
        if (cdf_sizes[i] < 2 || cdf_sizes[i] > cdf_stride) return DCVC_RANS_E_INDEX;
    uint64_t x = d->state;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t row = indexes[i];
        if ((uint32_t)row >= (uint32_t)n_cdfs) return DCVC_RANS_E_INDEX;
        const int32_t *cdf = cdfs + (size_t)row * cdf_stride;
        const int32_t size = cdf_sizes[row], sentinel = size - 2;
        const uint32_t cum = (uint32_t)(x & 0xFFFFu);
        // first entry > cum, minus one (the reference scans linearly; same answer)
        int32_t lo = 0, hi = size;
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if ((uint32_t)cdf[mid] > cum) hi = mid; else lo = mid + 1;
        }
        int32_t s = lo - 1;
        if (s < 0 || s + 1 >= size) return DCVC_RANS_E_STREAM;
        const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
        x = (uint64_t)freq * (x >> kProbBits) + (x & 0xFFFFu) - start;
        if (!refill(d, x)) return DCVC_RANS_E_STREAM;
        int32_t v = s;
        if (s == sentinel) {
            int32_t nb = 0, cnt = 0;
            if (!take_nibble(d, x, nb)) return DCVC_RANS_E_STREAM;
            cnt = nb;
            while (nb == kNibbleMax) {
                if (!take_nibble(d, x, nb)) return DCVC_RANS_E_STREAM;
                cnt += nb;
            }
            // a 32-bit escape value needs at most 8 nibbles; a larger count can only come from a
            // corrupt stream (and would shift an int32 by >= 32 bits).  The device decoder applies
            // the same limit (rans_device.hip).
            if (cnt > 8) return DCVC_RANS_E_STREAM;
            uint32_t raw_u = 0;
            for (int32_t j = 0; j < cnt; ++j) {
                if (!take_nibble(d, x, nb)) return DCVC_RANS_E_STREAM;
                raw_u |= (uint32_t)nb << (j * kNibbleBits);
            }
            const int32_t raw = (int32_t)raw_u;
            v = raw >> 1;
            v = (raw & 1) ? -v - 1 : v + sentinel;
        }
        out[i] = v + offsets[row];
    }
    d->state = x;
    return 0;
}

extern "C" int dcvc_pmf_to_quantized_cdf(const float *pmf, int32_t n, int32_t precision, uint32_t *cdf) {
    if (!pmf || !cdf || n <= 0 || precision < 1 || precision > 31) return DCVC_RANS_E_ARG;
    const float scale = (float)(1u << precision);
    std::vector<uint32_t> f((size_t)n + 1);
    f[0] = 0;
    uint32_t total = 0;
    for (int32_t i = 0; i < n; ++i) {
        f[i + 1] = (uint32_t)(std::round(pmf[i] * scale) + 0.5);
        total += f[i + 1];
    }
    if (total == 0) return DCVC_RANS_E_ARG;
    uint32_t run = 0;
    for (int32_t i = 0; i <= n; ++i) {
        run += (uint32_t)((((uint64_t)1 << precision) * f[i]) / total);
        cdf[i] = run;
    }
    cdf[n] = 1u << precision;
    // every symbol needs a non-zero width: take one count from the narrowest bin wider than 1
    for (int32_t i = 0; i < n; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        uint32_t best = ~0u;
        int32_t donor = -1;
        for (int32_t j = 0; j < n; ++j) {
            const uint32_t w = cdf[j + 1] - cdf[j];
            if (w > 1 && w < best) {
                best = w;
                donor = j;
            }
        }
        if (donor < 0) return DCVC_RANS_E_ARG;
        if (donor < i)
            for (int32_t j = donor + 1; j <= i; ++j) cdf[j]--;
        else
            for (int32_t j = i + 1; j <= donor; ++j) cdf[j]++;
    }
    return 0;
}
