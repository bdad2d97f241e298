/* ORACLE (test infrastructure only) -- plain-C restatement of the reference's entropy
 * coder for the DCVC-HEM bitstream.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or load this.
 *
 * Follows /root/reference/DCVC_HEM/src/cpp/rans/rans_interface.cpp:
 *   encode_with_indexes :85-145   flush :147-172   Rans64EncPutBits :46-64
 *   set_stream :176-182           decode_stream :184-244   Rans64DecGetBits :66-82
 * and /root/reference/DCVC_HEM/src/cpp/ops/ops.cpp:24-82 (pmf_to_quantized_cdf).
 *
 * PARITY UNPINNED for the byte stream: the reference includes <rans64.h> from the
 * third-party rygorous/ryg_rans pinned at commit c9d162d996fd600315af9ae8eb89d832576cb32d
 * (3rdparty/ryg_rans/CMakeLists.txt.in:8-9), which is fetched at cmake time and is absent
 * from the reference tree, and the reference holds no golden byte strings.  The six
 * primitives below (EncInit/EncPut/EncFlush/DecInit/DecGet/DecAdvance, RANS64_L = 2^31)
 * restate that header's published 64-bit rANS with 32-bit renormalisation; the in-tree
 * PutBits/GetBits helpers (cited above) fix the same conventions and are followed line
 * for line.  pmf_to_quantized_cdf IS pinned: tests compare it with the reference's own
 * ops.cpp compiled into oracle/_ref.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define RANS64_L (1ull << 31)
#define PRECISION 16
#define BYPASS_BITS 4
#define MAX_BYPASS ((1 << BYPASS_BITS) - 1)

typedef struct { uint16_t start, range; uint8_t bypass; } sym_t;

typedef struct {
    sym_t *v; size_t n, cap;
} ref_encoder;

ref_encoder *ref_enc_new(void) { return (ref_encoder *)calloc(1, sizeof(ref_encoder)); }
void ref_enc_free(ref_encoder *e) { if (e) { free(e->v); free(e); } }
void ref_enc_reset(ref_encoder *e) { e->n = 0; }

static void push(ref_encoder *e, uint16_t start, uint16_t range, int bypass) {
    if (e->n == e->cap) {
        e->cap = e->cap ? e->cap * 2 : 1024;
        e->v = (sym_t *)realloc(e->v, e->cap * sizeof(sym_t));
    }
    e->v[e->n].start = start; e->v[e->n].range = range; e->v[e->n].bypass = (uint8_t)bypass;
    e->n++;
}

/* rans_interface.cpp:85-145 */
void ref_enc_encode_with_indexes(ref_encoder *e, const int32_t *symbols, const int32_t *indexes, size_t n,
                                 const int32_t *cdfs, int cdf_stride, const int32_t *cdf_sizes,
                                 const int32_t *offsets) {
    for (size_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        const int32_t *cdf = cdfs + (size_t)ci * cdf_stride;
        const int32_t max_value = cdf_sizes[ci] - 2;
        int32_t value = symbols[i] - offsets[ci];
        uint32_t raw = 0;
        if (value < 0) { raw = (uint32_t)(-2 * value - 1); value = max_value; }
        else if (value >= max_value) { raw = (uint32_t)(2 * (value - max_value)); value = max_value; }
        push(e, (uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), 0);
        if (value == max_value) {
            int32_t nb = 0;
            while ((raw >> (nb * BYPASS_BITS)) != 0) ++nb;
            int32_t val = nb;
            while (val >= MAX_BYPASS) { push(e, MAX_BYPASS, MAX_BYPASS + 1, 1); val -= MAX_BYPASS; }
            push(e, (uint16_t)val, (uint16_t)(val + 1), 1);
            for (int32_t j = 0; j < nb; ++j) {
                int32_t v1 = (raw >> (j * BYPASS_BITS)) & MAX_BYPASS;
                push(e, (uint16_t)v1, (uint16_t)(v1 + 1), 1);
            }
        }
    }
}

/* published rans64.h: x' = (x / freq << bits) + x % freq + start, renormalising 32 bits */
static void enc_put(uint64_t *r, uint32_t **pp, uint32_t start, uint32_t freq, uint32_t bits) {
    uint64_t x = *r;
    uint64_t x_max = ((RANS64_L >> bits) << 32) * freq;
    if (x >= x_max) { *pp -= 1; **pp = (uint32_t)x; x >>= 32; }
    *r = ((x / freq) << bits) + (x % freq) + start;
}

/* rans_interface.cpp:46-64 */
static void enc_put_bits(uint64_t *r, uint32_t **pp, uint32_t val, uint32_t nbits) {
    uint64_t x = *r;
    uint32_t freq = 1u << (16 - nbits);
    uint64_t x_max = ((RANS64_L >> 16) << 32) * freq;
    if (x >= x_max) { *pp -= 1; **pp = (uint32_t)x; x >>= 32; }
    *r = (x << nbits) | val;
}

/* rans_interface.cpp:147-172.  Returns malloc'd bytes; *nbytes = length. */
uint8_t *ref_enc_flush(ref_encoder *e, size_t *nbytes) {
    size_t words = e->n + 2;              /* the reference sizes the buffer as n words; +2 keeps
                                             the final state in bounds for tiny inputs */
    uint32_t *buf = (uint32_t *)malloc(words * sizeof(uint32_t));
    uint32_t *ptr = buf + words;
    uint64_t r = RANS64_L;                /* Rans64EncInit */
    for (size_t i = e->n; i-- > 0;) {
        const sym_t s = e->v[i];
        if (!s.bypass) enc_put(&r, &ptr, s.start, s.range, PRECISION);
        else enc_put_bits(&r, &ptr, s.start, BYPASS_BITS);
    }
    ptr -= 2; ptr[0] = (uint32_t)r; ptr[1] = (uint32_t)(r >> 32);   /* Rans64EncFlush */
    *nbytes = (size_t)(buf + words - ptr) * sizeof(uint32_t);
    uint8_t *out = (uint8_t *)malloc(*nbytes ? *nbytes : 1);
    memcpy(out, ptr, *nbytes);
    free(buf);
    return out;
}

void ref_free(void *p) { free(p); }

typedef struct { uint32_t *buf, *ptr; uint64_t r; } ref_decoder;

ref_decoder *ref_dec_new(void) { return (ref_decoder *)calloc(1, sizeof(ref_decoder)); }
void ref_dec_free(ref_decoder *d) { if (d) { free(d->buf); free(d); } }

/* rans_interface.cpp:176-182 + Rans64DecInit */
void ref_dec_set_stream(ref_decoder *d, const uint8_t *bytes, size_t n) {
    free(d->buf);
    d->buf = (uint32_t *)malloc(n + 16);
    memset(d->buf, 0, n + 16);
    memcpy(d->buf, bytes, n);
    d->ptr = d->buf;
    d->r = (uint64_t)d->ptr[0] | ((uint64_t)d->ptr[1] << 32);
    d->ptr += 2;
}

/* rans_interface.cpp:66-82 */
static uint32_t dec_get_bits(ref_decoder *d, uint32_t nbits) {
    uint64_t x = d->r;
    uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
    x >>= nbits;
    if (x < RANS64_L) { x = (x << 32) | *d->ptr; d->ptr += 1; }
    d->r = x;
    return val;
}

/* rans_interface.cpp:184-244 */
void ref_dec_decode_stream(ref_decoder *d, const int32_t *indexes, size_t n, const int32_t *cdfs, int cdf_stride,
                           const int32_t *cdf_sizes, const int32_t *offsets, int32_t *out) {
    for (size_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        const int32_t *cdf = cdfs + (size_t)ci * cdf_stride;
        const int32_t max_value = cdf_sizes[ci] - 2;
        const uint32_t cum = (uint32_t)(d->r & ((1u << PRECISION) - 1));      /* Rans64DecGet */
        int32_t k = 0;
        while (k < cdf_sizes[ci] && (uint32_t)cdf[k] <= cum) ++k;              /* find_if(v > cum) */
        const uint32_t s = (uint32_t)(k - 1);
        {   /* Rans64DecAdvance */
            uint64_t x = d->r;
            uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
            x = freq * (x >> PRECISION) + (x & ((1ull << PRECISION) - 1)) - start;
            if (x < RANS64_L) { x = (x << 32) | *d->ptr; d->ptr += 1; }
            d->r = x;
        }
        int32_t value = (int32_t)s;
        if (value == max_value) {
            int32_t val = (int32_t)dec_get_bits(d, BYPASS_BITS);
            int32_t nb = val;
            while (val == MAX_BYPASS) { val = (int32_t)dec_get_bits(d, BYPASS_BITS); nb += val; }
            int32_t raw = 0;
            if (nb > 8) nb = 8;  /* corrupt stream (a 32-bit value has 8 nibbles): the reference is undefined here */
            for (int j = 0; j < nb; ++j) { val = (int32_t)dec_get_bits(d, BYPASS_BITS); raw |= val << (j * BYPASS_BITS); }
            value = raw >> 1;
            if (raw & 1) value = -value - 1; else value += max_value;
        }
        out[i] = value + offsets[ci];
    }
}

/* ops.cpp:24-82.  cdf must hold n+1 entries. */
void ref_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf) {
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)(roundf(pmf[i] * (float)(1 << precision)) + 0.5);
    uint32_t total = 0;
    for (int i = 0; i <= n; ++i) total += cdf[i];
    for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)(((1ull << precision) * cdf[i]) / total);
    for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
    cdf[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            uint32_t best_freq = ~0u; int best = -1;
            for (int j = 0; j < n; ++j) {
                uint32_t f = cdf[j + 1] - cdf[j];
                if (f > 1 && f < best_freq) { best_freq = f; best = j; }
            }
            if (best < 0) return;
            if (best < i) { for (int j = best + 1; j <= i; ++j) cdf[j]--; }
            else { for (int j = i + 1; j <= best; ++j) cdf[j]++; }
        }
    }
}
