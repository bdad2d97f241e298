// rans_fuzz.cpp -- memory-safety fuzz of the host entropy coder through its C ABI (include/dcvc_rans.h), built with
// -fsanitize=address,undefined by tests/test_rans.py (CPU only).  A .bin file comes from disk: the decoder must answer a
// truncated or corrupted payload with a status code (or with wrong symbols), never with an out-of-bounds access --
// the reference's decoder has assert-only checks (rans_interface.cpp:184-244).
//   rounds of: random tables + random symbol planes (in-table and escape-coded) -> encode -> decode == symbols;
//   then the same payload truncated / bit-flipped / extended with garbage / with its escape counts inflated -> decode.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "dcvc_rans.h"

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 1);
    auto uni = [&](int lo, int hi) { return (int)(lo + rng() % (uint64_t)(hi - lo + 1)); };
    long decoded_ok = 0, refused = 0, wrong = 0;
    for (int r = 0; r < rounds; ++r) {
        // tables: n_cdfs rows of random pmfs through the product's own quantiser
        const int n_cdfs = uni(1, 12), stride = uni(4, 70);
        std::vector<int32_t> cdfs((size_t)n_cdfs * stride, 0), sizes(n_cdfs), offs(n_cdfs);
        for (int i = 0; i < n_cdfs; ++i) {
            const int nsym = uni(2, stride - 1);  // symbols incl. the sentinel slot
            std::vector<float> pmf(nsym);
            float tot = 0;
            for (float &p : pmf) tot += (p = (float)(1 + rng() % 1000));
            for (float &p : pmf) p /= tot;
            std::vector<uint32_t> q(nsym + 1);
            if (dcvc_pmf_to_quantized_cdf(pmf.data(), nsym, 16, q.data()) != 0) return 2;
            for (int k = 0; k <= nsym; ++k) cdfs[(size_t)i * stride + k] = (int32_t)q[k];
            sizes[i] = nsym + 1;
            offs[i] = -uni(0, nsym);
        }
        const int64_t n = uni(0, 3000);
        std::vector<int32_t> sym(n), idx(n), out(n + 8, 0x55aa55aa);
        for (int64_t k = 0; k < n; ++k) {
            idx[k] = uni(0, n_cdfs - 1);
            const int span = sizes[idx[k]] - 2;  // in-table values are offset .. offset + span - 1
            sym[k] = (rng() % 10 == 0) ? uni(-4000, 4000) : uni(0, span > 0 ? span - 1 : 0) + offs[idx[k]];  // (value = symbol - offset)
        }
        dcvc_rans_encoder *e = dcvc_rans_encoder_create();
        dcvc_rans_decoder *d = dcvc_rans_decoder_create();
        if (!e || !d) return 2;
        if (dcvc_rans_encoder_encode_with_indexes(e, sym.data(), idx.data(), n, cdfs.data(), n_cdfs, stride, sizes.data(), offs.data()) != 0) return 3;
        const int64_t cap = dcvc_rans_encoder_flush_bound(e);
        if (cap < 0) return 3;
        std::vector<uint8_t> buf((size_t)cap);
        const int64_t nb = dcvc_rans_encoder_flush(e, buf.data(), cap);
        if (nb < 0 || nb > cap) return 3;
        // exact-size heap copy: any over-read of the stream is an ASan report
        auto decode = [&](const std::vector<uint8_t> &bytes, int64_t want_n) {
            uint8_t *heap = (uint8_t *)malloc(bytes.size() ? bytes.size() : 1);
            if (!bytes.empty()) memcpy(heap, bytes.data(), bytes.size());
            int rc = dcvc_rans_decoder_set_stream(d, heap, (int64_t)bytes.size());
            if (rc == 0) {
                std::vector<int32_t> o((size_t)want_n + 1, 0);
                std::vector<int32_t> ix(idx);
                ix.resize((size_t)want_n, 0);
                rc = dcvc_rans_decoder_decode_stream(d, ix.data(), want_n, cdfs.data(), n_cdfs, stride, sizes.data(), offs.data(), o.data());
                if (rc == 0 && want_n == n) rc = memcmp(o.data(), sym.data(), (size_t)n * 4) == 0 ? 0 : 1;
            }
            dcvc_rans_decoder_set_stream(d, heap, 0);  // the decoder must not keep reading freed memory afterwards
            free(heap);
            return rc;
        };
        std::vector<uint8_t> good(buf.begin(), buf.begin() + nb);
        if (decode(good, n) != 0) {
            fprintf(stderr, "round %d: clean round trip failed\n", r);
            return 4;
        }
        ++decoded_ok;
        for (int v = 0; v < 12; ++v) {
            std::vector<uint8_t> bad = good;
            switch (v % 4) {
                case 0: bad.resize(bad.size() ? rng() % bad.size() : 0); break;                       // truncated
                case 1: for (int k = 0; k < 1 + (int)(rng() % 8) && !bad.empty(); ++k) bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() % 8)); break;
                case 2: for (int k = 0; k < 16; ++k) bad.push_back((uint8_t)rng()); break;            // trailing garbage
                case 3: for (auto &b : bad) if (rng() % 6 == 0) b = 0xff; break;                      // long escape runs
            }
            const int rc = decode(bad, (v % 3 == 0) ? n + uni(1, 50) : n);  // sometimes ask for more symbols than were coded
            if (rc < 0) ++refused; else if (rc == 1) ++wrong; else ++decoded_ok;
        }
        dcvc_rans_encoder_destroy(e);
        dcvc_rans_decoder_destroy(d);
    }
    printf("rans_fuzz: %d rounds, %ld decodes fine, %ld refused with a status, %ld decoded to other symbols\n", rounds, decoded_ok, refused, wrong);
    return 0;
}
