// rans_device.hip -- lane-interleaved rANS64 on the GPU (include/dcvc_hip_rans.h, opt-in format).
//
// Same coder as csrc/rans.cpp (the reference's format, rans_interface.cpp:85-244), run by up to
// 1024 lanes per symbol plane, 64 per workgroup: lane j codes symbols j, j+L, ... into private
// scratch (rANS is last-in-first-out, so the lane walks its symbols backwards and fills its
// scratch from the end), a prefix sum of the lane sizes gives every stream its place
// and the workgroups copy the streams behind the section header.  Integer / byte work, a few
// hundred symbols per lane: latency-bound, tens of microseconds per plane, and it runs on the
// launch stream between the network kernels, so the symbol planes never leave HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "dcvc_hip_rans.h"

namespace {

constexpr uint64_t kLower = 1ull << 31;
constexpr int kProbBits = 16, kNibbleBits = 4, kNibbleMax = 15;

struct Tables {
    const int32_t *cdfs;
    const int32_t *sizes;
    const int32_t *offsets;
    int n_cdfs, stride;
};

// The CDF table of the call is copied into LDS as 16-bit values first (256 x 103 entries = 52 KB for
// the scale table): the serial per-lane loops then chase LDS, not L2.  The one value that does not
// fit 16 bits, 65536 at the end of a row, wraps to 0; widths are taken mod 2^16 (a width is < 65536
// because every row has at least two non-empty bins) and the decoder never compares the last entry.
//
// Work split: a workgroup is ONE wave of 64 lanes (the per-symbol state update is ~100-250 serial
// instructions, so lanes of one plane are spread over lanes/64 compute units instead of queueing
// on one); every workgroup rebuilds the lane-size prefix it needs from the section header.
extern __shared__ uint16_t lds_dyn[];
constexpr int WG = 64;      // coding lanes per workgroup (wave 0)
constexpr int STAGE = 256;  // threads per workgroup: waves 1-3 only help staging the tables, then leave

__device__ __forceinline__ int tab_halves(const Tables &T) { return (T.n_cdfs * T.stride + 7) & ~7; }

// LDS image: [n_cdfs * stride] u16 table | [n_cdfs] i32 sizes | [n_cdfs] i32 offsets | (decoder) bucket LUT
__device__ __forceinline__ void load_tables(const Tables &T, uint16_t *tab, int32_t *&sizes, int32_t *&offsets) {
    const int total = T.n_cdfs * T.stride;
    sizes = (int32_t *)(tab + tab_halves(T));
    offsets = sizes + T.n_cdfs;
    // 16-byte loads, 8 in flight per thread: a handful of memory round trips for the whole table
    const int quads = total >> 2;
    const int4 *src = (const int4 *)T.cdfs;
#pragma unroll 8
    for (int i = threadIdx.x; i < quads; i += blockDim.x) {
        const int4 v = src[i];
        ushort4 h;
        h.x = (uint16_t)v.x;
        h.y = (uint16_t)v.y;
        h.z = (uint16_t)v.z;
        h.w = (uint16_t)v.w;
        *(ushort4 *)&tab[i * 4] = h;
    }
    for (int i = quads * 4 + threadIdx.x; i < total; i += blockDim.x) tab[i] = (uint16_t)T.cdfs[i];
    for (int i = threadIdx.x; i < T.n_cdfs; i += blockDim.x) {
        sizes[i] = T.sizes[i];
        offsets[i] = T.offsets[i];
    }
    __syncthreads();
}

// x / freq and x % freq for x < 2^63, freq < 2^16 by three 32-bit divisions (the compiler's generic
// 64-bit division is several times longer, and this sits on the serial path of every symbol)
__device__ __forceinline__ void divmod(uint64_t x, uint32_t freq, uint64_t &q, uint32_t &r) {
    const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
    const uint32_t q1 = hi / freq, r1 = hi - q1 * freq;
    const uint32_t t2 = (r1 << 16) | (lo >> 16);
    const uint32_t q2 = t2 / freq, r2 = t2 - q2 * freq;
    const uint32_t t3 = (r2 << 16) | (lo & 0xFFFFu);
    const uint32_t q3 = t3 / freq;
    r = t3 - q3 * freq;
    q = ((uint64_t)q1 << 32) + ((uint64_t)q2 << 16) + q3;
}

__device__ __forceinline__ void put(uint64_t &x, uint32_t *&p, uint32_t start, uint32_t freq) {
    const uint64_t lim = ((kLower >> kProbBits) << 32) * freq;
    if (x >= lim) {
        *--p = (uint32_t)x;
        x >>= 32;
    }
    uint64_t q;
    uint32_t r;
    divmod(x, freq, q, r);
    x = (q << kProbBits) + r + start;
}

__device__ __forceinline__ void put_nibble(uint64_t &x, uint32_t *&p, uint32_t v) {
    const uint64_t lim = ((kLower >> 16) << 32) * (uint64_t)(1u << (16 - kNibbleBits));
    if (x >= lim) {
        *--p = (uint32_t)x;
        x >>= 32;
    }
    x = (x << kNibbleBits) | v;
}

__device__ __forceinline__ int row_of(const int32_t *__restrict__ idx, int i, int chan_hw, int chan_c) {
    return idx ? idx[i] : (i / chan_hw) % chan_c;
}

// pass 1: every lane codes its symbols into its private scratch slice (filled from the end)
__global__ __launch_bounds__(STAGE) void drans_encode_lanes(const int32_t *__restrict__ sym, const int32_t *__restrict__ idx, int chan_hw,
                                   int chan_c, int n, int L, Tables T, uint32_t *__restrict__ scratch, int cap,
                                   uint32_t *__restrict__ lane_words, int32_t *status) {
    uint16_t *tab = lds_dyn;
    int32_t *lsizes, *loffsets;
    load_tables(T, tab, lsizes, loffsets);
    if (threadIdx.x >= WG) return;
    const int lane = blockIdx.x * WG + threadIdx.x;
    if (lane >= L) return;
    uint32_t *end = scratch + (size_t)(lane + 1) * cap, *p = end;
    uint32_t *floor_ = scratch + (size_t)lane * cap + 4;  // room for one worst-case symbol + the final state
    uint64_t x = kLower;
    int bad = 0;
    const int cnt = lane < n ? (n - lane + L - 1) / L : 0;
    // The symbol / index loads do not depend on the coder state: they are issued a batch of B at a
    // time so that the serial loop waits for global memory once per batch, not once per symbol.
    constexpr int B = 8;
    bool full = false;
    for (int k0 = cnt - 1; k0 >= 0 && !full; k0 -= B) {
        int rows[B], syms[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int k = k0 - b;
            const int i = lane + (k < 0 ? 0 : k) * L;
            syms[b] = k >= 0 ? sym[i] : 0;
            rows[b] = k >= 0 ? row_of(idx, i, chan_hw, chan_c) : 0;
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            if (k0 - b < 0 || full) continue;
            const int row = rows[b], sv = syms[b];
            if ((unsigned)row >= (unsigned)T.n_cdfs) {
                bad |= DCVC_DRANS_BAD_INDEX;
                continue;
            }
            if (p < floor_) {
                bad |= DCVC_DRANS_BAD_SPACE;
                full = true;
                continue;
            }
            const uint16_t *cdf = tab + row * T.stride;
            const int sentinel = lsizes[row] - 2;
            int v = sv - loffsets[row];
            uint32_t raw = 0;
            bool escape = false;
            if (v < 0) {
                raw = (uint32_t)(-2 * (int64_t)v - 1);
                escape = true;
            } else if (v >= sentinel) {
                raw = (uint32_t)(2 * ((int64_t)v - sentinel));
                escape = true;
            }
            if (escape) {  // records of this symbol in reverse: raw nibbles, count nibble, then the sentinel
                v = sentinel;
                int nib = 0;
                while (nib < 8 && (raw >> (nib * kNibbleBits)) != 0) ++nib;
                for (int j = nib - 1; j >= 0; --j) put_nibble(x, p, (raw >> (j * kNibbleBits)) & kNibbleMax);
                put_nibble(x, p, (uint32_t)nib);  // nib <= 8 < 15: one count nibble (the host coder's loop never iterates either)
            }
            put(x, p, (uint32_t)cdf[v], (uint32_t)(uint16_t)(cdf[v + 1] - cdf[v]));
        }
    }
    *--p = (uint32_t)(x >> 32);
    *--p = (uint32_t)x;
    lane_words[lane] = (uint32_t)(end - p);
    if (bad) atomicOr(status, bad);
}

// sum of words[0 .. first) by the 64 threads of the workgroup (every thread gets the result)
__device__ uint32_t prefix_before(const uint32_t *__restrict__ words, int first, uint32_t *sm) {
    uint32_t v = 0;
    for (int i = threadIdx.x; i < first; i += WG) v += words[i];
    sm[threadIdx.x] = v;
    __syncthreads();
    uint32_t s = 0;
    for (int k = 0; k < WG; ++k) s += sm[k];
    __syncthreads();
    return s;
}

// pass 2: place every lane's stream behind the section header at payload[*cursor_in]
__global__ __launch_bounds__(WG) void drans_pack_section(const uint32_t *__restrict__ scratch, int cap, const uint32_t *__restrict__ lane_words,
                                   int n, int L, uint32_t *__restrict__ payload, int64_t payload_words,
                                   const int32_t *cursor_in, int32_t *cursor_out, int32_t *status) {
    __shared__ uint32_t sm[WG];
    const int first = blockIdx.x * WG, lane = first + threadIdx.x;
    const uint32_t total = prefix_before(lane_words, L, sm);
    uint32_t off = prefix_before(lane_words, first, sm);
    const uint32_t mine = lane < L ? lane_words[lane] : 0u;
    sm[threadIdx.x] = mine;
    __syncthreads();
    for (int k = 0; k < (int)threadIdx.x; ++k) off += sm[k];
    const int64_t base = *cursor_in;
    const int64_t need = base + 2 + L + (int64_t)total;
    if (need > payload_words) {
        if (lane == 0) {
            atomicOr(status, DCVC_DRANS_BAD_SPACE);
            *cursor_out = (int32_t)base;
        }
        return;
    }
    uint32_t *sec = payload + base;
    if (lane == 0) {
        sec[0] = (uint32_t)n;
        sec[1] = (uint32_t)L;
        *cursor_out = (int32_t)need;
    }
    if (lane >= L) return;
    sec[2 + lane] = mine;
    const uint32_t *src = scratch + (size_t)(lane + 1) * cap - mine;
    uint32_t *dst = sec + 2 + L + off;
    for (uint32_t k = 0; k < mine; ++k) dst[k] = src[k];
}

__global__ __launch_bounds__(STAGE) void drans_decode_kernel(const uint32_t *__restrict__ payload, int64_t payload_words, const int32_t *cursor_in,
                                    int32_t *cursor_out, const int32_t *__restrict__ idx, int chan_hw, int chan_c, int n,
                                    int L, Tables T, const uint8_t *__restrict__ lut_g, int32_t *__restrict__ out,
                                    int32_t *status) {
    __shared__ uint32_t sm[WG];
    uint16_t *tab = lds_dyn;
    int32_t *lsizes, *loffsets;
    // lut[row][cum >> 8]: the bin holding cumulative count (bucket << 8); the search starts there
    uint8_t *lut = (uint8_t *)((int32_t *)(tab + tab_halves(T)) + 2 * T.n_cdfs);
#pragma unroll 8
    for (int e = threadIdx.x; e < T.n_cdfs * 64; e += blockDim.x) ((uint32_t *)lut)[e] = ((const uint32_t *)lut_g)[e];
    load_tables(T, tab, lsizes, loffsets);
    if (threadIdx.x >= WG) return;
    const int first = blockIdx.x * WG, lane = first + threadIdx.x;
    const int64_t base = *cursor_in;
    bool header_ok = base + 2 + L <= payload_words;
    const uint32_t *sec = payload + base;
    if (header_ok) header_ok = sec[0] == (uint32_t)n && sec[1] == (uint32_t)L;
    if (!header_ok) {
        if (lane == 0) {
            atomicOr(status, DCVC_DRANS_BAD_STREAM);
            *cursor_out = (int32_t)base;
        }
        return;
    }
    const uint32_t *lane_words = sec + 2;
    const uint32_t total = prefix_before(lane_words, L, sm);
    uint32_t off = prefix_before(lane_words, first, sm);
    const uint32_t words = lane < L ? lane_words[lane] : 0u;
    sm[threadIdx.x] = words;
    __syncthreads();
    for (int k = 0; k < (int)threadIdx.x; ++k) off += sm[k];
    const int64_t need = base + 2 + L + (int64_t)total;
    if (need > payload_words) {
        if (lane == 0) {
            atomicOr(status, DCVC_DRANS_BAD_STREAM);
            *cursor_out = (int32_t)base;
        }
        return;
    }
    if (lane == 0) *cursor_out = (int32_t)need;
    if (lane >= L) return;
    if (words < 2) {
        atomicOr(status, DCVC_DRANS_BAD_STREAM);
        return;
    }
    const uint32_t *w = sec + 2 + L + off, *wend = w + words;
    uint64_t x = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    w += 2;
    int bad = 0;
    // two words of the lane's stream are always in flight ahead of the state (each lane reads its
    // own scattered words: the latency, not the bandwidth, is what the serial loop would wait for)
    uint32_t w0 = w < wend ? w[0] : 0u, w1 = w + 1 < wend ? w[1] : 0u;
    auto refill = [&]() {
        if (x < kLower) {
            if (w >= wend) {
                bad |= DCVC_DRANS_BAD_STREAM;
                x |= kLower;  // keep going on garbage rather than reading out of bounds
            } else {
                x = (x << 32) | w0;
                ++w;
                w0 = w1;
                w1 = w + 1 < wend ? w[1] : 0u;
            }
        }
    };
    auto nibble = [&]() {
        const int v = (int)(x & kNibbleMax);
        x >>= kNibbleBits;
        refill();
        return v;
    };
    // Symbols are taken in batches of B: the batch's CDF rows are loaded together and its results
    // stored together, so the loop waits for global memory once per batch instead of once per symbol
    // (vmcnt retires in order: a wait for one load is a wait for everything issued before it).
    constexpr int B = 8;
    for (int i0 = lane; i0 < n; i0 += B * L) {
        int rows[B], res[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int i = i0 + b * L;
            rows[b] = i < n ? row_of(idx, i, chan_hw, chan_c) : 0;
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int i = i0 + b * L;
            res[b] = 0;
            if (i >= n) continue;
            const int row = rows[b];
            if ((unsigned)row >= (unsigned)T.n_cdfs) {
                bad |= DCVC_DRANS_BAD_INDEX;
                continue;
            }
            const uint16_t *cdf = tab + row * T.stride;
            const int size = lsizes[row], sentinel = size - 2;
            const uint32_t cum = (uint32_t)(x & 0xFFFFu);
            int s = lut[(row << 8) + (cum >> 8)];
            // entry size-1 is 65536 > cum and is never compared; the four reads below are independent
            // (one LDS round trip) and settle all but very flat rows
            {
                const uint32_t c1 = cdf[min(s + 1, size - 2)], c2 = cdf[min(s + 2, size - 2)];
                const uint32_t c3 = cdf[min(s + 3, size - 2)], c4 = cdf[min(s + 4, size - 2)];
                const int lim = size - 2 - s;  // how far s may still move
                int adv = 0;
                if (lim > 0 && c1 <= cum) adv = 1;
                if (lim > 1 && adv == 1 && c2 <= cum) adv = 2;
                if (lim > 2 && adv == 2 && c3 <= cum) adv = 3;
                if (lim > 3 && adv == 3 && c4 <= cum) adv = 4;
                s += adv;
                if (adv == 4)
                    while (s + 2 < size && (uint32_t)cdf[s + 1] <= cum) ++s;
            }
            const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(uint16_t)(cdf[s + 1] - cdf[s]);
            x = (uint64_t)freq * (x >> kProbBits) + (x & 0xFFFFu) - start;
            refill();
            int v = s;
            if (s == sentinel) {
                int nb = nibble(), c = nb;
                while (nb == kNibbleMax && c < 64) {
                    nb = nibble();
                    c += nb;
                }
                if (c > 8) bad |= DCVC_DRANS_BAD_STREAM;  // a 32-bit escape has 8 nibbles (same limit as rans.cpp)
                uint32_t raw_u = 0;
                for (int j = 0; j < c && j < 8; ++j) raw_u |= (uint32_t)nibble() << (j * kNibbleBits);
                const int raw = (int)raw_u;
                v = raw >> 1;
                v = (raw & 1) ? -v - 1 : v + sentinel;
            }
            res[b] = v + loffsets[row];
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int i = i0 + b * L;
            if (i < n) out[i] = res[b];
        }
    }
    if (bad) atomicOr(status, bad);
}

inline int per_lane_cap(int64_t n, int lanes) { return (int)(((n + lanes - 1) / lanes) * 2 + 8); }

constexpr int kMaxTableEntries = 32 * 1024;  // 64 KB of LDS for the 16-bit table
constexpr int kMaxRows = 256;                // 64 KB for the decoder's bucket table

// The dynamic-LDS ceiling of a kernel is a process-wide property of the function: raise it once to
// the most this file ever asks for (callers may be on several host threads with different tables,
// so setting it per launch to the launch's own size would race).
constexpr size_t kLdsCeiling = 64 * 1024 + 2 * 1024 + 64 * 1024;  // table + sizes/offsets + bucket table

template <typename K>
int set_lds(K kernel, size_t bytes) {
    static std::once_flag once;
    static int status = DCVC_OK;
    std::call_once(once, [&] {
        status = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsCeiling) == hipSuccess
                     ? DCVC_OK : DCVC_E_LAUNCH;
    });
    return bytes <= kLdsCeiling ? status : DCVC_E_ARG;
}

inline size_t table_lds(int n_cdfs, int stride) { return (((size_t)n_cdfs * stride + 7) & ~(size_t)7) * 2 + (size_t)n_cdfs * 8; }

}  // namespace

extern "C" int32_t dcvc_drans_default_lanes(int64_t n) {
    int64_t l = (n + 511) / 512;
    l = (l + 63) / 64 * 64;
    return (int32_t)(l < 64 ? 64 : (l > 1024 ? 1024 : l));
}

extern "C" int64_t dcvc_drans_scratch_words(int64_t n, int32_t lanes) {
    if (n < 0 || lanes < 1 || lanes > DCVC_DRANS_MAX_LANES) return DCVC_E_ARG;
    return (int64_t)lanes * per_lane_cap(n, lanes) + lanes;
}

extern "C" int dcvc_drans_build_lut(const int32_t *cdfs, int32_t n_cdfs, int32_t cdf_stride, const int32_t *cdf_sizes,
                                    uint8_t *lut) {
    if (!cdfs || !cdf_sizes || !lut || n_cdfs <= 0 || n_cdfs > kMaxRows || cdf_stride < 2 || cdf_stride > 256) return DCVC_E_ARG;
    for (int row = 0; row < n_cdfs; ++row) {
        const int size = cdf_sizes[row];
        if (size < 2 || size > cdf_stride) return DCVC_E_ARG;
        const int32_t *cdf = cdfs + (size_t)row * cdf_stride;
        int s = 0;
        for (int b = 0; b < 256; ++b) {
            while (s + 2 < size && (uint32_t)cdf[s + 1] <= (uint32_t)b << 8) ++s;
            lut[row * 256 + b] = (uint8_t)s;
        }
    }
    return DCVC_OK;
}

extern "C" int dcvc_drans_encode(const int32_t *sym, const int32_t *idx, int32_t chan_hw, int32_t chan_c, int64_t n,
                                 const int32_t *cdfs, int32_t n_cdfs, int32_t cdf_stride, const int32_t *cdf_sizes,
                                 const int32_t *offsets, int32_t lanes, uint32_t *scratch, int64_t scratch_words,
                                 uint32_t *payload, int64_t payload_words, const int32_t *cursor_in, int32_t *cursor_out,
                                 int32_t *status, void *stream) {
    if (!sym || !cdfs || !cdf_sizes || !offsets || !scratch || !payload || !cursor_in || !cursor_out || cursor_in == cursor_out ||
        !status || n <= 0 || n >= (1ll << 31) || lanes < 1 || lanes > DCVC_DRANS_MAX_LANES || n_cdfs <= 0 || cdf_stride < 2)
        return DCVC_E_ARG;
    if (!idx && (chan_hw <= 0 || chan_c <= 0)) return DCVC_E_ARG;
    if (scratch_words < dcvc_drans_scratch_words(n, lanes)) return DCVC_E_ARG;
    if ((int64_t)n_cdfs * cdf_stride > kMaxTableEntries || ((uintptr_t)cdfs & 15)) return DCVC_E_ARG;
    Tables T = {cdfs, cdf_sizes, offsets, n_cdfs, cdf_stride};
    const size_t lds = table_lds(n_cdfs, cdf_stride);
    if (set_lds(drans_encode_lanes, lds)) return DCVC_E_LAUNCH;
    const int cap = per_lane_cap(n, lanes);
    uint32_t *lane_words = scratch + (size_t)lanes * cap;
    const unsigned blocks = (unsigned)((lanes + WG - 1) / WG);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(drans_encode_lanes, dim3(blocks), dim3(STAGE), lds, st, sym, idx, chan_hw, chan_c, (int)n, lanes, T,
                       scratch, cap, lane_words, status);
    hipLaunchKernelGGL(drans_pack_section, dim3(blocks), dim3(WG), 0, st, scratch, cap, lane_words, (int)n, lanes, payload,
                       payload_words, cursor_in, cursor_out, status);
    return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH;
}

extern "C" int dcvc_drans_decode(const uint32_t *payload, int64_t payload_words, const int32_t *cursor_in,
                                 int32_t *cursor_out, const int32_t *idx, int32_t chan_hw, int32_t chan_c, int64_t n,
                                 const int32_t *cdfs, int32_t n_cdfs, int32_t cdf_stride, const int32_t *cdf_sizes,
                                 const int32_t *offsets, const uint8_t *lut, int32_t lanes, int32_t *out, int32_t *status,
                                 void *stream) {
    if (!payload || !cursor_in || !cursor_out || cursor_in == cursor_out || !cdfs || !cdf_sizes || !offsets || !lut || !out ||
        !status || n <= 0 || n >= (1ll << 31) || n_cdfs <= 0 || cdf_stride < 2 || lanes < 1 || lanes > DCVC_DRANS_MAX_LANES)
        return DCVC_E_ARG;
    if (!idx && (chan_hw <= 0 || chan_c <= 0)) return DCVC_E_ARG;
    if ((int64_t)n_cdfs * cdf_stride > kMaxTableEntries || n_cdfs > kMaxRows || cdf_stride > 256 || ((uintptr_t)cdfs & 15) ||
        ((uintptr_t)lut & 3))
        return DCVC_E_ARG;
    Tables T = {cdfs, cdf_sizes, offsets, n_cdfs, cdf_stride};
    const size_t lds = table_lds(n_cdfs, cdf_stride) + (size_t)n_cdfs * 256;
    if (set_lds(drans_decode_kernel, lds)) return DCVC_E_LAUNCH;
    hipLaunchKernelGGL(drans_decode_kernel, dim3((unsigned)((lanes + WG - 1) / WG)), dim3(STAGE), lds, (hipStream_t)stream, payload,
                       payload_words, cursor_in, cursor_out, idx, chan_hw, chan_c, (int)n, lanes, T, lut, out, status);
    return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH;
}
