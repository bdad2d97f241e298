"""ORACLE (test infrastructure only) -- pure-Python restatement of the rANS wire format,
independent of both C implementations (oracle/rans_ref.c, vcm_ts_amd/csrc/rans.cpp).

Follows /root/reference/DCVC_HEM/src/cpp/rans/rans_interface.cpp:46-244 on the published
rans64.h update rule (see rans_ref.c header: the byte stream is PARITY UNPINNED against the
reference because ryg_rans is absent from its tree).  Slow: small inputs only.
"""
import struct

L = 1 << 31
MASK32 = (1 << 32) - 1


def encode(symbol_batches):
    """symbol_batches: list of (symbols, indexes, cdfs, sizes, offsets) -> bytes."""
    recs = []
    for symbols, indexes, cdfs, sizes, offsets in symbol_batches:
        for s, ci in zip(symbols, indexes):
            cdf = cdfs[ci]
            mx = int(sizes[ci]) - 2
            v = int(s) - int(offsets[ci])
            raw = None
            if v < 0:
                raw, v = -2 * v - 1, mx
            elif v >= mx:
                raw, v = 2 * (v - mx), mx
            recs.append((int(cdf[v]), int(cdf[v + 1]) - int(cdf[v]), False))
            if raw is not None:
                nb = 0
                while (raw >> (4 * nb)) != 0:
                    nb += 1
                val = nb
                while val >= 15:
                    recs.append((15, 16, True))
                    val -= 15
                recs.append((val, val + 1, True))
                for j in range(nb):
                    recs.append(((raw >> (4 * j)) & 15, 0, True))
    return flush_records(recs)


def flush_records(recs):
    """(start, freq, bypass) records in decode order -> bytes (rans_interface.cpp:147-172).  Tests
    also call it with forged records to build streams no encoder emits."""
    x = L
    words = []
    for start, freq, bypass in reversed(recs):
        if not bypass:
            if x >= ((L >> 16) << 32) * freq:
                words.append(x & MASK32)
                x >>= 32
            x = ((x // freq) << 16) + (x % freq) + start
        else:
            if x >= ((L >> 16) << 32) * (1 << 12):
                words.append(x & MASK32)
                x >>= 32
            x = (x << 4) | start
    words.append(x >> 32)
    words.append(x & MASK32)
    words.reverse()
    return struct.pack("<%dI" % len(words), *words)


class Decoder:
    def __init__(self, data):
        n = len(data) // 4
        self.w = struct.unpack("<%dI" % n, data[: 4 * n])
        self.x = self.w[0] | (self.w[1] << 32)
        self.p = 2

    def _renorm(self):
        if self.x < L:
            self.x = (self.x << 32) | self.w[self.p]
            self.p += 1

    def _bits(self):
        v = self.x & 15
        self.x >>= 4
        self._renorm()
        return v

    def decode(self, indexes, cdfs, sizes, offsets):
        out = []
        for ci in indexes:
            cdf = cdfs[ci]
            mx = int(sizes[ci]) - 2
            cum = self.x & 0xFFFF
            s = 0
            while s + 1 < sizes[ci] and cdf[s + 1] <= cum:
                s += 1
            start, freq = int(cdf[s]), int(cdf[s + 1]) - int(cdf[s])
            self.x = freq * (self.x >> 16) + (self.x & 0xFFFF) - start
            self._renorm()
            v = s
            if s == mx:
                val = self._bits()
                nb = val
                while val == 15:
                    val = self._bits()
                    nb += val
                if nb > 8:  # a 32-bit escape value has at most 8 nibbles: corrupt stream
                    raise ValueError("corrupt stream: escape of %d nibbles" % nb)
                raw = 0
                for j in range(nb):
                    raw |= self._bits() << (4 * j)
                v = raw >> 1
                v = -v - 1 if raw & 1 else v + mx
            out.append(v + int(offsets[ci]))
        return out
