// Developer probe (NOT part of libdcvc_hip.so): ablation builds of the pre-split convolution kernel.
//   probe bit 1: no LDS-DMA after the first step   2: no MFMA phase   4: no epilogue
// Build: make -C tools/probes ; used by tools/conv_probe.py --ablate
#include "../../vcm_ts_amd/csrc/conv_s16.hip"

extern "C" int dcvc_conv2d_s16_probe(const dcvc_conv_s16_args *a, int probe, void *stream) {
    S16K k;
    int CB = 0;
    const int rc = build_s16k(a, k, CB);
    if (rc != DCVC_OK || CB != 64) return rc ? rc : DCVC_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (probe) {
        case 0: return launch_s16<2, 0>(k, st);
        case 1: return launch_s16<2, 1>(k, st);
        case 2: return launch_s16<2, 2>(k, st);
        case 3: return launch_s16<2, 3>(k, st);
        case 4: return launch_s16<2, 4>(k, st);
        case 5: return launch_s16<2, 5>(k, st);
        case 6: return launch_s16<2, 6>(k, st);
        default: return DCVC_E_ARG;
    }
}
