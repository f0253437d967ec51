// jpeg_host.h — the serial half of the JPEG decoder (marker parsing + Huffman decoding), plain C++ so that the CPU
// sanitizers and fuzz tests can reach it (tests/test_sanitizers.py); the inverse DCT kernel is in jpeg.hip.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "../../include/jn_stereo.h"

namespace jnav {
const int kJpegMaxDim = 8192;   // frames larger than this are refused (a 65535x65535 header would ask for 8.6 GB of coefficients)
struct JpegFrame { int width = 0, height = 0, bw = 0, bh = 0; uint16_t quant[64]; };   // bw x bh luminance blocks (MCU-padded)
// Entropy-decodes the luminance coefficients (natural order, NOT dequantised) into coef [bh*bw][64].
jn_status jpeg_parse_and_decode(const uint8_t* data, size_t n, JpegFrame& out, std::vector<int16_t>& coef);
}  // namespace jnav
