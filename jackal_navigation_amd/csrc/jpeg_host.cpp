// jpeg_host.cpp + jpeg.hip — cv::imdecode(data, CV_LOAD_IMAGE_GRAYSCALE) for the node's compressed camera frames
// (src/obstacle_avoidance/point_cloud.cpp:436, :478), product code.
//
// OpenCV hands JPEG data to libjpeg(-turbo) with out_color_space = JCS_GRAYSCALE, which entropy-decodes every component
// but reconstructs only luminance, with the default "slow integer" inverse DCT.  Neither OpenCV nor libjpeg is part of
// the reference tree (third-party arithmetic); this file restates the published baseline-JPEG decoding
// procedure (ITU-T T.81: marker syntax, Huffman decoding, DC prediction, zig-zag order, restart intervals) and the
// Loeffler-Ligtenberg-Moschytz 8x8 inverse DCT in the 13-bit fixed-point form the Independent JPEG Group's "islow" method
// defines (CONST_BITS 13, PASS1_BITS 2, range limiting modulo 1024 around +128), so that the grey image equals what
// libjpeg produces bit for bit.  Pinned by fixtures generated with Pillow (libjpeg-turbo, the same IDCT) in
// tests/golden/make_jpeg_golden.py.
//
// Split: the entropy decoder is inherently serial per scan and runs on the calling host thread; coefficients of the
// luminance blocks go to pinned memory, dequantisation + inverse DCT + range limit run on the GPU (8 lanes per block, the
// 8x8 workspace in LDS), the grey image stays in device memory for jn_remap_bilinear / jn_elas_*.
// Supported: baseline sequential DCT (SOF0) and extended sequential with 8-bit samples (SOF1), Huffman coding, 1 or 3
// components, luminance sampling factors 1 or 2, restart intervals.  Progressive, arithmetic-coded, 12-bit, lossless and
// multi-scan files return JN_ERR_UNSUPPORTED.
#include "jpeg_host.h"
#include <cstdio>
#include <cstring>
#include <new>

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool present = false;
  uint8_t vals[256];
  int32_t mincode[17], maxcode[18], valptr[17];     // per code length 1..16 (T.81 F.2.2.3)
  uint8_t look_len[256], look_val[256];             // first 8 bits -> (length, value) for codes of <= 8 bits
  // Returns false for a table that is not a prefix code (more codes of some length than that length has room for, or the
  // all-ones code in use): libjpeg's JERR_BAD_HUFF_TABLE check.  Without it `code` outgrows 2^len and the look-up index
  // below runs past look_len/look_val.
  bool build(const uint8_t counts[16], const uint8_t* symbols, int nsym) {
    present = false;
    if (nsym < 0 || nsym > 256) return false;
    memcpy(vals, symbols, (size_t)nsym);
    int code = 0, k = 0;
    memset(look_len, 0, sizeof(look_len));
    for (int len = 1; len <= 16; len++) {
      valptr[len] = k; mincode[len] = code;
      if (code + (int)counts[len - 1] >= (1 << len) && counts[len - 1]) return false;   // codes of this length would reach 2^len - 1
      for (int i = 0; i < counts[len - 1]; i++, k++, code++)
        if (len <= 8)
          for (int fill = 0; fill < (1 << (8 - len)); fill++) { const int idx = (code << (8 - len)) | fill; look_len[idx] = (uint8_t)len; look_val[idx] = vals[k]; }
      maxcode[len] = counts[len - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    present = true;
    return true;
  }
};

struct BitReader {
  const uint8_t* p; const uint8_t* end;
  uint64_t acc = 0; int bits = 0;                     // the next `bits` bits of the stream sit at the top of acc
  bool hit_marker = false;
  // Keep >= 33 bits when possible (a symbol of <= 16 bits and its <= 16 extra bits need no second refill).  Four bytes at a
  // time while none of them is 0xFF; byte-wise around stuffed bytes (0xFF00 -> 0xFF) and markers (a marker feeds zeros).
  inline void fill() {
    if (bits > 32) return;
    if (!hit_marker && p + 4 <= end) {
      uint32_t w; memcpy(&w, p, 4);
      const uint32_t inv = ~w;
      if (((inv - 0x01010101u) & ~inv & 0x80808080u) == 0) {               // no byte of w is 0xFF
        acc |= (uint64_t)__builtin_bswap32(w) << (32 - bits);
        bits += 32; p += 4;
        return;
      }
    }
    while (bits <= 56) {
      int b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) p += 2;
          else { hit_marker = true; b = 0; }
        } else p++;
      } else hit_marker = true;
      acc |= (uint64_t)b << (56 - bits);
      bits += 8;
    }
  }
  inline int peek(int n) const { return (int)(acc >> (64 - n)); }
  inline void drop(int n) { acc <<= n; bits -= n; }
  void restart() { acc = 0; bits = 0; hit_marker = false; }
};

// One Huffman symbol; the caller has filled the reader.
inline int decode_symbol(BitReader& br, const Huff& h) {
  const int top = br.peek(8);
  if (h.look_len[top]) { br.drop(h.look_len[top]); return h.look_val[top]; }
  int code = top, len = 8;
  do { len++; code = br.peek(len); } while (len <= 16 && code > h.maxcode[len]);
  if (len > 16) return -1;
  br.drop(len);
  return h.vals[(h.valptr[len] + code - h.mincode[len]) & 255];
}
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }   // T.81 F.2.2.1

struct Component { int id, h, v, tq, td, ta; };

using Decoded = jnav::JpegFrame;

}  // namespace

jn_status jnav::jpeg_parse_and_decode(const uint8_t* data, size_t n, Decoded& out, std::vector<int16_t>& coef) {
  if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) return JN_ERR_INVALID;
  uint16_t qt[4][64]; bool qt_ok[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  Component comp[3]; int ncomp = 0;
  int restart_interval = 0;
  bool have_frame = false;
  size_t pos = 2;
  while (pos + 4 <= n) {
    if (data[pos] != 0xFF) { pos++; continue; }
    const int m = data[pos + 1];
    if (m == 0xFF) { pos++; continue; }
    pos += 2;
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (m == 0xD9) break;
    if (pos + 2 > n) return JN_ERR_INVALID;
    const size_t len = ((size_t)data[pos] << 8) | data[pos + 1];
    if (len < 2 || pos + len > n) return JN_ERR_INVALID;
    const uint8_t* seg = data + pos + 2; const size_t slen = len - 2;
    if (m == 0xDB) {                                                             // DQT
      size_t i = 0;
      while (i < slen) {
        const int pq = seg[i] >> 4, tq = seg[i] & 15; i++;
        if (tq > 3 || i + (pq ? 128 : 64) > slen) return JN_ERR_INVALID;
        for (int k = 0; k < 64; k++) { qt[tq][kZigzag[k]] = pq ? (uint16_t)((seg[i] << 8) | seg[i + 1]) : seg[i]; i += pq ? 2 : 1; }
        qt_ok[tq] = true;
      }
    } else if (m == 0xC4) {                                                      // DHT
      size_t i = 0;
      while (i + 17 <= slen) {
        const int tc = seg[i] >> 4, th = seg[i] & 15;
        if (tc > 1 || th > 3) return JN_ERR_INVALID;
        int total = 0;
        for (int k = 0; k < 16; k++) total += seg[i + 1 + k];
        if (total > 256 || i + 17 + total > slen) return JN_ERR_INVALID;
        if (!(tc ? ac[th] : dc[th]).build(seg + i + 1, seg + i + 17, total)) return JN_ERR_INVALID;
        i += 17 + total;
      }
      if (i != slen) return JN_ERR_INVALID;
    } else if (m == 0xC0 || m == 0xC1) {                                         // SOF0 / SOF1 (Huffman, sequential)
      if (slen < 6 || seg[0] != 8) return JN_ERR_UNSUPPORTED;
      out.height = (seg[1] << 8) | seg[2]; out.width = (seg[3] << 8) | seg[4];
      ncomp = seg[5];
      if ((ncomp != 1 && ncomp != 3) || slen < (size_t)(6 + 3 * ncomp) || out.width < 1 || out.height < 1) return JN_ERR_UNSUPPORTED;
      if (out.width > jnav::kJpegMaxDim || out.height > jnav::kJpegMaxDim) return JN_ERR_UNSUPPORTED;  // the same limit as jn_elas_create; bounds `coef`
      for (int c = 0; c < ncomp; c++) { comp[c].id = seg[6 + 3 * c]; comp[c].h = seg[7 + 3 * c] >> 4; comp[c].v = seg[7 + 3 * c] & 15; comp[c].tq = seg[8 + 3 * c]; }
      have_frame = true;
    } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8)) {
      return JN_ERR_UNSUPPORTED;                                                 // progressive, lossless, arithmetic, hierarchical
    } else if (m == 0xDD) {                                                      // DRI
      if (slen < 2) return JN_ERR_INVALID;
      restart_interval = (seg[0] << 8) | seg[1];
    } else if (m == 0xDA) {                                                      // SOS: the one scan of a baseline file
      if (!have_frame || slen < 1 || seg[0] != ncomp || slen < (size_t)(1 + 2 * ncomp + 3)) return JN_ERR_UNSUPPORTED;
      for (int c = 0; c < ncomp; c++) {
        if (seg[1 + 2 * c] != comp[c].id) return JN_ERR_UNSUPPORTED;
        comp[c].td = seg[2 + 2 * c] >> 4; comp[c].ta = seg[2 + 2 * c] & 15;
        if (comp[c].td > 3 || comp[c].ta > 3 || !dc[comp[c].td].present || !ac[comp[c].ta].present) return JN_ERR_INVALID;
      }
      if (comp[0].tq > 3 || !qt_ok[comp[0].tq]) return JN_ERR_INVALID;
      const int hmax = ncomp == 1 ? 1 : comp[0].h, vmax = ncomp == 1 ? 1 : comp[0].v;
      if (hmax < 1 || hmax > 2 || vmax < 1 || vmax > 2) return JN_ERR_UNSUPPORTED;
      if (ncomp == 3 && (comp[1].h > hmax || comp[1].v > vmax || comp[2].h > hmax || comp[2].v > vmax || comp[1].h < 1 || comp[2].h < 1 || comp[1].v < 1 || comp[2].v < 1))
        return JN_ERR_UNSUPPORTED;                                               // luminance must carry the largest factors
      const int hy = ncomp == 1 ? 1 : comp[0].h, vy = ncomp == 1 ? 1 : comp[0].v;
      const int mcux = (out.width + 8 * hmax - 1) / (8 * hmax), mcuy = (out.height + 8 * vmax - 1) / (8 * vmax);
      out.bw = mcux * hy; out.bh = mcuy * vy;
      memcpy(out.quant, qt[comp[0].tq], sizeof(out.quant));
      coef.assign((size_t)out.bw * out.bh * 64, 0);
      BitReader br; br.p = data + pos + len; br.end = data + n;
      int pred[3] = {0, 0, 0};
      int until_restart = restart_interval, next_rst = 0;
      for (int my = 0; my < mcuy; my++)
        for (int mx = 0; mx < mcux; mx++) {
          if (restart_interval && until_restart == 0) {                          // T.81 F.2.2.4: byte-align, RSTm, reset predictors
            const uint8_t* q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
            if (q + 1 >= br.end || q[1] != 0xD0 + next_rst) return JN_ERR_INVALID;
            br.p = q + 2; br.restart();
            next_rst = (next_rst + 1) & 7; until_restart = restart_interval;
            pred[0] = pred[1] = pred[2] = 0;
          }
          for (int c = 0; c < ncomp; c++) {
            const int ch = ncomp == 1 ? 1 : comp[c].h, cv = ncomp == 1 ? 1 : comp[c].v;
            const Huff& hd = dc[comp[c].td]; const Huff& ha = ac[comp[c].ta];
            for (int by = 0; by < cv; by++)
              for (int bx = 0; bx < ch; bx++) {
                int16_t* blk = c == 0 ? &coef[((size_t)(my * vy + by) * out.bw + (mx * hy + bx)) * 64] : nullptr;
                br.fill();
                int s = decode_symbol(br, hd);
                if (s < 0 || s > 11) return JN_ERR_INVALID;
                if (s) { pred[c] += extend(br.peek(s), s); br.drop(s); }
                if (blk) blk[0] = (int16_t)pred[c];
                for (int k = 1; k < 64;) {
                  br.fill();                                                     // >= 33 bits: the symbol and its extra bits
                  const int rs = decode_symbol(br, ha);
                  if (rs < 0) return JN_ERR_INVALID;
                  const int r = rs >> 4; s = rs & 15;
                  if (s) {
                    k += r;
                    if (k > 63) return JN_ERR_INVALID;
                    const int v = extend(br.peek(s), s); br.drop(s);
                    if (blk) blk[kZigzag[k]] = (int16_t)v;
                    k++;
                  } else if (r == 15) k += 16;                                   // ZRL
                  else break;                                                    // EOB
                }
              }
          }
          if (restart_interval) until_restart--;
        }
      return JN_OK;
    }
    pos += len;
  }
  return JN_ERR_INVALID;                                                         // no scan found
}


extern "C" {

jn_status jn_jpeg_info(const uint8_t* jpeg, int64_t nbytes, int32_t* width, int32_t* height) {
  if (!jpeg || nbytes < 4 || !width || !height) return JN_ERR_INVALID;
  if (jpeg[0] != 0xFF || jpeg[1] != 0xD8) return JN_ERR_INVALID;
  size_t pos = 2; const size_t n = (size_t)nbytes;
  while (pos + 4 <= n) {
    if (jpeg[pos] != 0xFF) { pos++; continue; }
    const int m = jpeg[pos + 1];
    if (m == 0xFF) { pos++; continue; }
    pos += 2;
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    const size_t len = ((size_t)jpeg[pos] << 8) | jpeg[pos + 1];
    if (len < 2 || pos + len > n) return JN_ERR_INVALID;
    if (m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      if (len < 7) return JN_ERR_INVALID;
      *height = (jpeg[pos + 3] << 8) | jpeg[pos + 4]; *width = (jpeg[pos + 5] << 8) | jpeg[pos + 6];
      return (m == 0xC0 || m == 0xC1) && jpeg[pos + 2] == 8 ? JN_OK : JN_ERR_UNSUPPORTED;
    }
    if (m == 0xDA) break;
    pos += len;
  }
  return JN_ERR_INVALID;
}

int64_t jn_host_jpeg_coefficients(const uint8_t* jpeg, int64_t nbytes, int16_t* coef, int64_t coef_capacity, uint16_t quant[64],
                                  int32_t* width, int32_t* height, int32_t* blocks_w, int32_t* blocks_h) {
  if (!jpeg || nbytes < 4 || !quant || !width || !height || !blocks_w || !blocks_h) return -JN_ERR_INVALID;
  jnav::JpegFrame d;
  std::vector<int16_t> c;
  jn_status st;
  try { st = jnav::jpeg_parse_and_decode(jpeg, (size_t)nbytes, d, c); } catch (const std::bad_alloc&) { return -(int64_t)JN_ERR_INTERNAL; }
  if (st != JN_OK) return -(int64_t)st;
  *width = d.width; *height = d.height; *blocks_w = d.bw; *blocks_h = d.bh;
  memcpy(quant, d.quant, sizeof(d.quant));
  if (coef) {
    if ((int64_t)c.size() > coef_capacity) return -JN_ERR_INVALID;
    memcpy(coef, c.data(), c.size() * sizeof(int16_t));
  }
  return (int64_t)c.size();
}

}  // extern "C"
