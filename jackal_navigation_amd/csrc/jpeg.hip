// jpeg.hip — the GPU half of cv::imdecode(data, CV_LOAD_IMAGE_GRAYSCALE) (point_cloud.cpp:436, :478): dequantisation,
// inverse DCT and range limit of the luminance blocks; the description of the whole decoder, its split and what it supports
// is at the top of jpeg_host.cpp, which holds the serial entropy decoder.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>
#include "../../include/jn_stereo.h"
#include "jpeg_host.h"

namespace {

// ---- dequantisation + 8x8 inverse DCT ("slow integer" form) + range limit: 8 lanes per block ----
#define JDESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
__device__ __forceinline__ void idct_1d(const int in[8], int out[8], int shift) {
  // even part
  int z2 = in[2], z3 = in[6];
  int z1 = (z2 + z3) * 4433;
  int tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
  int tmp0 = (in[0] + in[4]) * 8192, tmp1 = (in[0] - in[4]) * 8192;            // << CONST_BITS
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  // odd part
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; int z4 = tmp1 + tmp3;
  const int z5 = (z3 + z4) * 9633;
  tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
  z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  out[0] = JDESCALE(tmp10 + tmp3, shift); out[7] = JDESCALE(tmp10 - tmp3, shift);
  out[1] = JDESCALE(tmp11 + tmp2, shift); out[6] = JDESCALE(tmp11 - tmp2, shift);
  out[2] = JDESCALE(tmp12 + tmp1, shift); out[5] = JDESCALE(tmp12 - tmp1, shift);
  out[3] = JDESCALE(tmp13 + tmp0, shift); out[4] = JDESCALE(tmp13 - tmp0, shift);
}
__device__ __forceinline__ uint8_t range_limit(int x) {      // IJG sample range table, indexed modulo 1024 around +128
  const int i = x & 1023;
  return (uint8_t)(i < 128 ? 128 + i : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
}
struct QuantTable { uint16_t q[64]; };
__global__ void __launch_bounds__(256) k_jpeg_idct_gray(const int16_t* __restrict__ coef, QuantTable qt, int bw, int bh, int W, int H,
                                                        uint8_t* __restrict__ out, int pitch) {
  __shared__ int ws[32][64 + 8];
  const int lb = threadIdx.x >> 3, l = threadIdx.x & 7;
  const int block = blockIdx.x * 32 + lb;
  const bool in = block < bw * bh;
  int col[8], tmp[8];
  if (in) {
    const int16_t* c = coef + (size_t)block * 64;
#pragma unroll
    for (int r = 0; r < 8; r++) col[r] = (int)c[8 * r + l] * (int)qt.q[8 * r + l];     // column l, dequantised
    idct_1d(col, tmp, 13 - 2);                                                          // pass 1: CONST_BITS - PASS1_BITS
#pragma unroll
    for (int r = 0; r < 8; r++) ws[lb][8 * r + l] = tmp[r];
  }
  __syncthreads();
  if (!in) return;
#pragma unroll
  for (int k = 0; k < 8; k++) col[k] = ws[lb][8 * l + k];                               // row l of the workspace
  idct_1d(col, tmp, 13 + 2 + 3);                                                        // pass 2: CONST_BITS + PASS1_BITS + 3
  const int bx = block % bw, by = block / bw;
  const int y = by * 8 + l;
  if (y >= H) return;
#pragma unroll
  for (int k = 0; k < 8; k++) { const int x = bx * 8 + k; if (x < W) out[(size_t)y * pitch + x] = range_limit(tmp[k]); }
}

}  // namespace

#define JPG_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      fprintf(stderr, "libjn_stereo: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return JN_ERR_NO_DEVICE;                                                              \
    }                                                                                       \
  } while (0)

extern "C" {

jn_status jn_jpeg_decode_gray(int32_t device, const uint8_t* jpeg, int64_t nbytes, uint8_t* dOut, int32_t out_pitch, int32_t out_rows,
                              int32_t* width, int32_t* height) {
  if (!jpeg || nbytes < 4 || !dOut || !width || !height) return JN_ERR_INVALID;
  // frame size against the caller's buffer BEFORE any entropy decoding or allocation
  jn_status st = jn_jpeg_info(jpeg, nbytes, width, height);
  if (st != JN_OK) return st;
  if (*width > jnav::kJpegMaxDim || *height > jnav::kJpegMaxDim) return JN_ERR_UNSUPPORTED;
  if (out_pitch < *width || out_rows < *height) return JN_ERR_INVALID;
  jnav::JpegFrame d;
  static thread_local std::vector<int16_t> coef;
  try { st = jnav::jpeg_parse_and_decode(jpeg, (size_t)nbytes, d, coef); } catch (const std::bad_alloc&) { return JN_ERR_INTERNAL; }
  if (st != JN_OK) return st;
  if (d.width != *width || d.height != *height) return JN_ERR_INVALID;          // two frame headers that disagree
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return JN_ERR_NO_DEVICE;
  JPG_TRY(hipSetDevice(device));
  // grow-only device buffer per calling thread and device (no allocation per frame)
  struct Scratch { int16_t* p = nullptr; size_t cap = 0; int dev = -1; };
  static thread_local Scratch sc;
  const size_t need = coef.size() * sizeof(int16_t);
  if (sc.dev != device || sc.cap < need) {
    if (sc.p) { hipSetDevice(sc.dev); hipFree(sc.p); hipSetDevice(device); sc.p = nullptr; sc.cap = 0; }
    JPG_TRY(hipMalloc(reinterpret_cast<void**>(&sc.p), need));
    sc.cap = need; sc.dev = device;
  }
  JPG_TRY(hipMemcpy(sc.p, coef.data(), need, hipMemcpyHostToDevice));
  QuantTable qt;
  memcpy(qt.q, d.quant, sizeof(qt.q));
  const int blocks = d.bw * d.bh;
  hipLaunchKernelGGL(k_jpeg_idct_gray, dim3((blocks + 31) / 32), dim3(256), 0, nullptr, sc.p, qt, d.bw, d.bh, d.width, d.height, dOut, out_pitch);
  JPG_TRY(hipStreamSynchronize(nullptr));
  JPG_TRY(hipGetLastError());
  return JN_OK;
}

}  // extern "C"
