// jpeg.hip — the GPU half of cv::imdecode(data, CV_LOAD_IMAGE_GRAYSCALE) (point_cloud.cpp:436, :478): dequantisation,
// inverse DCT and range limit of the luminance blocks; the description of the whole decoder, its split and what it supports
// is at the top of jpeg_host.cpp, which holds the serial entropy decoder.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
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

// One eye: frame size against the caller's buffer, entropy decoding on the calling thread (the serial part), then upload +
// inverse DCT.  `sc` is the calling thread's grow-only device scratch for this eye.
// Coefficients travel host -> device from PINNED memory: from a pageable vector hipMemcpyAsync is staged and synchronous, and the pair
// call's two uploads + inverse DCTs would not be "queued together, waited for once".  The decoder's own vector is page-locked in place
// (hipHostRegister, redone only when the vector has moved or grown) — no second copy of the coefficients.  The scratch belongs to a thread
// (thread_local) or to an EyeHelper; its destructor releases the device buffer and the registration when that thread ends.
struct JpegScratch {
  int16_t* p = nullptr; size_t cap = 0; int dev = -1;
  void* reg_ptr = nullptr; size_t reg_bytes = 0;
  std::vector<int16_t> coef; jnav::JpegFrame frame;
  void unregister() { if (reg_ptr) { hipHostUnregister(reg_ptr); reg_ptr = nullptr; reg_bytes = 0; } }
  ~JpegScratch() {
    unregister();
    if (p) { hipSetDevice(dev); hipFree(p); }
  }
};

static jn_status jpeg_entropy(const uint8_t* jpeg, int64_t nbytes, int32_t out_pitch, int32_t out_rows, int32_t* width, int32_t* height, JpegScratch& sc) {
  if (!jpeg || nbytes < 4 || !width || !height) return JN_ERR_INVALID;
  // frame size against the caller's buffer BEFORE any entropy decoding or allocation
  jn_status st = jn_jpeg_info(jpeg, nbytes, width, height);
  if (st != JN_OK) return st;
  if (*width > jnav::kJpegMaxDim || *height > jnav::kJpegMaxDim) return JN_ERR_UNSUPPORTED;
  if (out_pitch < *width || out_rows < *height) return JN_ERR_INVALID;
  // room for every luminance block the frame can have (MCUs of up to 16x16 pixels), reserved BEFORE decoding: the decoder then never moves
  // the vector, so a page-locked (registered) vector is only ever released here, after its registration
  const size_t worst = (size_t)((*width + 15) / 16 * 2) * (size_t)((*height + 15) / 16 * 2) * 64;
  try {
    if (sc.coef.capacity() < worst) { sc.unregister(); sc.coef.reserve(worst); }
    st = jnav::jpeg_parse_and_decode(jpeg, (size_t)nbytes, sc.frame, sc.coef);
  } catch (const std::bad_alloc&) { return JN_ERR_INTERNAL; }
  if (st != JN_OK) return st;
  if (sc.frame.width != *width || sc.frame.height != *height) return JN_ERR_INVALID;      // two frame headers that disagree
  return JN_OK;
}
static jn_status jpeg_idct_launch(int32_t device, JpegScratch& sc, uint8_t* dOut, int32_t out_pitch) {
  const size_t need = sc.coef.size() * sizeof(int16_t);
  if (sc.dev != device || sc.cap < need) {                  // grow-only device buffer per calling thread, eye and device
    if (sc.p) { hipSetDevice(sc.dev); hipFree(sc.p); hipSetDevice(device); sc.p = nullptr; sc.cap = 0; }
    JPG_TRY(hipMalloc(reinterpret_cast<void**>(&sc.p), need));
    sc.cap = need; sc.dev = device;
  }
  const size_t have = sc.coef.capacity() * sizeof(int16_t);
  if (sc.reg_ptr != sc.coef.data() || sc.reg_bytes < need) {  // the vector moved or grew since it was page-locked
    sc.unregister();
    if (hipHostRegister(sc.coef.data(), have, hipHostRegisterDefault) == hipSuccess) { sc.reg_ptr = sc.coef.data(); sc.reg_bytes = have; }
    else (void)hipGetLastError();                              // not fatal: the copy below is then staged by the runtime
  }
  JPG_TRY(hipMemcpyAsync(sc.p, sc.coef.data(), need, hipMemcpyHostToDevice, nullptr));
  QuantTable qt;
  memcpy(qt.q, sc.frame.quant, sizeof(qt.q));
  const int blocks = sc.frame.bw * sc.frame.bh;
  hipLaunchKernelGGL(k_jpeg_idct_gray, dim3((blocks + 31) / 32), dim3(256), 0, nullptr, sc.p, qt, sc.frame.bw, sc.frame.bh, sc.frame.width, sc.frame.height, dOut, out_pitch);
  return JN_OK;
}

extern "C" {

jn_status jn_jpeg_decode_gray(int32_t device, const uint8_t* jpeg, int64_t nbytes, uint8_t* dOut, int32_t out_pitch, int32_t out_rows,
                              int32_t* width, int32_t* height) {
  if (!dOut) return JN_ERR_INVALID;
  static thread_local JpegScratch sc;
  jn_status st = jpeg_entropy(jpeg, nbytes, out_pitch, out_rows, width, height, sc);
  if (st != JN_OK) return st;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return JN_ERR_NO_DEVICE;
  JPG_TRY(hipSetDevice(device));
  if ((st = jpeg_idct_launch(device, sc, dOut, out_pitch)) != JN_OK) return st;
  JPG_TRY(hipStreamSynchronize(nullptr));
  JPG_TRY(hipGetLastError());
  return JN_OK;
}

// Both eyes of a stereo frame (point_cloud.cpp:436 and :478 decode them in two callbacks): the two entropy decodes — the
// serial, host-bound 2 x 0.4 ms of a 640x360 frame pair — run on two threads, then both inverse DCTs are queued and waited
// for once.  A long-lived helper thread per calling thread takes the right eye (no thread is created per frame).
namespace {
struct EyeHelper {
  std::thread th; std::mutex m; std::condition_variable cv;
  bool has = false, done = false, quit = false;
  const uint8_t* jpeg = nullptr; int64_t nbytes = 0; int32_t pitch = 0, rows = 0, w = 0, h = 0; jn_status st = JN_OK;
  JpegScratch sc;
  EyeHelper() { th = std::thread([this] { run(); }); }
  ~EyeHelper() { { std::lock_guard<std::mutex> l(m); quit = true; } cv.notify_all(); th.join(); }   // sc frees its own buffers
  void run() {
    std::unique_lock<std::mutex> l(m);
    for (;;) {
      cv.wait(l, [&] { return has || quit; });
      if (quit) return;
      has = false;
      l.unlock();
      const jn_status r = jpeg_entropy(jpeg, nbytes, pitch, rows, &w, &h, sc);
      l.lock();
      st = r; done = true;
      cv.notify_all();
    }
  }
};
}  // namespace

jn_status jn_jpeg_decode_gray_pair(int32_t device, const uint8_t* jpegL, int64_t nbytesL, const uint8_t* jpegR, int64_t nbytesR, uint8_t* dOutL,
                                   uint8_t* dOutR, int32_t out_pitch, int32_t out_rows, int32_t* width, int32_t* height) {
  if (!dOutL || !dOutR || !width || !height) return JN_ERR_INVALID;
  static thread_local JpegScratch scL;
  static thread_local std::unique_ptr<EyeHelper> helper;
  if (!helper) helper.reset(new EyeHelper());
  EyeHelper& hp = *helper;
  {
    std::lock_guard<std::mutex> l(hp.m);
    hp.jpeg = jpegR; hp.nbytes = nbytesR; hp.pitch = out_pitch; hp.rows = out_rows; hp.has = true; hp.done = false;
  }
  hp.cv.notify_all();
  const jn_status stL = jpeg_entropy(jpegL, nbytesL, out_pitch, out_rows, width, height, scL);
  jn_status stR;
  {
    std::unique_lock<std::mutex> l(hp.m);
    hp.cv.wait(l, [&] { return hp.done; });
    stR = hp.st;
  }
  if (stL != JN_OK) return stL;
  if (stR != JN_OK) return stR;
  if (hp.w != *width || hp.h != *height) return JN_ERR_INVALID;                 // the two eyes must be the same size
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return JN_ERR_NO_DEVICE;
  JPG_TRY(hipSetDevice(device));
  jn_status st;
  if ((st = jpeg_idct_launch(device, scL, dOutL, out_pitch)) != JN_OK) return st;
  if ((st = jpeg_idct_launch(device, hp.sc, dOutR, out_pitch)) != JN_OK) return st;
  JPG_TRY(hipStreamSynchronize(nullptr));
  JPG_TRY(hipGetLastError());
  return JN_OK;
}

}  // extern "C"
