// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the stereo hot path.  Product code.
//
// Every kernel states the reference lines it reproduces.  Integer results are bit-exact with the
// reference CPU path; float results are bit-exact too because the same IEEE operations are issued
// in the same order (built with -ffp-contract=off; products/sums that must not fuse use
// __fmul_rn/__fadd_rn explicitly).
#include "kernels.h"
#include "hooks.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace jnav {

#define DEV static __device__ __forceinline__

DEV int sad16(const uint4& a, const uint4& b) {
  // v_sad_u8: 4 byte-wise absolute differences + accumulate, per instruction
  unsigned s = __builtin_amdgcn_sad_u8(a.x, b.x, 0u);
  s = __builtin_amdgcn_sad_u8(a.y, b.y, s);
  s = __builtin_amdgcn_sad_u8(a.z, b.z, s);
  s = __builtin_amdgcn_sad_u8(a.w, b.w, s);
  return (int)s;
}
DEV unsigned sad16_acc(const uint4& a, const uint4& b, unsigned s) {   // the same, continuing a running sum
  s = __builtin_amdgcn_sad_u8(a.x, b.x, s);
  s = __builtin_amdgcn_sad_u8(a.y, b.y, s);
  s = __builtin_amdgcn_sad_u8(a.z, b.z, s);
  return __builtin_amdgcn_sad_u8(a.w, b.w, s);
}
// v_sad_hi_u8 adds the 4-byte SAD shifted left by 16 to its accumulator: a chain of them on `d` builds the key (cost << 16) | d
DEV unsigned sadhi16(const uint4& a, const uint4& b, unsigned acc) {      // acc + (SAD16(a, b) << 16)
  acc = __builtin_amdgcn_sad_hi_u8(a.x, b.x, acc);
  acc = __builtin_amdgcn_sad_hi_u8(a.y, b.y, acc);
  acc = __builtin_amdgcn_sad_hi_u8(a.z, b.z, acc);
  return __builtin_amdgcn_sad_hi_u8(a.w, b.w, acc);
}
DEV unsigned med3u(unsigned a, unsigned b, unsigned c) { return max(min(a, b), min(max(a, b), c)); }   // v_med3_u32
DEV int texture16(const uint4& a) {   // sum |byte - 128| (elas.cpp:301-305, :715-719)
  const unsigned k = 0x80808080u;
  unsigned s = __builtin_amdgcn_sad_u8(a.x, k, 0u);
  s = __builtin_amdgcn_sad_u8(a.y, k, s);
  s = __builtin_amdgcn_sad_u8(a.z, k, s);
  s = __builtin_amdgcn_sad_u8(a.w, k, s);
  return (int)s;
}
DEV int sat_u8(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }

// ------------------------------------------------------------------------------------------------
// Sobel + descriptor, fused (filter.cpp:372-416, :227-267, :176-222; descriptor.cpp:84-111).
// One workgroup per 64x16 tile of descriptors: the (64+8)x(16+6) image patch is staged in LDS,
//   S = I[v-1]+2I[v]+I[v+1], T = I[v-1]-I[v+1]                       (int16 column pass)
//   du = sat(((S[u-1]-S[u+1])>>2)+128), dv = sat(((T[u-1]+2T[u]+T[u+1])>>2)+128)
// are formed in LDS for the (64+4)x(16+4) / (64+2)x(16+2) pixels the tile's descriptors tap, and each
// descriptor (12 du taps on a 5-row diamond + 4 dv taps) leaves as one 16-byte store.
// Everything moves four pixels at a time: LDS rows are dword arrays, a thread reads two aligned dwords
// (8 neighbouring bytes) per row and v_perm_b32 picks the bytes; the Sobel sums run as two 16-bit halves per
// register (plain adds where no carry can cross, v_pk_* where signs matter).  Per descriptor that is ~11
// instructions and 4 LDS reads instead of ~50 and 16.
// du/dv are only ever tapped at rows 1..H-2, columns 1..W-2 (descriptors exist for u in [3,W-4], v in [3,H-4]),
// so whatever the patch border produces elsewhere is never looked at.  Pixels outside that range get zeros
// (uninitialised in the reference).
enum { kDescTW = 64, kDescTH = 16, kDescIW = (kDescTW + 8) / 4, kDescDW = (kDescTW + 4) / 4,   // dwords per LDS row
       kDescTiles = 2 };                                                                        // tiles per workgroup
typedef short pk16 __attribute__((ext_vector_type(2)));
DEV pk16 as_pk(uint32_t x) { union { uint32_t u; pk16 p; } c; c.u = x; return c.p; }
DEV uint32_t as_u32(pk16 p) { union { uint32_t u; pk16 p; } c; c.p = p; return c.u; }
// ((x >> 2) + 128) saturated to a byte, on both halves
DEV pk16 sobel_norm(pk16 x) {
  const pk16 lo = {0, 0}, hi = {255, 255}, off = {128, 128};
  return __builtin_elementwise_min(__builtin_elementwise_max((x >> 2) + off, lo), hi);
}
__global__ void __launch_bounds__(256) k_descriptor_fused(DevParams dp, const uint8_t* __restrict__ I1, const uint8_t* __restrict__ I2,
                                                          int in_pitch, long long in_stride, int n, uint4* __restrict__ desc) {
  __shared__ uint32_t s_I[kDescTH + 6][kDescIW];           // image rows v0-3.., columns u0-4.. (4 bytes per word)
  __shared__ uint32_t s_du[kDescTH + 4][kDescDW];          // du rows v0-2.., columns u0-2..
  __shared__ uint32_t s_dv[kDescTH + 2][kDescDW];          // dv rows v0-1.., columns u0-1..
  __shared__ uint4 s_out[kDescTW * kDescTH];              // finished descriptors, swizzled (see below)
  const int img = blockIdx.z, W = dp.W, H = dp.H;
  const uint8_t* I = (img < n ? I1 + (long long)img * in_stride : I2 + (long long)(img - n) * in_stride);
  const int v0 = blockIdx.y * kDescTH, tid = threadIdx.x;
  const bool aligned = ((reinterpret_cast<uintptr_t>(I) | (uintptr_t)in_pitch) & 3) == 0;
  // A workgroup walks kDescTiles tiles of its tile row.  The image words of the next tile are fetched into
  // registers while the current tile is computed: with one tile per workgroup the chain load -> Sobel -> assemble
  // -> transpose -> store (three barriers) is latency bound even at 7 workgroups per CU (compute alone 128 us,
  // stores alone 138 us, one after the other 235 us); 2 tiles per workgroup: 184 us, 4: 189 us, 10: 217 us.
  constexpr int kWords = ((kDescTH + 6) * kDescIW + 255) / 256;            // patch words per thread
  auto fetch = [&](int u0, uint32_t (&w)[kWords]) {
#pragma unroll
    for (int q = 0; q < kWords; q++) {
      const int i = tid + 256 * q;
      const int r = i / kDescIW, k = i - r * kDescIW;
      const int v = v0 - 3 + r, u = u0 - 4 + 4 * k;
      uint32_t x = 0;
      if (i < (kDescTH + 6) * kDescIW && v >= 0 && v < H) {
        const uint8_t* src = I + (size_t)v * in_pitch + u;
        if (aligned && u >= 0 && u + 3 < W) x = *reinterpret_cast<const uint32_t*>(src);
        else
#pragma unroll
          for (int b = 0; b < 4; b++) if (u + b >= 0 && u + b < W) x |= (uint32_t)src[b] << (8 * b);
      }
      w[q] = x;
    }
  };
  const int tile0 = blockIdx.x * kDescTiles, tiles_x = (W + kDescTW - 1) / kDescTW;
  const int tile1 = min(tile0 + kDescTiles, tiles_x);
  uint32_t patch[kWords];
  fetch(tile0 * kDescTW, patch);
  for (int tile = tile0; tile < tile1; tile++) {
  const int u0 = tile * kDescTW;
#pragma unroll
  for (int q = 0; q < kWords; q++) {
    const int i = tid + 256 * q;
    if (i < (kDescTH + 6) * kDescIW) s_I[i / kDescIW][i % kDescIW] = patch[q];
  }
  if (tile + 1 < tile1) fetch(u0 + kDescTW, patch);                        // in flight during this tile's work
  __syncthreads();
  // bytes b0..b7 = image columns u0-4+4k .. +7 of one patch row; pair(j) = (b_j, b_j+1) as two 16-bit halves
#define JN_PAIR(hi, lo, j) __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)(j) | ((uint32_t)((j) + 1) << 16))
  for (int i = tid; i < (kDescTH + 4) * kDescDW; i += 256) {               // four du per item: columns u0-2+4k ..
    const int r = i / kDescDW, k = i - r * kDescDW;
    uint32_t S[3];                                                         // S at bytes (1,2), (3,4), (5,6)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const uint32_t a = JN_PAIR(s_I[r][k + 1], s_I[r][k], 2 * j + 1), b = JN_PAIR(s_I[r + 1][k + 1], s_I[r + 1][k], 2 * j + 1),
                     c = JN_PAIR(s_I[r + 2][k + 1], s_I[r + 2][k], 2 * j + 1);
      S[j] = a + 2 * b + c;                                                // <= 1020 per half: no carry between halves
    }
    const pk16 d01 = sobel_norm(as_pk(S[0]) - as_pk(S[1])), d23 = sobel_norm(as_pk(S[1]) - as_pk(S[2]));
    s_du[r][k] = __builtin_amdgcn_perm(as_u32(d23), as_u32(d01), 0x06040200u);
  }
  for (int i = tid; i < (kDescTH + 2) * kDescDW; i += 256) {               // four dv per item: columns u0-1+4k ..
    const int r = i / kDescDW, k = i - r * kDescDW;
    pk16 T[3];                                                             // T at bytes (2,3), (4,5), (6,7)
#pragma unroll
    for (int j = 0; j < 3; j++)
      T[j] = as_pk(JN_PAIR(s_I[r + 1][k + 1], s_I[r + 1][k], 2 * j + 2)) - as_pk(JN_PAIR(s_I[r + 3][k + 1], s_I[r + 3][k], 2 * j + 2));
    const pk16 m01 = as_pk(__builtin_amdgcn_perm(as_u32(T[1]), as_u32(T[0]), 0x05040302u));   // (T3, T4)
    const pk16 m23 = as_pk(__builtin_amdgcn_perm(as_u32(T[2]), as_u32(T[1]), 0x05040302u));   // (T5, T6)
    const pk16 d01 = sobel_norm(T[0] + m01 + m01 + T[1]), d23 = sobel_norm(T[1] + m23 + m23 + T[2]);
    s_dv[r][k] = __builtin_amdgcn_perm(as_u32(d23), as_u32(d01), 0x06040200u);
  }
#undef JN_PAIR
  __syncthreads();
  // A thread assembles four neighbouring descriptors (64 bytes), but a wave must store 1 KB of consecutive
  // bytes per instruction to reach HBM write speed (measured: 64-byte-strided 16-byte stores run at 3.1 TB/s,
  // contiguous ones at 6.8).  So the tile's descriptors take one trip through LDS; slot = pixel ^ ((pixel >> 4) & 3)
  // keeps both the 4-pixel-strided writes and the linear reads free of bank conflicts.
  const int xq = tid & 15, y = tid >> 4, v = v0 + y, ub = u0 + 4 * xq;
  uint32_t al[5], ah[5], bl[3], bh[3];
#pragma unroll
  for (int k = 0; k < 5; k++) { al[k] = s_du[y + k][xq]; ah[k] = s_du[y + k][xq + 1]; }    // du(v-2+k, ub-2 .. ub+5)
#pragma unroll
  for (int k = 0; k < 3; k++) { bl[k] = s_dv[y + k][xq]; bh[k] = s_dv[y + k][xq + 1]; }    // dv(v-1+k, ub-1 .. ub+6)
  const bool row_in = v >= 3 && v <= H - 4;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int u = ub + i;
    uint4 d = make_uint4(0, 0, 0, 0);
    if (row_in && u >= 3 && u <= W - 4) {
      // byte j of (ah[k]:al[k]) = du(v-2+k, u-2 + j-i); byte j of (bh[k]:bl[k]) = dv(v-1+k, u-1 + j-i)
      const uint32_t c = (uint32_t)i, z = 0x0cu;                           // selector 0x0c = constant zero byte
      d.x = __builtin_amdgcn_perm(ah[0], al[0], (z << 24) | (z << 16) | (z << 8) | (c + 2)) |
            __builtin_amdgcn_perm(ah[1], al[1], ((c + 4) << 24) | ((c + 2) << 16) | (c << 8) | z);
      d.y = __builtin_amdgcn_perm(ah[2], al[2], ((c + 3) << 24) | ((c + 2) << 16) | ((c + 2) << 8) | (c + 1));
      d.z = __builtin_amdgcn_perm(ah[3], al[3], (z << 24) | ((c + 4) << 16) | ((c + 2) << 8) | c) |
            __builtin_amdgcn_perm(ah[4], al[4], ((c + 2) << 24) | (z << 16) | (z << 8) | z);
      d.w = __builtin_amdgcn_perm(bh[0], bl[0], (z << 24) | (z << 16) | (z << 8) | (c + 1)) |
            __builtin_amdgcn_perm(bh[1], bl[1], (z << 24) | ((c + 2) << 16) | (c << 8) | z) |
            __builtin_amdgcn_perm(bh[2], bl[2], ((c + 1) << 24) | (z << 16) | (z << 8) | z);
    }
    const int p = 4 * tid + i;                                             // = y * 64 + 4 * xq + i
    s_out[p ^ ((p >> 4) & 3)] = d;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int p = 256 * k + tid, vv = v0 + (p >> 6), uu = u0 + (p & 63);
    if (vv < H && uu < W) desc[((size_t)img * H + vv) * W + uu] = s_out[p ^ ((p >> 4) & 3)];
  }
  }   // next tile: the two barriers before s_out is written again order these reads before those writes
}

// ------------------------------------------------------------------------------------------------
// The plane data flow (round 5).  A descriptor is 16 bytes PICKED from the two Sobel responses (descriptor.cpp:84-111: twelve taps of
// du on a five-row diamond, four of dv), so materialising it costs 32 bytes per pixel and pair in HBM to write and as much again for
// every kernel that reads it back — the path's largest traffic term.  Instead the two responses are stored as byte planes (2 bytes per
// pixel and image) and the kernels that match descriptors assemble the ones they need, where they stage them in LDS anyway.
//
// Plane layout, per image [du | dv][H][Wp] bytes (Wp a multiple of 64, >= W + 8), images L0..L(n-1), R0..R(n-1):
//   P_du[v][x] = du(v, x - 2),   P_dv[v][x] = dv(v, x - 1)
// so that the descriptors of the four columns 4g .. 4g+3 of row v tap bytes 4g .. 4g+7 of du rows v-2 .. v+2 and of dv rows
// v-1 .. v+1: one aligned 8-byte load per plane row, eight per four descriptors.  Descriptors exist for u in [3, W-4], v in [3, H-4]
// (descriptor.cpp:84-88); outside they are ZERO (uninitialised in the reference) — the assembling kernels apply that rule,
// the plane bytes those taps would touch are never looked at.
struct PlaneRows { uint32_t al[5], ah[5], bl[3], bh[3]; };   // lo / hi dword of the 8-byte load: du rows v-2..v+2, dv rows v-1..v+1
// descriptor of column 4g + C (C a compile-time 0..3) from the rows loaded at plane byte 4g: eight v_perm_b32
template <int C>
DEV uint4 desc_from_rows(const PlaneRows& p) {
  constexpr uint32_t c = C, X = 0x0cu;                       // selector 0x0c = constant zero byte; 0-3 pick from the second operand, 4-7 from the first
  uint4 d;
  // d.x = du(v-2,u) | du(v-1,u-2) | du(v-1,u) | du(v-1,u+2): row 0 byte c+2; row 1 bytes c, c+2, c+4
  uint32_t t = __builtin_amdgcn_perm(p.ah[1], p.al[1], ((c + 4) << 24) | ((c + 2) << 16) | (c << 8) | X);
  d.x = __builtin_amdgcn_perm(t, c + 2 < 4 ? p.al[0] : p.ah[0], (7u << 24) | (6u << 16) | (5u << 8) | ((c + 2) & 3));
  // d.y = du(v,u-1) | du(v,u) | du(v,u) | du(v,u+1)
  d.y = __builtin_amdgcn_perm(p.ah[2], p.al[2], ((c + 3) << 24) | ((c + 2) << 16) | ((c + 2) << 8) | (c + 1));
  // d.z = du(v+1,u-2) | du(v+1,u) | du(v+1,u+2) | du(v+2,u)
  t = __builtin_amdgcn_perm(p.ah[3], p.al[3], (X << 24) | ((c + 4) << 16) | ((c + 2) << 8) | c);
  d.z = __builtin_amdgcn_perm(c + 2 < 4 ? p.al[4] : p.ah[4], t, ((4 + ((c + 2) & 3)) << 24) | (2u << 16) | (1u << 8) | 0u);
  // d.w = dv(v-1,u) | dv(v,u-1) | dv(v,u+1) | dv(v+1,u): dv row 0 byte c+1; row 1 bytes c, c+2; row 2 byte c+1
  t = __builtin_amdgcn_perm(p.bh[1], p.bl[1], (X << 24) | ((c + 2) << 16) | (c << 8) | X);
  t = __builtin_amdgcn_perm(t, c + 1 < 4 ? p.bl[0] : p.bh[0], (7u << 24) | (6u << 16) | (5u << 8) | ((c + 1) & 3));
  d.w = __builtin_amdgcn_perm(c + 1 < 4 ? p.bl[2] : p.bh[2], t, ((4 + ((c + 1) & 3)) << 24) | (2u << 16) | (1u << 8) | 0u);
  return d;
}
// The eight plane rows of descriptor row v at plane byte x (a multiple of 4, 0 <= x <= Wp - 8), v in [2, H-3].  The buffer resources span
// the image's du / dv plane, x rides in the vector offset, the row in the scalar one: no address arithmetic per load.
DEV PlaneRows load_plane_rows(__amdgpu_buffer_rsrc_t rdu, __amdgpu_buffer_rsrc_t rdv, int x, int v, int Wp) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  PlaneRows p;
  const int o = x + (v - 2) * Wp;                            // first du row; the rows below it ride in the scalar offset
#pragma unroll
  for (int k = 0; k < 5; k++) { const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rdu, o, k * Wp, 0); p.al[k] = t.x; p.ah[k] = t.y; }
#pragma unroll
  for (int k = 0; k < 3; k++) { const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rdv, o, (k + 1) * Wp, 0); p.bl[k] = t.x; p.bh[k] = t.y; }
  return p;
}
DEV __amdgpu_buffer_rsrc_t plane_rsrc(const uint8_t* planes, int img, int kind, int H, int Wp) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(planes + (size_t)(img * 2 + kind) * H * Wp), 0, H * Wp, 0x00020000);
}
// One descriptor at a per-lane column: the rows at the aligned byte below it, shifted down by the lane's u & 3 (v_alignbit), then
// the C = 0 pick.  Zero outside the descriptor image.  v in [2, H-3], u in [0, W-1].
DEV uint4 descriptor_at(__amdgpu_buffer_rsrc_t rdu, __amdgpu_buffer_rsrc_t rdv, int u, int v, int W, int H, int Wp) {
  PlaneRows p = load_plane_rows(rdu, rdv, u & ~3, v, Wp);
  const uint32_t sh = 8u * (uint32_t)(u & 3);
#pragma unroll
  for (int k = 0; k < 5; k++) { p.al[k] = __builtin_amdgcn_alignbit(p.ah[k], p.al[k], sh); p.ah[k] >>= sh; }   // (ah of rows 0, 2, 4 is not used: dropped by the compiler)
#pragma unroll
  for (int k = 0; k < 3; k++) p.bl[k] = __builtin_amdgcn_alignbit(p.bh[k], p.bl[k], sh);
  const uint4 d = desc_from_rows<0>(p);
  const bool in = u >= 3 && u <= W - 4 && v >= 3 && v <= H - 4;
  return in ? d : make_uint4(0, 0, 0, 0);
}

// Sobel responses as byte planes (filter.cpp:372-416, :227-267, :176-222 — the arithmetic of k_descriptor_fused above, whose tile of
// finished descriptors this replaces).  A thread owns four plane bytes of du and of dv (image columns x-4 .. x+3 of three rows feed them)
// and walks down kPlaneRows rows with a sliding window of three image rows: one 8-byte load and two 4-byte stores per row.
enum { kPlaneRows = 16 };
__global__ void __launch_bounds__(256) k_sobel_planes(DevParams dp, const uint8_t* __restrict__ I1, const uint8_t* __restrict__ I2,
                                                      int in_pitch, long long in_stride, int n, uint8_t* __restrict__ planes, int Wp) {
  const int img = blockIdx.z, W = dp.W, H = dp.H;
  const uint8_t* I = (img < n ? I1 + (long long)img * in_stride : I2 + (long long)(img - n) * in_stride);
  const int x = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));                  // plane byte; image columns x-4 .. x+3
  const int v_begin = (blockIdx.y * 4 + (threadIdx.x >> 6)) * kPlaneRows;
  if (x >= Wp || v_begin >= H) return;
  const bool aligned = ((reinterpret_cast<uintptr_t>(I) | (uintptr_t)in_pitch) & 3) == 0;
  auto word = [&](int v, int u) -> uint32_t {                                  // image bytes u .. u+3 of row v, zero outside the image
    if (v < 0 || v >= H || u + 3 < 0 || u >= W) return 0u;
    const uint8_t* src = I + (size_t)v * in_pitch + u;
    if (aligned && u >= 0 && u + 3 < W) return *reinterpret_cast<const uint32_t*>(src);
    uint32_t w = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) if (u + b >= 0 && u + b < W) w |= (uint32_t)src[b] << (8 * b);
    return w;
  };
  uint8_t* pdu = planes + (size_t)(img * 2) * H * Wp;
  uint8_t* pdv = pdu + (size_t)H * Wp;
  uint32_t lo0 = word(v_begin - 1, x - 4), hi0 = word(v_begin - 1, x), lo1 = word(v_begin, x - 4), hi1 = word(v_begin, x);
#define JN_PAIR(hi, lo, j) __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)(j) | ((uint32_t)((j) + 1) << 16))
  const int v_end = min(v_begin + kPlaneRows, H);
  for (int v = v_begin; v < v_end; v++) {
    const uint32_t lo2 = word(v + 1, x - 4), hi2 = word(v + 1, x);
    // bytes b0..b7 = image columns x-4 .. x+3; pair(j) = (b_j, b_j+1) as two 16-bit halves
    uint32_t S[3];                                                           // S = I[v-1] + 2 I[v] + I[v+1] at bytes (1,2), (3,4), (5,6)
    pk16 T[3];                                                               // T = I[v-1] - I[v+1] at bytes (2,3), (4,5), (6,7)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      S[j] = JN_PAIR(hi0, lo0, 2 * j + 1) + 2 * JN_PAIR(hi1, lo1, 2 * j + 1) + JN_PAIR(hi2, lo2, 2 * j + 1);   // <= 1020 per half: no carry
      T[j] = as_pk(JN_PAIR(hi0, lo0, 2 * j + 2)) - as_pk(JN_PAIR(hi2, lo2, 2 * j + 2));
    }
    // du(c) = sat(((S(c-1) - S(c+1)) >> 2) + 128) at columns x-2 .. x+1 = plane bytes x .. x+3
    const pk16 d01 = sobel_norm(as_pk(S[0]) - as_pk(S[1])), d23 = sobel_norm(as_pk(S[1]) - as_pk(S[2]));
    *reinterpret_cast<uint32_t*>(pdu + (size_t)v * Wp + x) = __builtin_amdgcn_perm(as_u32(d23), as_u32(d01), 0x06040200u);
    // dv(c) = sat(((T(c-1) + 2 T(c) + T(c+1)) >> 2) + 128) at columns x-1 .. x+2 = plane bytes x .. x+3
    const pk16 m01 = as_pk(__builtin_amdgcn_perm(as_u32(T[1]), as_u32(T[0]), 0x05040302u));   // (T3, T4)
    const pk16 m23 = as_pk(__builtin_amdgcn_perm(as_u32(T[2]), as_u32(T[1]), 0x05040302u));   // (T5, T6)
    const pk16 e01 = sobel_norm(T[0] + m01 + m01 + T[1]), e23 = sobel_norm(T[1] + m23 + m23 + T[2]);
    *reinterpret_cast<uint32_t*>(pdv + (size_t)v * Wp + x) = __builtin_amdgcn_perm(as_u32(e23), as_u32(e01), 0x06040200u);
    lo0 = lo1; hi0 = hi1; lo1 = lo2; hi1 = hi2;
  }
#undef JN_PAIR
}

// ------------------------------------------------------------------------------------------------
// Support matching (elas.cpp:269-373 per candidate, :395-413 forward + backward check).
// One wave64 per lattice candidate; lanes stride the disparity range, keep (best, first d of best,
// second best) privately, then a 6-step butterfly merges them.  The reference's sequential
// "best / second best with strict <" equals: E1 = min, d1 = smallest d attaining it, E2 = second
// smallest of the multiset — which is what the merge computes.
struct Best { int e1, d1, e2; };
DEV Best merge(Best a, int e1, int d1, int e2) {
  Best r;
  const bool take_b = (e1 < a.e1) || (e1 == a.e1 && d1 >= 0 && (a.d1 < 0 || d1 < a.d1));
  r.e1 = take_b ? e1 : a.e1;
  r.d1 = take_b ? d1 : a.d1;
  const int hi = a.e1 > e1 ? a.e1 : e1;          // larger of the two minima
  const int lo2 = a.e2 < e2 ? a.e2 : e2;         // smaller of the two runners-up
  r.e2 = hi < lo2 ? hi : lo2;
  return r;
}

DEV int match_candidate(const DevParams& dp, const uint4* __restrict__ A, const uint4* __restrict__ B, int u, int v,
                        bool right, int lane) {
  const int W = dp.W, H = dp.H;
  if (!(u >= 5 && u <= W - 6 && v >= 5 && v <= H - 6)) return -1;            // :283
  if (texture16(A[(size_t)v * W + u]) < dp.support_texture) return -1;       // :301-305
  const int dmax = right ? min(dp.disp_max, W - u - 5) : min(dp.disp_max, u - 5);   // :325-326
  const int dmin = dp.disp_min;                                              // :323 (max(disp_min, 0), taken by the host)
  if (dmax - dmin < 10) return -1;                                           // :329
  const size_t top = (size_t)(v - 2) * W, bot = (size_t)(v + 2) * W;
  const uint4 a0 = A[top + u - 2], a1 = A[top + u + 2], a2 = A[bot + u - 2], a3 = A[bot + u + 2];
  Best b{32767, -1, 32767};
  for (int d = dmin + lane; d <= dmax; d += 64) {                            // :333
    const int uw = right ? u + d : u - d;
    const int s = sad16(a0, B[top + uw - 2]) + sad16(a1, B[top + uw + 2]) + sad16(a2, B[bot + uw - 2]) + sad16(a3, B[bot + uw + 2]);
    if (s < b.e1) { b.e2 = b.e1; b.e1 = s; b.d1 = d; }
    else if (s < b.e2) b.e2 = s;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int e1 = __shfl_xor(b.e1, off), d1 = __shfl_xor(b.d1, off), e2 = __shfl_xor(b.e2, off);
    b = merge(b, e1, d1, e2);
  }
  // :366 — dmax - dmin >= 10 guarantees a second-best exists
  if (b.d1 >= 0 && (float)b.e1 < dp.support_threshold * (float)b.e2) return b.d1;
  return -1;
}

__global__ void __launch_bounds__(256) k_support(DevParams dp, int n, const uint4* __restrict__ desc, int16_t* __restrict__ d_can) {
  const int lane = threadIdx.x & 63;
  const int cand = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int frame = blockIdx.y;
  if (cand >= dp.cw * dp.ch) return;
  const int uc = cand % dp.cw, vc = cand / dp.cw;
  int16_t out = 0;                                                         // row/column 0 keep calloc's 0 (:388,:395-397)
  if (uc >= 1 && vc >= 1) {
    const uint4* L = desc + (size_t)frame * dp.H * dp.W;
    const uint4* R = desc + (size_t)(n + frame) * dp.H * dp.W;
    const int u = uc * dp.step, v = vc * dp.step;
    out = -1;
    const int d = match_candidate(dp, L, R, u, v, false, lane);
    if (d >= 0) {
      const int d2 = match_candidate(dp, R, L, u - d, v, true, lane);
      if (d2 >= 0 && abs(d - d2) <= dp.lr_threshold) out = (int16_t)d;
    }
  }
  if (lane == 0) d_can[(size_t)frame * dp.cw * dp.ch + cand] = out;
}

// LDS variant: one 1024-thread workgroup per lattice row stages the four descriptor rows its candidates
// read (v-2 and v+2 of both images, 64*W bytes).  LANES lanes share a candidate and stride the disparity
// range, merged by log2(LANES) xor-shuffles.  Measured at 720p/D=128: 4 lanes 416 us, 8 lanes 329 us,
// 16 lanes 383 us, 1 lane (no merge, 256 threads) 447 us per 32-pair batch — 8 consecutive descriptors per
// candidate spread the ds_read_b128 over more banks than 4 do, 16 pay more for the merge.  Used when
// 64*W bytes fit the 160 KB LDS; otherwise the global-memory kernel above runs.
enum { kSupportLanes = 4 };
// LDS layout of k_support_lds: four descriptor rows [Ltop | Lbot | Rtop | Rbot] of PITCH columns each, PITCH a
// compile-time constant so that a candidate's taps (u-2 / u+2, top / bottom) sit at immediate offsets from one
// address.  Four lanes per candidate, lane j taking the disparities j, j+4, ...: consecutive columns per lane and
// 5-column steps between candidates spread a wave's 16-byte reads evenly over the 64 banks.  (Interleaving top and
// bottom per column, or splitting the range between two lane groups 64 columns apart, puts the groups on the same
// banks and costs 10-30 %.)  PMC: 21 M ds_read_b128 per launch at 8 lanes x 4 reads per disparity = 277 us of LDS
// pipe time out of 317 us; with the tap-pair re-use below it is half that and the kernel runs in 262 us.
// The running best / second best (reference: strict `<`, first d wins, elas.cpp:354-362) are kept as packed keys
// energy << 16 | d: E1 = smallest key's energy with the smallest d attaining it, E2 = second smallest key's energy =
// second smallest energy of the multiset (two disparities sharing the minimum give E2 = E1, as in the reference).
template <int LANES, int PITCH, typename TEX>
DEV int quad_match(const DevParams& dp, const uint4* __restrict__ At, const uint4* __restrict__ Bt,
                   TEX&& texture_at, int u, bool right, bool active, int j) {
  static_assert(LANES == 4, "lane j walks the disparities j, j+4, j+8, ...");
  const int W = dp.W;
  bool ok = active && u >= 5 && u <= W - 6;                                   // :283 (rows checked by the caller)
  if (ok) ok = texture_at(u) >= dp.support_texture;                           // :301-305 (the descriptor at (u, v) itself)
  const int dmax = right ? min(dp.disp_max, W - u - 5) : min(dp.disp_max, u - 5);   // :325-326
  const int dmin = dp.disp_min;                                               // :323
  ok = ok && dmax - dmin >= 10;                                               // :329
  constexpr unsigned kNone = 0x7FFFFFFFu;
  unsigned k1 = kNone, k2 = kNone;
  if (ok && dmin + j <= dmax) {
    // Stepping d by 4 moves the matched column by 4 = the distance between the two tap columns, so the tap pair that
    // led (column +/- 2 in the direction of motion) is the trailing pair of the next step: one new (top, bottom) pair
    // is read from LDS per disparity instead of two.  Three register sets rotate through lead / trail / prefetch,
    // hence the loop unrolled by three.  (The kernel was bound by LDS bandwidth: 4 x 16 B per lane and disparity.)
    const int dir = right ? 1 : -1;
    const uint4* a = At + (u - 2);
    const uint4 am_t = a[0], ap_t = a[4], am_b = a[PITCH], ap_b = a[PITCH + 4];  // descriptors at u-2 / u+2, top / bottom row
    const uint4 aL_t = right ? ap_t : am_t, aL_b = right ? ap_b : am_b;       // partner of the leading pair
    const uint4 aT_t = right ? am_t : ap_t, aT_b = right ? am_b : ap_b;       // partner of the trailing pair
    int d = dmin + j;                                                         // :333 (lane j walks dmin + j, dmin + j + 4, ...)
    const uint4* b = Bt + (u + dir * d);                                      // column matched at disparity d
    uint4 p0_t = b[-2 * dir], p0_b = b[PITCH - 2 * dir];                      // trailing pair
    uint4 p1_t = b[2 * dir], p1_b = b[PITCH + 2 * dir];                       // leading pair
    uint4 p2_t, p2_b;
    const uint4* nx = b + 6 * dir;                                            // leading column of the next step
#define JN_STEP(LD, TR, NX)                                                                                   \
    {                                                                                                         \
      const bool more = d + 4 <= dmax;                                                                        \
      NX##_t = nx[0]; NX##_b = nx[PITCH];   /* read ahead unconditionally: past the last step it stays inside the four staged rows */ \
      /* one accumulator chain on d: 16 x v_sad_hi_u8 give (energy << 16) | d directly */                     \
      const unsigned key = sadhi16(aT_b, TR##_b, sadhi16(aL_b, LD##_b, sadhi16(aT_t, TR##_t, sadhi16(aL_t, LD##_t, (unsigned)d)))); \
      k2 = med3u(k1, k2, key);              /* k1 <= k2: the second smallest of the three */                  \
      k1 = min(k1, key);                                                                                      \
      if (!more) break;                                                                                       \
      d += 4; nx += 4 * dir;                                                                                  \
    }
    for (;;) {
      JN_STEP(p1, p0, p2)
      JN_STEP(p2, p1, p0)
      JN_STEP(p0, p2, p1)
    }
#undef JN_STEP
  }
#pragma unroll
  for (int off = 1; off < LANES; off <<= 1) {
    const unsigned o1 = __shfl_xor(k1, off), o2 = __shfl_xor(k2, off);
    k2 = min(max(k1, o1), min(k2, o2));
    k1 = min(k1, o1);
  }
  const int e1 = (int)(k1 >> 16), d1 = (int)(k1 & 255u), e2 = k2 == kNone ? 32767 : (int)(k2 >> 16);
  return (ok && k1 != kNone && (float)e1 < dp.support_threshold * (float)e2) ? d1 : -1;   // :366
}

// A lattice row may be cut into `nseg` column segments, one workgroup each (images wider than the 1280-column bucket:
// a whole 1920-column row needs all 160 KB of LDS = one workgroup of 16 waves per CU; two half rows need 80 KB each and run
// two per CU).  A segment with candidates u in [u_lo, u_hi] stages the left rows over [u_lo - dmax - 2, u_hi + dmax + 2]
// (the backward match of the right-image candidate u - d walks up to dmax to the right again) and the right rows over
// [u_lo - dmax - 2, u_hi + 2]; the row pointers are shifted by the window start so that quad_match keeps indexing by column.
// Read-aheads past a window's end stay inside the four staged rows (the right rows follow the left ones).
// PL = true: `src` holds the Sobel planes (pitch Wp) and the four rows are ASSEMBLED here (the plane data flow above): wave-sized passes of
// 64 x 4 columns, eight 8-byte loads and 32 v_perm per lane and pass; the texture test assembles the candidate's own descriptor.
// PL = false: `src` holds materialised descriptors (the route kept for what the plane flow does not take, launch_support()).
template <int LANES, int PITCH, bool PL>
__global__ void __launch_bounds__(1024) k_support_lds(DevParams dp, int n, const void* __restrict__ src, int Wp, int16_t* __restrict__ d_can, int nseg) {
  extern __shared__ uint4 rows[];                       // [Ltop | Lbot | Rtop | Rbot], PITCH each
  // XCD-aware work order (as k_dense): consecutive workgroups go round-robin to the 8 XCDs, each with its own L2; workgroup b takes item
  // (b % 8) * per_xcd + b / 8 of the list ordered (frame, lattice row, segment), so that one XCD walks a frame's lattice rows in order —
  // neighbouring rows tap overlapping plane rows (v +- 4 around rows 5 apart) and a row's segments overlap by 2 disp_max columns.  [PMC: 0.43 GB
  // fetched per batch in launch order, every workgroup's rows from memory]
  int item = blockIdx.x;
  {
    const int total = dp.ch * nseg * n, per_xcd = (total + 7) / 8;
    item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= total) return;
  }
  const int seg = item % nseg, vc = (item / nseg) % dp.ch, frame = item / (nseg * dp.ch), W = dp.W;
  const int v = vc * dp.step, nthr = blockDim.x;           // 1024 threads, fewer when a segment has fewer than 256 candidates
  int16_t* out_row = d_can + ((size_t)frame * dp.ch + vc) * dp.cw;
  const int per_seg = (dp.cw + nseg - 1) / nseg, uc_lo = seg * per_seg, uc_hi = min(uc_lo + per_seg, dp.cw);   // candidates [uc_lo, uc_hi)
  const bool row_ok = vc >= 1 && v >= 5 && v <= dp.H - 6;
  if (!row_ok) {                                        // row 0 keeps calloc's zeros, other out-of-range rows are -1
    for (int uc = uc_lo + threadIdx.x; uc < uc_hi; uc += nthr) out_row[uc] = (vc == 0 || uc == 0) ? 0 : -1;
    return;
  }
  const int u_lo = uc_lo * dp.step, u_hi = (uc_hi - 1) * dp.step;
  const int c0 = nseg == 1 ? 0 : max(u_lo - dp.disp_max - 2, 0);                        // window start, both images
  const int c1L = nseg == 1 ? W : min(u_hi + dp.disp_max + 3, W), c1R = nseg == 1 ? W : min(u_hi + 3, W);
  uint4* Lt = rows - c0; uint4* Rt = rows + 2 * PITCH - c0;                             // indexed by image column
  const uint4* L = nullptr; const uint4* R = nullptr;
  __amdgpu_buffer_rsrc_t rduL, rdvL, rduR, rdvR;
  if constexpr (PL) {
    const uint8_t* planes = static_cast<const uint8_t*>(src);
    rduL = plane_rsrc(planes, frame, 0, dp.H, Wp); rdvL = plane_rsrc(planes, frame, 1, dp.H, Wp);
    rduR = plane_rsrc(planes, n + frame, 0, dp.H, Wp); rdvR = plane_rsrc(planes, n + frame, 1, dp.H, Wp);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = nthr >> 6, lane = threadIdx.x & 63;
    const int g0 = c0 >> 2, npass = (((c1L - 1) >> 2) - g0) / 64 + 1;                  // passes of 64 four-column groups per row (c1L >= c1R)
    for (int it = wave; it < 4 * npass; it += nwaves) {
      const int rr = it & 3, pass = it >> 2;                                          // which of the four rows (wave-uniform)
      const int g = g0 + pass * 64 + lane, c1 = rr < 2 ? c1L : c1R;
      const PlaneRows p = load_plane_rows(rr < 2 ? rduL : rduR, rr < 2 ? rdvL : rdvR, min(4 * g, Wp - 8), (rr & 1) ? v + 2 : v - 2, Wp);
      uint4* dst = rows + rr * PITCH - c0;
      const uint4 zero = make_uint4(0, 0, 0, 0);
      const int col = 4 * g;                                                          // descriptors exist for columns [3, W-4]
      if (col + 0 >= c0 && col + 0 < c1) dst[col + 0] = (col + 0 >= 3 && col + 0 <= W - 4) ? desc_from_rows<0>(p) : zero;
      if (col + 1 >= c0 && col + 1 < c1) dst[col + 1] = (col + 1 >= 3 && col + 1 <= W - 4) ? desc_from_rows<1>(p) : zero;
      if (col + 2 >= c0 && col + 2 < c1) dst[col + 2] = (col + 2 >= 3 && col + 2 <= W - 4) ? desc_from_rows<2>(p) : zero;
      if (col + 3 >= c0 && col + 3 < c1) dst[col + 3] = (col + 3 >= 3 && col + 3 <= W - 4) ? desc_from_rows<3>(p) : zero;
    }
  } else {
    L = static_cast<const uint4*>(src) + (size_t)frame * dp.H * W;
    R = static_cast<const uint4*>(src) + (size_t)(n + frame) * dp.H * W;
    for (int i = c0 + threadIdx.x; i < c1L; i += nthr) { Lt[i] = L[(size_t)(v - 2) * W + i]; Lt[PITCH + i] = L[(size_t)(v + 2) * W + i]; }
    for (int i = c0 + threadIdx.x; i < c1R; i += nthr) { Rt[i] = R[(size_t)(v - 2) * W + i]; Rt[PITCH + i] = R[(size_t)(v + 2) * W + i]; }
  }
  __syncthreads();
  // row v itself is only read for the texture test
  auto texL = [&](int u) -> int { if constexpr (PL) return texture16(descriptor_at(rduL, rdvL, u, v, W, dp.H, Wp)); else return texture16(L[(size_t)v * W + u]); };
  auto texR = [&](int u) -> int { if constexpr (PL) return texture16(descriptor_at(rduR, rdvR, u, v, W, dp.H, Wp)); else return texture16(R[(size_t)v * W + u]); };
  const int j = threadIdx.x & (LANES - 1);
  // Which candidate a quad of lanes takes.  A wave's 16-byte LDS reads are served in four groups of 16 lanes — quads
  // {0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15} (MI355X_MICROARCH.md, LDS: ds_read_b128) — and a descriptor column
  // c lies on banks 4c .. 4c+3 (mod 64), so a group is conflict-free when its 16 lanes read 16 different columns mod 16.
  // A candidate's four lanes read 4 consecutive columns and candidates are 5 columns apart: candidates i and i' share a
  // column mod 16 unless 5i and 5i' differ by at least 4 (mod 16), which holds exactly for i = i' (mod 4).  So each group
  // gets the four candidates of one residue class (round 2 took them in lane order: 59 % of the LDS cycles were conflicts).
  static_assert(LANES == 4, "the candidate permutation below is written for four lanes per candidate (16 candidates per wave); launch_support_pitch starts whole waves");
  const int quad = (threadIdx.x >> 2) & 15;
  const int cand_in_wave = (int)((0xFEAB6732DC894510ull >> (4 * quad)) & 15u);
  const int cand_in_wg = (threadIdx.x >> 6) * 16 + cand_in_wave;
  for (int uc0 = uc_lo; uc0 < uc_hi; uc0 += nthr / LANES) {
    const int uc = uc0 + cand_in_wg;
    const bool active = uc >= 1 && uc < uc_hi;
    const int u = uc * dp.step;
    int res = -1;
    const int d = quad_match<LANES, PITCH>(dp, Lt, Rt, texL, u, false, active, j);
    const int d2 = quad_match<LANES, PITCH>(dp, Rt, Lt, texR, u - d, true, active && d >= 0, j);
    if (d >= 0 && d2 >= 0 && abs(d - d2) <= dp.lr_threshold) res = d;         // :404-411
    if (j == 0 && uc < uc_hi) out_row[uc] = (int16_t)(uc == 0 ? 0 : res);
  }
}

// ------------------------------------------------------------------------------------------------
// Support-point filters on the candidate lattice, in place, with the reference's exact sequential
// semantics (removeInconsistentSupportPoints, elas.cpp:153-179; removeRedundantSupportPoints
// vertical then horizontal, elas.cpp:181-235, called at :416-422).  One workgroup per frame, the
// lattice (int16 [ch][cw]) lives in LDS.
//  * Inconsistency filter: the reference sweeps u-outer / v-inner and deletions act immediately, so
//    a point sees earlier points of its (2w+1)^2 window already filtered and later ones untouched.
//    With K = w+1, all points of equal t = K*u + v are independent (same-t points are >= K rows
//    apart, earlier-in-sweep window points have smaller t, later ones larger), so the workgroup walks t
//    as a skewed wavefront: sixteen lanes per point share the window, ~ch/K points per step.
//  * Redundancy filters: the vertical pass only looks along a column, the horizontal pass only along
//    a row, so columns resp. rows are independent and each thread sweeps one line sequentially.
// The LDS copy carries a border of WIN invalid cells on every side, so no window needs bounds checks and the
// (2*WIN+1)^2 loops unroll completely (the window size is a template parameter; the reference uses 5).
// One line of a redundancy pass (removeRedundantSupportPoints, elas.cpp:181-235; max_dist 5, threshold 1):
// a point goes when both directions along the line hold a point within 1 of it.  The sweep is in place,
// so "before" sees this pass's deletions and "after" does not.  An 11-cell register window slides along
// the line: one LDS read per step, issued a step ahead, and branch-free compares.
__device__ __forceinline__ void redundant_line(int16_t* line, int stride, int n) {
  // The window of 11 values is a RING in registers: element e of the line sits in slot (e + 5) mod 11, the loop is unrolled by 11 so that every
  // slot index is a constant, and the element that enters (i + 6) takes the slot of the one that leaves (i - 5).  (Rounds 1-4 shifted the
  // window by ten register moves a cell — a third of the pass, which one thread a line walks alone: instruction issue is all it costs.)
  int w[11];
#pragma unroll
  for (int k = 0; k < 11; k++) w[k] = line[(k - 5) * stride];
  for (int i0 = 0; i0 < n; i0 += 11) {
#pragma unroll
    for (int s = 0; s < 11; s++) {
      const int i = i0 + s;
      if (i >= n) break;
      const int nxt = line[min(i + 6, n + 4) * stride];
      const int d = w[(s + 5) % 11];
      const int lo = max(d - 1, 0);
      const unsigned span = (unsigned)(d + 1 - lo);
      bool before = false, after = false;
#pragma unroll
      for (int k = 0; k < 5; k++) before |= (unsigned)(w[(s + k) % 11] - lo) <= span;
#pragma unroll
      for (int k = 6; k < 11; k++) after |= (unsigned)(w[(s + k) % 11] - lo) <= span;
      if (d >= 0 && before && after) { line[i * stride] = -1; w[(s + 5) % 11] = -1; }
      w[s] = nxt;                                          // slot of element i - 5 = (i0 + s) mod 11 = s
    }
  }
}

constexpr int kFilterThreads = 512;
// Exclusive prefix sum of `mine` over the workgroup's kFilterThreads threads (thread order), and the total: a DPP scan inside each wave
// (row_shr 1, 2, 4, 8, row_bcast 15 / 31), the waves' totals through LDS — two barriers where the log-step scan through LDS of rounds 1-4
// took eighteen (k_filter_resolve runs three of these on a lone pair's critical path).
__device__ __forceinline__ int block_scan_excl(int mine, int& total) {
  __shared__ int s_wtot[kFilterThreads / 64];
  int v = mine;
#define JN_DPP_ADD(ctrl, rmask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, true)
  JN_DPP_ADD(0x111, 0xf); JN_DPP_ADD(0x112, 0xf); JN_DPP_ADD(0x114, 0xf); JN_DPP_ADD(0x118, 0xf);
  JN_DPP_ADD(0x142, 0xa); JN_DPP_ADD(0x143, 0xc);
#undef JN_DPP_ADD
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 63) s_wtot[wv] = v;
  __syncthreads();
  int before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kFilterThreads / 64; w++) { const int t = s_wtot[w]; before += w < wv ? t : 0; all += t; }
  __syncthreads();                                        // s_wtot may be written again by the next scan
  total = all;
  return before + v - mine;
}

// Copy lattice columns [c0,c1) x rows [r0,r1) (cells outside the lattice read as invalid) into an LDS block of
// pitch pw, and the interior back.
__device__ __forceinline__ void lattice_load(int16_t* s, const int16_t* g, int cw, int ch, int c0, int c1, int r0, int r1, int pw) {
  const int w = c1 - c0, n = w * (r1 - r0);
  for (int i = threadIdx.x; i < n; i += kFilterThreads) {
    const int r = r0 + i / w, c = c0 + i % w;
    s[(r - r0) * pw + (c - c0)] = (r >= 0 && r < ch && c >= 0 && c < cw) ? g[r * cw + c] : (int16_t)-1;
  }
}
__device__ __forceinline__ void lattice_store(const int16_t* base, int16_t* g, int cw, int u0, int u1, int v0, int v1, int pw) {
  const int w = u1 - u0, n = w * (v1 - v0);                 // base[(v - v0) * pw + (u - u0)] = lattice (u, v)
  for (int i = threadIdx.x; i < n; i += kFilterThreads) {
    const int v = v0 + i / w, u = u0 + i % w;
    g[v * cw + u] = base[(v - v0) * pw + (u - u0)];
  }
}
// Whole-lattice variants for widths that are multiples of 8: 16-byte global accesses, all of a thread's loads in
// flight together (the generic loops above wait out one 2-byte load per iteration), the border written separately.
__device__ __forceinline__ void lattice_load_all(int16_t* s, const int16_t* g, int cw, int ch, int win, int pw) {
  const int ph = ch + 2 * win;
  if (cw & 7) { lattice_load(s, g, cw, ch, -win, cw + win, -win, ch + win, pw); return; }
  for (int i = threadIdx.x; i < pw * ph; i += kFilterThreads) {            // border cells
    const int r = i / pw, c = i - r * pw;
    if (r < win || r >= ch + win || c < win || c >= cw + win) s[i] = -1;
  }
  const int per_row = cw >> 3, total = per_row * ch;
  const uint4* g4 = reinterpret_cast<const uint4*>(g);
#pragma unroll 4
  for (int i = threadIdx.x; i < total; i += kFilterThreads) {
    const int r = i / per_row, c = (i - r * per_row) << 3;
    const uint4 x = g4[i];
    int16_t* d = s + (r + win) * pw + win + c;
    d[0] = (int16_t)(x.x & 0xFFFF); d[1] = (int16_t)(x.x >> 16); d[2] = (int16_t)(x.y & 0xFFFF); d[3] = (int16_t)(x.y >> 16);
    d[4] = (int16_t)(x.z & 0xFFFF); d[5] = (int16_t)(x.z >> 16); d[6] = (int16_t)(x.w & 0xFFFF); d[7] = (int16_t)(x.w >> 16);
  }
}
__device__ __forceinline__ void lattice_store_all(const int16_t* s, int16_t* g, int cw, int ch, int win, int pw) {
  if (cw & 7) { lattice_store(s + win * pw + win, g, cw, 0, cw, 0, ch, pw); return; }
  const int per_row = cw >> 3, total = per_row * ch;
  uint4* g4 = reinterpret_cast<uint4*>(g);
#pragma unroll 4
  for (int i = threadIdx.x; i < total; i += kFilterThreads) {
    const int r = i / per_row, c = (i - r * per_row) << 3;
    const uint16_t* d = reinterpret_cast<const uint16_t*>(s + (r + win) * pw + win + c);
    g4[i] = make_uint4(d[0] | ((uint32_t)d[1] << 16), d[2] | ((uint32_t)d[3] << 16), d[4] | ((uint32_t)d[5] << 16), d[6] | ((uint32_t)d[7] << 16));
  }
}
// The skewed wavefront of the inconsistency filter over columns [0,ncols) of an LDS block (base = cell (0,0),
// a border of WIN cells readable on every side).  L lanes per point share the (2*WIN+1)^2 window cells;
// kFilterThreads / L points per step cover lattices of up to K * kFilterThreads / L rows.
template <int WIN, int L>
__device__ __forceinline__ void incon_wavefront(int16_t* base, int pw, int ncols, int ch, int tol, int min_support) {
  constexpr int K = WIN + 1, CELLS = (2 * WIN + 1) * (2 * WIN + 1);
  const int tid = threadIdx.x, j = tid / L, part = tid % L;
  constexpr int NC = (CELLS + L - 1) / L;
  int off[NC];
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int cell = min(part + L * c, CELLS - 1);
    off[c] = (cell / (2 * WIN + 1) - WIN) * pw + cell % (2 * WIN + 1) - WIN;
  }
  const bool tail = part + L * (NC - 1) < CELLS;         // whether this lane's last slot is a real cell
  // step t = K*q + phase: point j sits at v = phase + K*j, u = q - j.  q and phase are wave-uniform, so the
  // lane-dependent part of the address is the constant K*j*pw - j and the bounds are two compares.
  const int lane_cell = K * j * pw - j, v_room = ch - K * j;
  const int qmax = (K * (ncols - 1) + ch - 1) / K;
  for (int q = 0; q <= qmax; q++) {
#pragma unroll
    for (int phase = 0; phase < K; phase++) {
      const bool on = phase < v_room && (unsigned)(q - j) < (unsigned)ncols;
      int16_t* p = base + (on ? lane_cell + phase * pw + q : 0);
      const int d = on ? (int)p[0] : -1;
      // all loads first, then branch-free arithmetic: a short-circuit on e >= 0 makes the compiler wait
      // out every LDS read one at a time.  e in [max(d-tol,0), d+tol] as one unsigned compare.
      int e[NC];
#pragma unroll
      for (int c = 0; c < NC; c++) e[c] = p[off[c]];
      const int lo = max(d - tol, 0);
      const unsigned span = (unsigned)(d + tol - lo);
      int count = 0;
#pragma unroll
      for (int c = 0; c < NC; c++)
        count += (int)((unsigned)(e[c] - lo) <= span) & (int)(c < NC - 1 || tail);
      count += __builtin_amdgcn_update_dpp(0, count, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
      count += __builtin_amdgcn_update_dpp(0, count, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
      count += __builtin_amdgcn_update_dpp(0, count, 0x141, 0xF, 0xF, false);   // row_half_mirror
      if (L == 16) count += __builtin_amdgcn_update_dpp(0, count, 0x140, 0xF, 0xF, false);   // row_mirror
      // points of one step are K rows apart, outside each other's windows: their writes touch nothing
      // this step reads, so one barrier (before the next step's reads) is enough.
      if (d >= 0 && part == 0 && count < min_support) p[0] = -1;
      __syncthreads();
    }
  }
}
// ---- The inconsistency filter without the serial sweep --------------------------------------------------------
// When the sweep reaches a point p, the points after p (in sweep order) still hold their original values, so they
// contribute later(p) = #{q in window, q after p, agreeing} no matter what happened before; the points before p
// contribute only if they survived.  Hence, from the ORIGINAL lattice alone:
//   1 + later(p) >= min_support                 p survives for sure
//   1 + later(p) + earlier0(p) < min_support    p is deleted for sure   (earlier0 = agreeing points before p)
// and only the rest — a few per cent, at the rims of sparse regions — depends on the fate of earlier points.
// k_filter_classify computes that in parallel over all points and frames (code: 0 dead or invalid, 255 survivor,
// otherwise 1 + later(p), the part of the count that is already certain).  k_filter_resolve then walks the undecided
// points of a frame in sweep order with one wave: each looks up which of its earlier agreeing neighbours survived
// (all decided by then), one lane per window cell.  Exactly the reference's result, ~30 dependent steps per frame
// instead of 6*cw.
template <int WIN>
__global__ void __launch_bounds__(256) k_filter_classify(DevParams dp, int tol, int min_support, const int16_t* __restrict__ d_can,
                                                         uint8_t* __restrict__ code) {
  constexpr int T = 16, TW = T + 2 * WIN;
  __shared__ int16_t s_t[TW][TW + 2];
  const int cw = dp.cw, ch = dp.ch, frame = blockIdx.z;
  const int u0 = blockIdx.x * T, v0 = blockIdx.y * T;
  const int16_t* g = d_can + (size_t)frame * cw * ch;
  for (int i = threadIdx.x; i < TW * TW; i += 256) {
    const int r = i / TW, c = i - r * TW, v = v0 - WIN + r, u = u0 - WIN + c;
    s_t[r][c] = (v >= 0 && v < ch && u >= 0 && u < cw) ? g[v * cw + u] : (int16_t)-1;
  }
  __syncthreads();
  const int x = threadIdx.x & (T - 1), y = threadIdx.x / T, u = u0 + x, v = v0 + y;
  if (u >= cw || v >= ch) return;
  const int d = s_t[y + WIN][x + WIN];
  int out = 0;
  if (d >= 0) {
    const int lo = max(d - tol, 0);
    const unsigned span = (unsigned)(d + tol - lo);
    int later = 0, earlier = 0;
#pragma unroll
    for (int du = -WIN; du <= WIN; du++)
#pragma unroll
      for (int dv = -WIN; dv <= WIN; dv++) {
        if (du == 0 && dv == 0) continue;
        const int hit = (unsigned)((int)s_t[y + WIN + dv][x + WIN + du] - lo) <= span ? 1 : 0;
        if (du > 0 || (du == 0 && dv > 0)) later += hit; else earlier += hit;      // sweep order: u outer, v inner
      }
    const int sure = 1 + later;
    out = sure >= min_support ? 255 : (sure + earlier < min_support ? 0 : sure);
  }
  code[(size_t)frame * cw * ch + (size_t)u * ch + v] = (uint8_t)out;             // column-major = sweep order
}
constexpr int kFilterTodo = 2048;                                   // undecided points listed per frame (more: in-order scan)
template <int WIN>
__global__ void __launch_bounds__(kFilterThreads) k_filter_resolve(DevParams dp, int tol, int min_support, int16_t* __restrict__ d_can,
                                                                   const uint8_t* __restrict__ code, int16_t* __restrict__ list,
                                                                   int32_t* __restrict__ count, int list_cap, int rounds) {
  extern __shared__ int16_t s_lat[];                      // lattice with a border of WIN cells, then the codes
  static_assert(WIN == 5, "redundant_line's window is the reference's fixed max_dist 5");
  const int cw = dp.cw, ch = dp.ch, tid = threadIdx.x, N = cw * ch;
  const int pw = cw + 2 * WIN, ph = ch + 2 * WIN;
  int16_t* g = d_can + (size_t)blockIdx.x * N;
  uint8_t* s_code = reinterpret_cast<uint8_t*>(s_lat + (size_t)pw * ph);
  lattice_load_all(s_lat, g, cw, ch, WIN, pw);
  {
    const uint8_t* src = code + (size_t)blockIdx.x * N;
    if ((N & 3) == 0 && ((pw * ph) & 1) == 0) {                          // dword copy: frame offset and LDS offset are multiples of 4
      const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src);
      uint32_t* d4 = reinterpret_cast<uint32_t*>(s_code);
#pragma unroll 4
      for (int i = tid; i < (N >> 2); i += kFilterThreads) d4[i] = s4[i];
    } else
      for (int i = tid; i < N; i += kFilterThreads) s_code[i] = src[i];
  }
  __syncthreads();
  int16_t* base = s_lat + WIN * pw + WIN;                  // base[v * pw + u] = lattice (u, v)
  // List the undecided points in sweep order: every thread owns a contiguous stretch of the (column-major) codes,
  // counts its undecided points, an exclusive scan over the threads gives its place in the list.  Points that are
  // dead for sure leave the lattice on the way (the resolution below never looks at the value of a dead point).
  __shared__ uint16_t s_todo[kFilterTodo];
  const int per = (N + kFilterThreads - 1) / kFilterThreads, i_lo = min(tid * per, N), i_hi = min(i_lo + per, N);
  int mine = 0;
  {
    int u = i_lo / ch, v = i_lo - u * ch;
    for (int i = i_lo; i < i_hi; i++) {
      const int c = s_code[i];
      mine += (c != 0 && c != 255) ? 1 : 0;
      if (c == 0) base[v * pw + u] = -1;
      if (++v == ch) { v = 0; u++; }
    }
  }
  int scan_total1;
  const int scan_at1 = block_scan_excl(mine, scan_total1);
  const int total = scan_total1;
  if (total <= kFilterTodo) {
    int at = scan_at1;
    for (int i = i_lo; i < i_hi; i++) { const int c = s_code[i]; if (c != 0 && c != 255) s_todo[at++] = (uint16_t)i; }
  }
  __syncthreads();
  {
    // lane -> one of the 60 window cells that precede the point in sweep order: 5 columns to the left (11 rows each),
    // then the 5 cells above in its own column
    constexpr int ROWS = 2 * WIN + 1, LEFT = WIN * ROWS;
    const int lane = tid & 63, wv = tid >> 6;
    const int du = lane < LEFT ? lane / ROWS - WIN : 0;
    const int dv = lane < LEFT ? lane % ROWS - WIN : lane - LEFT - WIN;
    const bool cell = lane < LEFT + WIN;
    if (total <= kFilterTodo && rounds) {
      // The walk in sweep order, without the walk: in every round each wave takes every 8th listed point that is still undecided; a point
      // whose earlier agreeing neighbours are ALL decided gets its verdict now (its fate depends on nothing else, and theirs is final), the
      // others wait for the next round.  The earliest undecided point of the sweep order is never blocked, so every round makes progress;
      // the undecided points sit at the rims of sparse regions, a few per cent of the lattice, and chains among them are short: two or
      // three rounds instead of one dependent step per point.  Reading a neighbour while another wave decides it is harmless: old code =
      // wait, new code = the final answer, and a dead point's value (-1) never agrees.
      for (;;) {
        int waiting = 0;
        for (int k = wv; k < total; k += kFilterThreads / 64) {
          const int pidx = s_todo[k];
          const int sure = s_code[pidx];
          if (sure == 0 || sure == 255) continue;              // decided in an earlier round (wave-uniform)
          const int u = pidx / ch, v = pidx - u * ch;
          const int d = base[v * pw + u];
          bool hit = false, blocked = false;
          if (cell) {
            const int e = base[(v + dv) * pw + (u + du)];      // the border reads as invalid
            if (e >= 0 && abs(d - e) <= tol) {
              const int nc = s_code[(u + du) * ch + (v + dv)];
              hit = nc == 255; blocked = nc != 0 && nc != 255;
            }
          }
          if (__ballot(blocked) != 0ull) { waiting = 1; continue; }
          const int count = sure + __popcll(__ballot(hit));
          if (lane == 0) {
            const bool lives = count >= min_support;
            s_code[pidx] = lives ? 255 : 0;
            if (!lives) base[v * pw + u] = -1;
          }
        }
        if (!__syncthreads_or(waiting)) break;
      }
    } else if (tid < 64) {                                   // more undecided points than the list holds (or JN_FILTER_ROUNDS=0): scan the codes in order, one wave
      auto resolve = [&](int pidx) {
        const int u = pidx / ch, v = pidx - u * ch;
        const int d = base[v * pw + u], sure = s_code[pidx];
        bool hit = false;
        if (cell) {
          const int e = base[(v + dv) * pw + (u + du)];
          if (e >= 0 && abs(d - e) <= tol) hit = s_code[(u + du) * ch + (v + dv)] == 255;
        }
        const int count = sure + __popcll(__ballot(hit));
        if (tid == 0) {
          const bool lives = count >= min_support;
          s_code[pidx] = lives ? 255 : 0;
          if (!lives) base[v * pw + u] = -1;
        }
      };
      for (int i0 = 0; i0 < N; i0 += 64) {
        const int c = i0 + tid < N ? (int)s_code[i0 + tid] : 0;
        unsigned long long todo = __ballot(c != 0 && c != 255);
        while (todo) { const int bit = __builtin_ctzll(todo); todo &= todo - 1; resolve(i0 + bit); }
      }
    }
  }
  __syncthreads();
  for (int u = tid; u < cw; u += kFilterThreads) redundant_line(base + u, pw, ch);        // vertical pass (elas.cpp:421)
  __syncthreads();
  for (int v = tid; v < ch; v += kFilterThreads) redundant_line(base + v * pw, 1, cw);    // horizontal pass (elas.cpp:422)
  __syncthreads();
  lattice_store_all(s_lat, g, cw, ch, WIN, pw);
  if (list) {
    // The support list (k_support_list's job: the surviving points in the reference's order, elas.cpp:424-431) straight from the lattice
    // in LDS — one launch and one trip through memory less on a lone pair's critical path.
    int mine2 = 0;
    {
      int u = i_lo / ch, v = i_lo - u * ch;
      for (int i = i_lo; i < i_hi; i++) {
        mine2 += (u >= 1 && v >= 1 && base[v * pw + u] >= 0) ? 1 : 0;
        if (++v == ch) { v = 0; u++; }
      }
    }
    int scan_total2;
    const int scan_at2 = block_scan_excl(mine2, scan_total2);
    int16_t* out = list + (size_t)blockIdx.x * list_cap * 3;
    int at = scan_at2;
    int u = i_lo / ch, v = i_lo - u * ch;
    for (int i = i_lo; i < i_hi; i++) {
      const int d = base[v * pw + u];
      if (u >= 1 && v >= 1 && d >= 0) {
        if (at < list_cap) { out[3 * at] = (int16_t)u; out[3 * at + 1] = (int16_t)v; out[3 * at + 2] = (int16_t)d; }
        at++;
      }
      if (++v == ch) { v = 0; u++; }
    }
    if (tid == 0) count[blockIdx.x] = scan_total2;
  }
}

// The same resolution for lattices too large to sit in LDS next to their codes (1920x1080: 166 KB + 83 KB): only the
// codes are in LDS, candidate values come from memory (a few dozen undecided points x 60 cells), deaths are written
// straight to the lattice; the redundancy passes follow in k_support_filters with the sweep switched off.
template <int WIN>
__global__ void __launch_bounds__(kFilterThreads) k_filter_resolve_big(DevParams dp, int tol, int min_support, int16_t* __restrict__ d_can,
                                                                       const uint8_t* __restrict__ code) {
  extern __shared__ int16_t s_lat[];
  uint8_t* s_code = reinterpret_cast<uint8_t*>(s_lat);
  __shared__ uint32_t s_todo[kFilterTodo / 2];             // 32-bit indices: these lattices can exceed 65536 points
  const int cw = dp.cw, ch = dp.ch, tid = threadIdx.x, N = cw * ch;
  int16_t* g = d_can + (size_t)blockIdx.x * N;
  const uint8_t* src = code + (size_t)blockIdx.x * N;
  for (int i = tid; i < N; i += kFilterThreads) s_code[i] = src[i];
  __syncthreads();
  const int per = (N + kFilterThreads - 1) / kFilterThreads, i_lo = min(tid * per, N), i_hi = min(i_lo + per, N);
  int mine = 0;
  {
    int u = i_lo / ch, v = i_lo - u * ch;
    for (int i = i_lo; i < i_hi; i++) {
      const int c = s_code[i];
      mine += (c != 0 && c != 255) ? 1 : 0;
      if (c == 0) g[v * cw + u] = -1;                        // dead for sure (or invalid already)
      if (++v == ch) { v = 0; u++; }
    }
  }
  int scan_total3;
  const int scan_at3 = block_scan_excl(mine, scan_total3);
  const int total = scan_total3;
  const bool listed = total <= kFilterTodo / 2;
  if (listed) {
    int at = scan_at3;
    for (int i = i_lo; i < i_hi; i++) { const int c = s_code[i]; if (c != 0 && c != 255) s_todo[at++] = (uint32_t)i; }
  }
  __syncthreads();
  if (tid >= 64) return;
  constexpr int ROWS = 2 * WIN + 1, LEFT = WIN * ROWS;
  const int du = tid < LEFT ? tid / ROWS - WIN : 0;
  const int dv = tid < LEFT ? tid % ROWS - WIN : tid - LEFT - WIN;
  const bool cell = tid < LEFT + WIN;
  auto resolve = [&](int pidx) {
    const int u = pidx / ch, v = pidx - u * ch;
    const int d = g[v * cw + u], sure = s_code[pidx];
    bool hit = false;
    const int uu = u + du, vv = v + dv;
    if (cell && uu >= 0 && uu < cw && vv >= 0 && vv < ch && s_code[uu * ch + vv] == 255) {   // survivors keep their value
      const int e = g[vv * cw + uu];
      hit = abs(d - e) <= tol;
    }
    const int count = sure + __popcll(__ballot(hit));
    if (tid == 0) {
      const bool lives = count >= min_support;
      s_code[pidx] = lives ? 255 : 0;
      if (!lives) g[v * cw + u] = -1;
    }
  };
  if (listed) {
    for (int k = 0; k < total; k++) resolve((int)s_todo[k]);
  } else {
    for (int i0 = 0; i0 < N; i0 += 64) {
      const int c = i0 + tid < N ? (int)s_code[i0 + tid] : 0;
      unsigned long long todo = __ballot(c != 0 && c != 255);
      while (todo) { const int bit = __builtin_ctzll(todo); todo &= todo - 1; resolve(i0 + bit); }
    }
  }
}

// Support list (elas.cpp:425-431) straight from the filtered lattice: lattice points with uc >= 1, vc >= 1 and a valid
// disparity, in the reference's order (u outer, v inner), as (uc, vc, d) int16 triples plus their count, written by
// the GPU into pinned host memory — the candidate lattice itself then never travels to the host.
// One workgroup per frame; every thread owns a contiguous stretch of the column-major index range.
__global__ void __launch_bounds__(kFilterThreads) k_support_list(DevParams dp, const int16_t* __restrict__ d_can, int16_t* __restrict__ list,
                                                                 int32_t* __restrict__ count, int cap) {
  const int cw = dp.cw, ch = dp.ch, tid = threadIdx.x, N = cw * ch;
  const int16_t* g = d_can + (size_t)blockIdx.x * N;
  int16_t* out = list + (size_t)blockIdx.x * cap * 3;
  const int per = (N + kFilterThreads - 1) / kFilterThreads, i_lo = min(tid * per, N), i_hi = min(i_lo + per, N);
  int mine = 0;
  {
    int u = i_lo / ch, v = i_lo - u * ch;
    for (int i = i_lo; i < i_hi; i++) {
      mine += (u >= 1 && v >= 1 && g[v * cw + u] >= 0) ? 1 : 0;
      if (++v == ch) { v = 0; u++; }
    }
  }
  int scan_total4;
  const int scan_at4 = block_scan_excl(mine, scan_total4);
  int at = scan_at4;
  int u = i_lo / ch, v = i_lo - u * ch;
  for (int i = i_lo; i < i_hi; i++) {
    const int d = g[v * cw + u];
    if (u >= 1 && v >= 1 && d >= 0) {
      if (at < cap) { out[3 * at] = (int16_t)u; out[3 * at + 1] = (int16_t)v; out[3 * at + 2] = (int16_t)d; }
      at++;
    }
    if (++v == ch) { v = 0; u++; }
  }
  if (tid == 0) count[blockIdx.x] = scan_total4;
}

// seg_c == 0: the whole lattice (plus border) sits in LDS for all three passes.  Otherwise the lattice is larger
// than the LDS and every pass streams it through in pieces, global memory holding the state in between: the
// inconsistency filter by column segments [u0,u1) — the sweep is column-major, so a segment only needs the final
// values of the WIN columns before it and the untouched WIN columns after it — the vertical redundancy pass by
// the same column segments, the horizontal one by row segments of seg_r rows.
template <int WIN, int L>
__global__ void __launch_bounds__(kFilterThreads) k_support_filters(DevParams dp, int tol, int min_support, int16_t* __restrict__ d_can,
                                                                    int seg_c, int seg_r, int sweep) {
  // sweep == 0: the inconsistency filter has been applied already (k_filter_classify + k_filter_resolve_big); only
  // the two redundancy passes run
  extern __shared__ int16_t s_lat[];
  static_assert(WIN == 5, "redundant_line's window is the reference's fixed max_dist 5");
  static_assert(L == 16 || L == 8, "the DPP reduction covers 8 or 16 lanes");
  const int cw = dp.cw, ch = dp.ch, tid = threadIdx.x;
  int16_t* g = d_can + (size_t)blockIdx.x * cw * ch;
  if (seg_c == 0) {
    const int pw = cw + 2 * WIN;
    lattice_load(s_lat, g, cw, ch, -WIN, cw + WIN, -WIN, ch + WIN, pw);
    __syncthreads();
    int16_t* base = s_lat + WIN * pw + WIN;                  // base[v * pw + u] = lattice (u, v)
    if (sweep) incon_wavefront<WIN, L>(base, pw, cw, ch, tol, min_support);
    __syncthreads();
    for (int u = tid; u < cw; u += kFilterThreads) redundant_line(base + u, pw, ch);        // vertical pass (elas.cpp:421)
    __syncthreads();
    for (int v = tid; v < ch; v += kFilterThreads) redundant_line(base + v * pw, 1, cw);    // horizontal pass (elas.cpp:422)
    __syncthreads();
    lattice_store(base, g, cw, 0, cw, 0, ch, pw);
    return;
  }
  const int pwc = seg_c + 2 * WIN;
  int16_t* cbase = s_lat + WIN * pwc + WIN;
  for (int u0 = 0; sweep && u0 < cw; u0 += seg_c) {          // inconsistency filter (elas.cpp:416)
    const int u1 = min(u0 + seg_c, cw);
    lattice_load(s_lat, g, cw, ch, u0 - WIN, u1 + WIN, -WIN, ch + WIN, pwc);
    __syncthreads();
    incon_wavefront<WIN, L>(cbase, pwc, u1 - u0, ch, tol, min_support);
    __syncthreads();
    lattice_store(cbase, g, cw, u0, u1, 0, ch, pwc);
    __threadfence();                                         // the next piece reads these columns back from memory
    __syncthreads();
  }
  for (int u0 = 0; u0 < cw; u0 += seg_c) {                   // vertical redundancy pass (elas.cpp:421)
    const int u1 = min(u0 + seg_c, cw);
    lattice_load(s_lat, g, cw, ch, u0 - WIN, u1 + WIN, -WIN, ch + WIN, pwc);
    __syncthreads();
    for (int u = tid; u < u1 - u0; u += kFilterThreads) redundant_line(cbase + u, pwc, ch);
    __syncthreads();
    lattice_store(cbase, g, cw, u0, u1, 0, ch, pwc);
    __threadfence();
    __syncthreads();
  }
  const int pwr = cw + 2 * WIN;
  for (int v0 = 0; v0 < ch; v0 += seg_r) {                   // horizontal redundancy pass (elas.cpp:422)
    const int v1 = min(v0 + seg_r, ch);
    lattice_load(s_lat, g, cw, ch, -WIN, cw + WIN, v0, v1, pwr);
    __syncthreads();
    for (int v = tid; v < v1 - v0; v += kFilterThreads) redundant_line(s_lat + v * pwr + WIN, 1, cw);
    __syncthreads();
    lattice_store(s_lat + WIN, g, cw, 0, cw, v0, v1, pwr);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Grid prior (createGrid, elas.cpp:579-659) as 256-bit candidate sets per 20x20 cell.
// mark: every support point sets d-1..d+1 in its cell (left: column u, right: column u-d).
DEV void grid_mark_point(const DevParams& dp, int frame, int u, int v, int d, uint32_t* __restrict__ mark);
__global__ void __launch_bounds__(256) k_grid_mark(DevParams dp, const FrameInfo* __restrict__ info, const uint8_t* __restrict__ payload,
                                                   long long payload_stride, uint32_t* __restrict__ mark) {
  const int i = blockIdx.x * 256 + threadIdx.x, frame = blockIdx.y;
  const FrameInfo& fi = info[frame];      // by reference: a by-value copy of the 48-byte struct lands in scratch memory
  if (!fi.ok || i >= fi.nsup) return;
  const int32_t* s = reinterpret_cast<const int32_t*>(payload + (long long)frame * payload_stride + fi.sup_offset) + 3 * i;
  grid_mark_point(dp, frame, s[0], s[1], s[2], mark);
}
// The same from the support list the GPU wrote itself (k_support_list: lattice coordinates and disparity, in the reference's order):
// without corner points (elas.cpp:435 add_corners = 0) the support points ARE that list, so the grid does not have to wait for the host
// stage — it is queued behind stage A and runs while the host triangulates.  A frame is ok when it has at least three points (:66-71).
__global__ void __launch_bounds__(256) k_grid_mark_list(DevParams dp, const int16_t* __restrict__ list, const int32_t* __restrict__ count, int cap,
                                                        uint32_t* __restrict__ mark) {
  const int i = blockIdx.x * 256 + threadIdx.x, frame = blockIdx.y;
  const int nsup = min(count[frame], cap);
  if (nsup < 3 || i >= nsup) return;
  const int16_t* s = list + ((size_t)frame * cap + i) * 3;
  grid_mark_point(dp, frame, (int)s[0] * dp.step, (int)s[1] * dp.step, (int)s[2], mark);
}
DEV void grid_mark_point(const DevParams& dp, int frame, int u, int v, int d, uint32_t* __restrict__ mark) {
  const size_t cells = (size_t)dp.gw * dp.gh;
  const int y = (int)floorf(__fdiv_rn((float)v, (float)dp.grid_size));                         // :606
  const int lo = max(d - 1, 0), hi = min(d + 1, dp.disp_max);                                  // :596-597
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const int x = side ? (int)floorf(__fdiv_rn((float)(u - d), (float)dp.grid_size))          // :605
                       : (int)floorf((float)(u / dp.grid_size));                               // :603 (integer division)
    if (x < 0 || x >= dp.gw || y < 0 || y >= dp.gh) continue;
    uint32_t* cell = mark + ((size_t)(frame * 2 + side) * cells + (size_t)y * dp.gw + x) * kGridWords;
    for (int dd = lo; dd <= hi; dd++) atomicOr(&cell[dd >> 5], 1u << (dd & 31));
  }
}
// dilate: 3x3 OR over the FLATTENED cell index, exactly the reference's pointer walk (:617-632):
// border columns wrap into the neighbouring rows, the first and last gw+1 cells stay empty.
__global__ void __launch_bounds__(256) k_grid_dilate(DevParams dp, const FrameInfo* __restrict__ info, const uint32_t* __restrict__ mark,
                                                     uint32_t* __restrict__ bits) {
  const int cells = dp.gw * dp.gh;
  const int i = blockIdx.x * 256 + threadIdx.x, fs = blockIdx.y;     // fs = frame*2 + side
  if (i >= cells * kGridWords || (info && !info[fs >> 1].ok)) return;   // info == nullptr: queued before the host stage (a failing frame's grid is never read)
  const int c = i / kGridWords, w = i % kGridWords;
  uint32_t acc = 0;
  if (c >= dp.gw + 1 && c < cells - dp.gw - 1) {
    const uint32_t* m = mark + (size_t)fs * cells * kGridWords;
    const int off[9] = {-dp.gw - 1, -dp.gw, -dp.gw + 1, -1, 0, 1, dp.gw - 1, dp.gw, dp.gw + 1};
#pragma unroll
    for (int k = 0; k < 9; k++) acc |= m[(size_t)(c + off[k]) * kGridWords + w];
  }
  bits[(size_t)fs * cells * kGridWords + i] = acc;
}

// ------------------------------------------------------------------------------------------------
// Triangle set-up: both plane fits (computeDisparityPlanes, elas.cpp:507-577, Gauss-Jordan with
// full pivoting in double, matrix.cpp:414-502), the validity flag (elas.cpp:872), the ascending-u
// corner sort and the three edge lines (elas.cpp:847-868).  One thread per triangle and side.
// Every operation is an explicitly rounded IEEE op in the reference's order, so the floats match.
DEV bool gauss_jordan3(double A[3][3], double b[3]) {
  int used[3] = {0, 0, 0};
#pragma unroll
  for (int step = 0; step < 3; step++) {
    double best = 0.0; int pr = 0, pc = 0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      if (used[r] == 1) continue;
#pragma unroll
      for (int c = 0; c < 3; c++)
        if (used[c] == 0 && fabs(A[r][c]) >= best) { best = fabs(A[r][c]); pr = r; pc = c; }    // `>=`: last maximum wins
    }
#pragma unroll
    for (int c = 0; c < 3; c++) if (c == pc) ++used[c];
    if (pr != pc) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double x = 0, y = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) { if (r == pr) x = A[r][c]; if (r == pc) y = A[r][c]; }
#pragma unroll
        for (int r = 0; r < 3; r++) { if (r == pr) A[r][c] = y; if (r == pc) A[r][c] = x; }
      }
      double x = 0, y = 0;
#pragma unroll
      for (int r = 0; r < 3; r++) { if (r == pr) x = b[r]; if (r == pc) y = b[r]; }
#pragma unroll
      for (int r = 0; r < 3; r++) { if (r == pr) b[r] = y; if (r == pc) b[r] = x; }
    }
    double piv = 0;
#pragma unroll
    for (int r = 0; r < 3; r++) if (r == pc) piv = A[r][r];
    if (fabs(piv) < 1e-20) return false;
    const double inv = __ddiv_rn(1.0, piv);
    double prow[3] = {0, 0, 0}, pb = 0;
#pragma unroll
    for (int r = 0; r < 3; r++)
      if (r == pc) {
        A[r][r] = 1.0;
#pragma unroll
        for (int c = 0; c < 3; c++) { A[r][c] = __dmul_rn(A[r][c], inv); prow[c] = A[r][c]; }
        b[r] = __dmul_rn(b[r], inv); pb = b[r];
      }
#pragma unroll
    for (int r = 0; r < 3; r++) {
      if (r == pc) continue;
      double f = 0;
#pragma unroll
      for (int c = 0; c < 3; c++) if (c == pc) { f = A[r][c]; A[r][c] = 0.0; }
#pragma unroll
      for (int c = 0; c < 3; c++) A[r][c] = __dsub_rn(A[r][c], __dmul_rn(prow[c], f));
      b[r] = __dsub_rn(b[r], __dmul_rn(pb, f));
    }
  }
  return true;
}

// the record of triangle t of a frame side (fp: the frame's payload): planes of both sides, edge lines, column and row ranges
DEV TriRec tri_record(const FrameInfo& fi, const uint8_t* __restrict__ fp, int side, int t) {
  const int32_t* sup = reinterpret_cast<const int32_t*>(fp + fi.sup_offset);
  const int32_t* c = reinterpret_cast<const int32_t*>(fp + fi.corner_offset[side]) + 3 * t;
  int su[3], sv[3], sd[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { const int32_t* s = sup + 3 * c[k]; su[k] = s[0]; sv[k] = s[1]; sd[k] = s[2]; }
  float pl[2][3];
#pragma unroll
  for (int s = 0; s < 2; s++) {                               // s = 0: left-image coordinates, 1: right-image
    double A[3][3], b[3];
#pragma unroll
    for (int r = 0; r < 3; r++) { A[r][0] = (double)(s ? su[r] - sd[r] : su[r]); A[r][1] = (double)sv[r]; A[r][2] = 1.0; b[r] = (double)sd[r]; }
    if (gauss_jordan3(A, b)) { pl[s][0] = (float)b[0]; pl[s][1] = (float)b[1]; pl[s][2] = (float)b[2]; }
    else { pl[s][0] = pl[s][1] = pl[s][2] = 0.0f; }
  }
  TriRec o;
  o.pa = pl[side][0]; o.pb = pl[side][1]; o.pc = pl[side][2];
  o.flags = ((double)fabsf(pl[side][0]) < 0.7 && (double)fabsf(pl[1 - side][0]) < 0.7) ? 1 : 0;
  float tu[3], tv[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { tu[k] = (float)(side ? su[k] - sd[k] : su[k]); tv[k] = (float)sv[k]; }
  // bubble sort of elas.cpp:847-854, unrolled: (j,k) = (1,0), (2,0), (2,1)
  if (tu[0] > tu[1]) { float x = tu[1]; tu[1] = tu[0]; tu[0] = x; x = tv[1]; tv[1] = tv[0]; tv[0] = x; }
  if (tu[0] > tu[2]) { float x = tu[2]; tu[2] = tu[0]; tu[0] = x; x = tv[2]; tv[2] = tv[0]; tv[0] = x; }
  if (tu[1] > tu[2]) { float x = tu[2]; tu[2] = tu[1]; tu[1] = x; x = tv[2]; tv[2] = tv[1]; tv[1] = x; }
  float ABa = 0, ACa = 0, BCa = 0;
  if ((int)tu[0] != (int)tu[1]) ABa = __fdiv_rn(__fsub_rn(tv[0], tv[1]), __fsub_rn(tu[0], tu[1]));
  if ((int)tu[0] != (int)tu[2]) ACa = __fdiv_rn(__fsub_rn(tv[0], tv[2]), __fsub_rn(tu[0], tu[2]));
  if ((int)tu[1] != (int)tu[2]) BCa = __fdiv_rn(__fsub_rn(tv[1], tv[2]), __fsub_rn(tu[1], tu[2]));
  o.ABa = ABa; o.ACa = ACa; o.BCa = BCa;
  o.ABb = __fsub_rn(tv[0], __fmul_rn(ABa, tu[0]));
  o.ACb = __fsub_rn(tv[0], __fmul_rn(ACa, tu[0]));
  o.BCb = __fsub_rn(tv[1], __fmul_rn(BCa, tu[1]));
  o.Au = (int16_t)tu[0]; o.Bu = (int16_t)tu[1]; o.Cu = (int16_t)tu[2];
  o.vmin = (int16_t)(min(sv[0], min(sv[1], sv[2])) - 1);      // truncated line values can land one row above the top corner
  o.vmax = (int16_t)max(sv[0], max(sv[1], sv[2]));
  return o;
}
__global__ void __launch_bounds__(256) k_tri_setup(DevParams dp, const FrameInfo* __restrict__ info, const uint8_t* __restrict__ payload,
                                                   long long payload_stride, int tri_cap, TriRec* __restrict__ recs) {
  const int t = blockIdx.x * 256 + threadIdx.x, frame = blockIdx.y, side = blockIdx.z;
  const FrameInfo& fi = info[frame];      // by reference: a by-value copy of the 48-byte struct lands in scratch memory
  if (!fi.ok || t >= fi.ntri[side]) return;
  recs[(size_t)(frame * 2 + side) * tri_cap + t] = tri_record(fi, payload + (long long)frame * payload_stride, side, t);
}

// ------------------------------------------------------------------------------------------------
// Triangle binning + per-tile rasterisation.  Eight lanes share a triangle; each lane takes some of
// the 32x8 tiles the triangle's bounding box touches, evaluates the reference's raster loops
// (elas.cpp:874-901) for the tile's 32 columns and, if any pixel is covered, appends
// {triangle, 32 row masks} to the tile's list.  Lists are capped at kBinCap; the count keeps growing
// past the cap so that the matcher can tell an overflowing tile and scan all triangles instead.
// Row masks of triangle q inside the 32x8 tile at (u0, v0): byte x of rows[] has bit r set iff the reference's
// raster loops visit pixel (u0 + x, v0 + r).  Returns false when the triangle misses the tile.
DEV bool bin_entry(const DevParams& dp, const TriRec& q, int t, int u0, int v0, int c0, int c1, BinEntry& e) {
  const int Au = q.Au, Bu = q.Bu, Cu = q.Cu;
  e.t = t; e.pa = q.pa; e.pb = q.pb; e.pc = q.pc; e.flags = q.flags; e.pad[0] = e.pad[1] = e.pad[2] = 0;
  uint32_t any = 0;
#pragma unroll
  for (int wd = 0; wd < kTileW / 4; wd++) {
    uint32_t packed = 0;
    if (u0 + wd * 4 + 3 >= c0 && u0 + wd * 4 <= c1)        // most of a tile's columns lie outside a ~10 px wide triangle
#pragma unroll
    for (int b = 0; b < 4; b++) {
      // straight-line per column: which edge bounds it (part 1 [Au,Bu): AB, :875-876; part 2 [Bu,Cu): BC, :890-891), the two
      // truncated line values (:878-879 / :893-894), their rows inside the tile as a mask (empty when the column is outside)
      const int uc = u0 + wd * 4 + b;
      const bool first = uc < Bu;
      const bool in = (first ? (Au != Bu && uc >= Au) : (Bu != Cu && uc < Cu)) && uc < dp.W;
      const float ea = first ? q.ABa : q.BCa, eb = first ? q.ABb : q.BCb;
      const float fu = (float)uc;
      const int v1 = (int)(unsigned)__fadd_rn(__fmul_rn(q.ACa, fu), q.ACb);
      const int v2 = (int)(unsigned)__fadd_rn(__fmul_rn(ea, fu), eb);
      const int lo = min(max(min(v1, v2) - v0, 0), kTileH), hi = max(min(max(v1, v2) - v0, kTileH), lo);   // rows [lo,hi) of this tile
      const uint32_t m = in ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
      packed |= m << (8 * b);
    }
    e.rows[wd] = packed; any |= packed;
  }
  return any != 0;                                        // bounding box touched the tile, the triangle may not
}
// 64 triangles per workgroup (one wave scans their box sizes).  Most triangles touch a handful of tiles, but Delaunay hulls carry a few long, thin
// ones whose bounding boxes span hundreds; any fixed lanes-per-triangle split lets those set the kernel's duration.
// So the (triangle, tile) pairs of the workgroup's triangles are numbered consecutively (prefix sum of the box
// sizes in LDS) and the 256 threads stride over that flat list.
enum { kBinTris = 64 };
// kSetup: the workgroup forms its 64 triangles' records itself (k_tri_setup's work, one thread a triangle) and writes them out for the
// kernels behind it, instead of reading them back from memory: one launch less on a batch's — and a lone pair's — critical path.
template <bool kSetup>
__global__ void __launch_bounds__(256) k_bin(DevParams dp, const FrameInfo* __restrict__ info, TriRec* __restrict__ recs,
                                             int tri_cap, int32_t* __restrict__ bin_count, BinEntry* __restrict__ bin_list,
                                             const uint8_t* __restrict__ payload, long long payload_stride) {
  __shared__ int s_first[kBinTris + 1];                    // first flat index of each triangle's tiles
  __shared__ int s_box[kBinTris][3];                       // tx0, ty0, ntx of the bounding box in tiles
  __shared__ int s_cols[kBinTris][2];                      // c0, c1
  const int frame = blockIdx.y, side = blockIdx.z, tid = threadIdx.x;
  const FrameInfo& fi = info[frame];      // by reference: a by-value copy of the 48-byte struct lands in scratch memory
  if (!fi.ok) return;
  const int tiles_x = (dp.W + kTileW - 1) / kTileW, tiles_y = (dp.H + kTileH - 1) / kTileH;
  const size_t base = (size_t)(frame * 2 + side) * tiles_x * tiles_y;
  TriRec* R = recs + (size_t)(frame * 2 + side) * tri_cap;
  const int t0 = blockIdx.x * kBinTris;
  // the workgroup's triangle records go to LDS in one coalesced sweep: read field by field from memory, every
  // item would wait out a dozen dependent global loads
  __shared__ TriRec s_rec[kBinTris];
  if constexpr (kSetup) {
    if (tid < kBinTris && t0 + tid < fi.ntri[side]) {
      const TriRec o = tri_record(fi, payload + (long long)frame * payload_stride, side, t0 + tid);
      s_rec[tid] = o;
      R[t0 + tid] = o;
    }
  } else {
    const int ntri_here = min(kBinTris, fi.ntri[side] - t0);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(R + t0);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_rec);
    for (int i = tid; i < ntri_here * (int)(sizeof(TriRec) / 4); i += 256) dst[i] = src[i];
  }
  __syncthreads();
  if (tid < kBinTris) {
    int total = 0;
    const int t = t0 + tid;
    if (t < fi.ntri[side]) {
      const TriRec& q = s_rec[tid];
      const int c0 = max(q.Au, 0), c1 = min(q.Cu, dp.W) - 1;                  // columns [Au, Cu)
      const int r0 = max((int)q.vmin, 0), r1 = min((int)q.vmax, dp.H - 1);
      if (c1 >= c0 && r1 >= r0) {
        const int tx0 = c0 / kTileW, ty0 = r0 / kTileH, ntx = c1 / kTileW - tx0 + 1;
        total = ntx * (r1 / kTileH - ty0 + 1);
        s_box[tid][0] = tx0; s_box[tid][1] = ty0; s_box[tid][2] = ntx;
        s_cols[tid][0] = c0; s_cols[tid][1] = c1;
      }
    }
    int incl = total;                                       // inclusive scan over the 32 lanes
#pragma unroll
    for (int off = 1; off < kBinTris; off <<= 1) { const int o = __shfl_up(incl, off); if (tid >= off) incl += o; }
    s_first[tid + 1] = incl;
    if (tid == 0) s_first[0] = 0;
  }
  __syncthreads();
  const int items = s_first[kBinTris];
  for (int idx = tid; idx < items; idx += 256) {
    int j = 0;                                              // triangle whose range holds idx (5-step binary search)
#pragma unroll
    for (int step = kBinTris / 2; step >= 1; step >>= 1) if (s_first[j + step] <= idx) j += step;
    const int k = idx - s_first[j], ntx = s_box[j][2];
    const int tx = s_box[j][0] + k % ntx, ty = s_box[j][1] + k / ntx;
    BinEntry e;
    if (!bin_entry(dp, s_rec[j], t0 + j, tx * kTileW, ty * kTileH, s_cols[j][0], s_cols[j][1], e)) continue;
    const size_t bin = base + (size_t)ty * tiles_x + tx;
    const int slot = atomicAdd(&bin_count[bin], 1);
    if (slot < kBinCap) bin_list[bin * kBinCap + slot] = e;
  }
}

// Does triangle (Au,Bu,Cu, lines) cover pixel (u,v) under the reference's raster loops
// (elas.cpp:874-901)?  Part 1 spans columns [Au,Bu) between lines AC and AB, part 2 spans [Bu,Cu)
// between AC and BC; rows are the half-open interval between the truncated line values.
DEV bool tri_covers(int Au, int Bu, int Cu, float ACa, float ACb, float ABa, float ABb, float BCa, float BCb, int u, int v) {
  float ea, eb;
  if (u < Bu) { if (Au == Bu || u < Au) return false; ea = ABa; eb = ABb; }
  else        { if (Bu == Cu || u >= Cu) return false; ea = BCa; eb = BCb; }
  const float fu = (float)u;
  const int v1 = (int)(unsigned)__fadd_rn(__fmul_rn(ACa, fu), ACb);           // :878 / :893
  const int v2 = (int)(unsigned)__fadd_rn(__fmul_rn(ea, fu), eb);             // :879 / :894
  return v >= min(v1, v2) && v < max(v1, v2);
}

// ------------------------------------------------------------------------------------------------
// Dense MAP matching (computeDisparity + findMatch, elas.cpp:683-907).  One 512-thread workgroup
// per 128x8 pixel strip and side; every thread owns 2 pixels of its row (64 columns apart), so that the
// 36 KB of LDS a strip needs still allow 32 waves per CU.
//  * The descriptors of the OTHER image that the strip can reach — 8 rows x (128 + disp_max)
//    columns — are staged once in LDS with coalesced 16-byte loads (2 loads per pixel instead of
//    one scattered 16-byte read per candidate disparity, ~13 per pixel); every candidate then
//    costs one ds_read_b128 + four v_sad_u8.
//  * Triangle lookup: k_bin left, per 32x8 tile, a list of {triangle, 8-bit row mask per tile
//    column} (bit r set <=> the reference's raster loops, elas.cpp:874-901, visit pixel
//    (u0+x, v0+r) for that triangle).  The wave that stages a tile's list ranks its triangles and
//    folds the masks into one 16-bit cover word per pixel, so a pixel finds its owner with one LDS
//    read and a count-leading-zeros.  Every pixel takes the LAST covering triangle in list order
//    — at a vertex column the two float edge lines of one triangle can round to different rows,
//    so a few pixels are covered twice and the reference keeps the later visitor's result
//    (findMatch's early-outs depend on the pixel only).
//  * Candidate disparities: the cell's 256-bit set is pre-masked per 32-bit word with the plane
//    range and the image-border range, so the inner loop is ctz -> read -> 4x v_sad_u8 -> select.
DEV uint32_t range_mask(int lo, int hi, int w) {               // bits of word w (disparities 32w..32w+31) inside [lo,hi]
  const int a = max(lo - 32 * w, 0), b = min(hi - 32 * w, 31);
  if (a > b) return 0u;
  return (0xFFFFFFFFu >> (31 - b)) & (0xFFFFFFFFu << a);
}

enum { kStripTiles = 4, kStripW = kStripTiles * kTileW, kDenseThreads = 512, kPxPerThread = kStripTiles * kTileW * kTileH / kDenseThreads };

__global__ void __launch_bounds__(kDenseThreads) k_dense(DevParams dp, int n, const FrameInfo* __restrict__ info,
                                               const TriRec* __restrict__ recs, int tri_cap, const int32_t* __restrict__ bin_count,
                                               const BinEntry* __restrict__ bin_list, const uint32_t* __restrict__ gridbits,
                                               const uint4* __restrict__ desc, int16_t* __restrict__ raw, int nbx, int nby, int xcd_order) {
  __shared__ uint32_t s_list[kStripTiles][kBinLds * kBinWords];   // candidate lists of the strip's four tiles
  __shared__ int s_cnt[kStripTiles];
  __shared__ uint16_t s_cover[kStripTiles][kTileH][kTileW];       // per pixel: bit k set <=> the k-th smallest listed triangle covers it
  __shared__ uint8_t s_slot[kStripTiles][kBinLds];                // rank -> list slot
  extern __shared__ uint4 s_B[];                             // [kTileH][kStripW + disp_max]
  // XCD-aware work order.  The hardware deals consecutive workgroups round-robin to the 8 XCDs, each with
  // its own 4 MB L2.  Workgroup b therefore takes logical item (b % 8) * per_xcd + b / 8, so that every
  // XCD walks a contiguous run of items ordered (frame, strip row, strip column, side): the two sides of a
  // strip read the same descriptor rows of both images back to back, and x-adjacent strips share their
  // (128 + disp_max)-column staging windows — both now hit in that XCD's L2 instead of going to HBM again.
  const int total = nbx * nby * 2 * n;
  int item = blockIdx.x;
  if (xcd_order) { const int per_xcd = (total + 7) / 8; item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3); }
  if (item >= total) return;
  const int side = item & 1;
  int rest = item >> 1;
  const int bx = rest % nbx; rest /= nbx;
  const int by = rest % nby;
  const int frame = rest / nby;
  const FrameInfo& fi = info[frame];      // by reference: a by-value copy of the 48-byte struct lands in scratch memory
  if (!fi.ok) return;
  const int W = dp.W, H = dp.H;
  const int tid = threadIdx.x;
  const int u0 = bx * kStripW, v0 = by * kTileH;
  const uint4* A = desc + (size_t)((side ? n : 0) + frame) * H * W;      // image being filled
  const uint4* B = desc + (size_t)((side ? 0 : n) + frame) * H * W;      // image searched
  const TriRec* R = recs + (size_t)(frame * 2 + side) * tri_cap;
  const int tiles_x = (W + kTileW - 1) / kTileW;
  const size_t bin_row = ((size_t)(frame * 2 + side) * nby + by) * tiles_x;
  const int x = tid & (kTileW - 1), r = (tid / kTileW) & (kTileH - 1), grp = tid / (kTileW * kTileH);   // grp 0: tiles 0,2; grp 1: tiles 1,3
  const int v = v0 + r;
  const int vr = max(min(v, H - 3), 2);                                    // :701
  const int nwords = (dp.disp_max >> 5) + 1;

  // ---- issue every global read of this thread up front, consume afterwards ----
  // (1) candidate lists: wave k < 4 owns tile k (count, then up to 16 entries of 13 dwords)
  const int lw = tid >> 6, lane = tid & 63;                  // waves 4..7 (second pixel pair) have no list duty
  const bool list_wave = lw < kStripTiles && bx * kStripTiles + lw < tiles_x;
  int cnt = 0, c16 = 0, myt = 0x7FFFFFFF;
  uint32_t lw0 = 0, lw1 = 0, lw2 = 0, lw3 = 0;
  if (list_wave) {
    const size_t bin = bin_row + bx * kStripTiles + lw;
    cnt = bin_count[bin];
    c16 = min(cnt, (int)kBinLds);
    const int words = c16 * kBinWords;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(bin_list + bin * kBinCap);
    if (lane < words) lw0 = src[lane];
    if (lane + 64 < words) lw1 = src[lane + 64];
    if (lane + 128 < words) lw2 = src[lane + 128];
    if (lane + 192 < words) lw3 = src[lane + 192];
    if (lane < c16) myt = (int)src[lane * kBinWords];
  }
  // (2) own descriptors and grid-cell candidate sets of the thread's two pixels
  uint4 a4[kPxPerThread];
  uint32_t cellw[kPxPerThread][kGridWords];
  const uint32_t* cells = gridbits + ((size_t)(frame * 2 + side) * dp.gw * dp.gh + (size_t)(min(v, H - 1) / dp.grid_size) * dp.gw) * kGridWords;
#pragma unroll
  for (int q = 0; q < kPxPerThread; q++) {
    const int u = min(u0 + (grp + 2 * q) * kTileW + x, W - 1);
    a4[q] = A[(size_t)vr * W + u];
    const uint32_t* cell = cells + (size_t)(u / dp.grid_size) * kGridWords;
#pragma unroll
    for (int w = 0; w < kGridWords; w++) cellw[q][w] = w < nwords ? cell[w] : 0u;
  }
  // (3) the other image's descriptors the strip can reach
  const int span = kStripW + dp.disp_max;                    // columns of B one strip row can reach
  const int base = side ? u0 : u0 - dp.disp_max;             // left image looks left (u-d), right image looks right (u+d)
#pragma unroll
  for (int rr = 0; rr < kTileH; rr++) {
    const uint4* src = B + (size_t)max(min(v0 + rr, H - 3), 2) * W;       // :701 row clamp
    for (int c = tid; c < span; c += kDenseThreads) {
      const int col = base + c;
      // the left image's window is stored mirrored, so that on both sides the descriptor matched at disparity d
      // sits d slots after the one matched at d = 0
      s_B[rr * span + (side ? c : span - 1 - c)] = (col >= 0 && col < W) ? src[col] : make_uint4(0, 0, 0, 0);
    }
  }
  // (1b) lists into LDS; rank the tile's triangles; fold the row masks into one cover word per pixel
  if (list_wave) {
    const int words = c16 * kBinWords;
    if (lane < words) s_list[lw][lane] = lw0;
    if (lane + 64 < words) s_list[lw][lane + 64] = lw1;
    if (lane + 128 < words) s_list[lw][lane + 128] = lw2;
    if (lane + 192 < words) s_list[lw][lane + 192] = lw3;
    // rank among the listed triangles (indices are distinct): a pixel's owner is the covering triangle with the
    // LARGEST index, i.e. the highest set bit of its cover word
    int rank = 0;
#pragma unroll
    for (int jj = 0; jj < kBinLds; jj++) { const int tj = __shfl(myt, jj); rank += (jj < c16 && tj < myt) ? 1 : 0; }
    if (lane < c16) s_slot[lw][rank] = (uint8_t)lane;
    // lane (x, half) accumulates rows half*4 .. half*4+3 of column x over all listed candidates
    const int xx = lane & (kTileW - 1), half = lane >> 5;
    unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    for (int c = 0; c < c16; c++) {
      const unsigned m = ((s_list[lw][c * kBinWords + 1 + (xx >> 2)] >> ((xx & 3) * 8)) & 0xFFu) >> (half * 4);
      const unsigned bit = 1u << __shfl(rank, c);
      w0 |= (m & 1u) ? bit : 0u; w1 |= (m & 2u) ? bit : 0u; w2 |= (m & 4u) ? bit : 0u; w3 |= (m & 8u) ? bit : 0u;
    }
    s_cover[lw][half * 4 + 0][xx] = (uint16_t)w0; s_cover[lw][half * 4 + 1][xx] = (uint16_t)w1;
    s_cover[lw][half * 4 + 2][xx] = (uint16_t)w2; s_cover[lw][half * 4 + 3][xx] = (uint16_t)w3;
  }
  if (lw < kStripTiles && lane == 0) s_cnt[lw] = cnt;
  __syncthreads();
  if (v >= H) return;

  // Bzero + u + d = descriptor of the column matched at disparity d for pixel column u (u - d left, u + d right)
  const uint4* Bzero = side ? s_B + r * span - base : s_B + r * span + span - 1 + base;
  constexpr unsigned kNoKey = 0xFFFFFFFFu;
  constexpr int kBias = 1 << 20;                             // makes cost + prior non-negative inside a key
  int16_t* out = raw + ((size_t)(frame * 2 + side) * H + v) * W;   // integer disparity, -1 no match, -10 not visited (:797-798)

#pragma unroll
  for (int q = 0; q < kPxPerThread; q++) {
    const int k = grp + 2 * q;
    const int u = u0 + k * kTileW + x;
    if (u >= W) break;
    // ---- which triangle owns this pixel: the last covering one in list order ----
    const int cnt = s_cnt[k];
    int t = -1; float pa = 0, pb = 0, pc = 0; bool valid = false;
    if (cnt <= kBinLds) {
      const unsigned cover = s_cover[k][r][x];
      if (cover) {
        const uint32_t* e = &s_list[k][s_slot[k][31 - __clz(cover)] * kBinWords];
        t = (int)e[0];
        pa = __uint_as_float(e[9]); pb = __uint_as_float(e[10]); pc = __uint_as_float(e[11]); valid = e[12] & 1u;
      }
    } else {
      if (cnt <= kBinCap) {                                  // long list: read it from global memory
        const BinEntry* list = bin_list + (bin_row + bx * kStripTiles + k) * kBinCap;
        for (int c = 0; c < cnt; c++) {
          const unsigned m = reinterpret_cast<const uint8_t*>(list[c].rows)[x];
          const int tc = list[c].t;
          if (((m >> r) & 1u) && tc > t) t = tc;
        }
      } else {                                               // overflowing tile: scan every triangle of this side
        for (int c = fi.ntri[side] - 1; c >= 0; c--) {
          const TriRec* q = R + c;
          if (tri_covers(q->Au, q->Bu, q->Cu, q->ACa, q->ACb, q->ABa, q->ABb, q->BCa, q->BCb, u, v)) { t = c; break; }
        }
      }
      if (t >= 0) { const TriRec* tr = R + t; pa = tr->pa; pb = tr->pb; pc = tr->pc; valid = tr->flags & 1; }
    }
    float result = -10.0f;                                                 // :797-798
    const uint4 a = a4[q];
    if (t >= 0 && u >= 2 && u < W - 2 && texture16(a) >= dp.match_texture) {   // :697, :715-719
      const int d_plane = (int)__fadd_rn(__fadd_rn(__fmul_rn(pa, (float)u), __fmul_rn(pb, (float)v)), pc);   // :722
      const int lo = max(d_plane - dp.radius, 0), hi = min(d_plane + dp.radius, dp.disp_max);                // :723-724
      // disparities whose warped column stays inside [2, W-2) (:746, :753, :764, :771)
      const int dmax_ok = side ? min(dp.disp_max, W - 3 - u) : min(dp.disp_max, u - 2);
      // The reference keeps (min_val, min_d) with a strict `<` over candidates in evaluation order (:735-756).  Within
      // one phase disparities ascend, so the minimum of the keys (cost << 8 | d) is the same choice; the plane phase
      // comes second and only wins with a strictly smaller cost.
      const uint4* Bu = side ? Bzero + u : Bzero - u;
      unsigned best1 = kNoKey, best2 = kNoKey;
#pragma unroll
      for (int w = 0; w < kGridWords; w++) {                               // grid candidates outside the plane range (:742-750)
        if (w >= nwords) break;
        uint32_t bits = cellw[q][w] & range_mask(0, dmax_ok, w) & ~range_mask(lo, hi, w);
        while (bits) {
          const int d = (w << 5) + __builtin_ctz(bits);
          bits &= bits - 1;
          const unsigned val = sad16_acc(a, Bu[d], (unsigned)kBias);      // the bias rides in the accumulator operand
          best1 = min(best1, (val << 8) | (unsigned)d);
        }
      }
      const int phi = min(hi, dmax_ok);
      const unsigned prior_on = valid ? 0xFFFFFFFFu : 0u;
#pragma unroll
      for (int off = -7; off <= 7; off++) {                                // plane neighbourhood with prior (:751-756); radius <= 7
        if ((off < 0 ? -off : off) > dp.radius) continue;                  // uniform
        const int d = d_plane + off;
        if (d >= lo && d <= phi) {
          const unsigned val = sad16_acc(a, Bu[d], (unsigned)kBias + ((unsigned)dp.P[off < 0 ? -off : off] & prior_on));
          best2 = min(best2, (val << 8) | (unsigned)d);
        }
      }
      const unsigned best = (best2 >> 8) < (best1 >> 8) ? best2 : best1;
      const int best_d = best == kNoKey ? -1 : (int)(best & 255u);
      result = best_d >= 0 ? (float)best_d : -1.0f;                        // :778-779
    }
    out[u] = (int16_t)result;
  }
}

// ------------------------------------------------------------------------------------------------
// Helpers of the LDS-window matchers (k_dense_row)
enum { kDense2Slack = 16, kCellBias = 8192, kCellPriorMax = 8000, kCellInvalid = 0x60000000 };
typedef unsigned int jn_u32x4 __attribute__((ext_vector_type(4)));
// LDS byte address of a pointer into __shared__ memory, and a 16-byte read at such an address
typedef __attribute__((address_space(3))) const jn_u32x4 jn_lds_u4;
DEV uint32_t lds_addr(const uint4* p) { return (uint32_t)(uintptr_t)(jn_lds_u4*)p; }
DEV uint4 lds_read16(uint32_t a) { const jn_u32x4 t = *(jn_lds_u4*)a; return make_uint4(t.x, t.y, t.z, t.w); }
typedef __attribute__((address_space(3))) jn_u32x4 jn_lds_u4w;
DEV void lds_write16(uint32_t a, const uint4& d) { jn_u32x4 t; t.x = d.x; t.y = d.y; t.z = d.z; t.w = d.w; *(jn_lds_u4w*)a = t; }
// min(max(x, 0), 63) as ONE instruction (the compiler splits the clamp around the addition that feeds it: max, add, min)
DEV int clamp_0_63(int x) { int r; asm("v_med3_i32 %0, %1, 0, 63" : "=v"(r) : "v"(x)); return r; }
// One pixel of the dense matching behind the ownership lookup (elas.cpp:722-779), used by k_dense_row: `a` the pixel's own
// descriptor, Bu[d] the descriptor of the column matched at disparity d in the LDS window (u + d in the right image's window, u - d in the
// mirrored left one), cw the candidate set of the pixel's grid cell.  Every lane of the wave must call it (one ballot inside).
// BORDER: the strip's disparity ranges can leave the image (the first strips of a left image, the last ones of a right image): only then do the
// grid candidates need the range mask — a template argument because the compiler turns a uniform `if` around six instructions per word into
// selects, which every strip then pays (48 vector instructions a wave row, 9 % of the kernel's).
template <int NW, bool BORDER>
DEV int match_pixel(const DevParams& dp, const uint4& a, bool elig, int d_plane, bool valid, int u, int side, const uint4* Bu,
                    const uint32_t (&cw)[NW], int dbg) {
  constexpr unsigned kNoKey = 0xFFFFFFFFu;
  const int radius = dp.radius, W = dp.W;
  const int lo = max(d_plane - radius, 0), hi = min(d_plane + radius, dp.disp_max);                      // :723-724
  // disparities whose warped column stays inside [2, W-2) (:746, :753, :764, :771)
  // (a strip without BORDER holds no pixel whose range leaves the image: dmax_ok is disp_max there, phi is hi)
  const int dmax_ok = !BORDER ? dp.disp_max : side ? min(dp.disp_max, W - 3 - u) : min(dp.disp_max, u - 2);
  const int dlow = d_plane - radius;
  const int phi = BORDER ? min(hi, dmax_ok) : hi;

  // ---- grid candidates outside the plane range (:742-750): per-lane bit scan, one (different) disparity per lane and round ----
  // Keys here are (SAD << 16) + LDS byte address of the candidate's descriptor: the address is what the read needs anyway, it orders
  // like d, and d = (address - address of Bu[0]) / 16 is recovered once at the end (the bias the plane keys carry is added there too).
  unsigned best1 = kNoKey;
  const uint32_t bu_a = lds_addr(Bu);
  if (elig && !(dbg & 1)) {
    // The plane range [lo, hi] (at most 15 wide) sits as ones in bits 16.. of T; word w of the exclusion mask is the HIGH half of
    // T << (lo + 16 - 32 w) with the shift clamped to [0, 63]: below 0 and from 48 up nothing of T lands in bits 32..63, in between it
    // is M << (lo - 32 w) or the spill M >> (32 w - lo) of a range that starts in the word below.  (hi < lo: T = 0.)
    const unsigned long long T = (unsigned long long)(((1u << max(hi - lo + 1, 0)) - 1u) << 16);
#pragma unroll
    for (int w = 0; w < NW; w++) {
#ifndef JN_AB_NO_EMPTY_WORD_SKIP
      if (__ballot(cw[w] != 0u) == 0ull) continue;          // no lane's cell holds a candidate in this word: six instructions saved for two spent
#endif
      const uint32_t excl = (uint32_t)((T << clamp_0_63(lo + 16 - 32 * w)) >> 32);
      uint32_t bits = cw[w] & ~excl;
      if (BORDER) bits &= range_mask(0, dmax_ok, w);
      while (bits) {
        const uint32_t ad = bu_a + (uint32_t)((w << 5) + __builtin_ctz(bits)) * 16u;
        bits &= bits - 1;
        best1 = min(best1, sadhi16(a, lds_read16(ad), ad));
      }
    }
  }

  // ---- plane neighbourhood with prior (:751-756): d = dlow + k, k = 0 .. 2r ----
  unsigned best2 = kNoKey;
  if (dbg & 2) {} else
  if (radius == 2) {
    // address window clamped so that it stays inside the LDS block (+- kDense2Slack); a clamped window holds no valid d
    const uint4* Bw = Bu + max(min(dlow, dp.disp_max), -2 * radius);
    uint4 nb[5];
#pragma unroll
    for (int kk = 0; kk < 5; kk++) nb[kk] = Bw[kk];
    const bool all_in = dlow >= 0 && dlow + 4 <= phi && valid;          // whole neighbourhood valid, prior on
    if (__ballot(elig && !all_in) == 0ull) {
      // every lane of the wave: five valid candidates with the prior — no per-candidate tests (scalar key bases)
      const unsigned i0 = ((unsigned)(kCellBias + dp.P[2]) << 16), i1 = ((unsigned)(kCellBias + dp.P[1]) << 16) + 1u,
                     i2 = ((unsigned)(kCellBias + dp.P[0]) << 16) + 2u, i3 = ((unsigned)(kCellBias + dp.P[1]) << 16) + 3u,
                     i4 = ((unsigned)(kCellBias + dp.P[2]) << 16) + 4u;
      // keys relative to dlow (scalar bases), dlow added to the minimum: 0 <= dlow and dlow + 4 <= 255, no carry into the cost
      const unsigned k0 = sadhi16(a, nb[0], i0), k1 = sadhi16(a, nb[1], i1), k2 = sadhi16(a, nb[2], i2),
                     k3 = sadhi16(a, nb[3], i3), k4 = sadhi16(a, nb[4], i4);
      best2 = min(min(min(k0, k1), min(k2, k3)), k4) + (unsigned)dlow;
    } else {
      const unsigned prior_on = valid ? 0xFFFFFFFFu : 0u;
#pragma unroll
      for (int kk = 0; kk < 5; kk++) {
        const int d = dlow + kk;
        const unsigned init = (d >= 0 && d <= phi) ? (((unsigned)kCellBias << 16) + (((unsigned)dp.P[kk < 2 ? 2 - kk : kk - 2] << 16) & prior_on) + (unsigned)d)
                                                   : (unsigned)kCellInvalid;
        best2 = min(best2, sadhi16(a, nb[kk], init));
      }
      if (best2 >= (unsigned)kCellInvalid) best2 = kNoKey;
    }
  } else if (elig) {                                                     // other radii: one read per candidate
#pragma unroll
    for (int off = -7; off <= 7; off++) {
      if ((off < 0 ? -off : off) > radius) continue;                     // uniform
      const int d = d_plane + off;
      if (d >= lo && d <= phi) {
        const unsigned init = (valid ? (unsigned)(kCellBias + dp.P[off < 0 ? -off : off]) << 16 : (unsigned)kCellBias << 16) + (unsigned)d;
        best2 = min(best2, sadhi16(a, Bu[d], init));
      }
    }
  }
  // the plane phase wins only with a strictly smaller cost; a missing grid key (0xFFFF + bias) loses to every plane key, a missing plane
  // key (cost field 0xFFFF) to every grid key (SAD <= 4080); both missing: the plane branch answers -1
  const bool plane_wins = (best2 >> 16) < (best1 >> 16) + (unsigned)kCellBias;
  int result = -10;                                                      // :797-798
  if (elig) result = plane_wins ? (best2 == kNoKey ? -1 : (int)(best2 & 255u)) : (int)(((best1 & 0xFFFFu) - bu_a) >> 4);   // :778-779
  return result;
}

// ------------------------------------------------------------------------------------------------
// The dense matching of the plane data flow (round 5): k_owner + k_dense_row.  They replace k_dense2 (rounds 2-4; git history, DESIGN_HISTORY.md):
// a workgroup per 128 x 8 strip with a barrier, whose list handling they take out of the matcher.
//
// k_dense2's workgroup spent half its life in a prologue: four of its eight waves rank a tile's triangle list and fold the row masks into cover
// words through a chain of LDS round trips while the others wait at the barrier.  None of that needs the descriptors.  k_owner does it
// ahead, one wave per 32x8 tile at full occupancy, and leaves ONE 16-bit word per pixel (in the matcher's own output image, which the
// matcher overwrites):  bit 15 = a triangle covers the pixel (the LAST covering one in list order counts, as in k_dense2), bit 14 = its
// plane prior is valid (elas.cpp:872), bits 0..13 = d_plane + 32 (elas.cpp:722), clamped to [-32, 8000] — every use of d_plane is a
// comparison against a range inside [-radius - 1, disp_max + radius + 1], so the clamp changes nothing.
enum { kOwnerBias = 32, kOwnerMax = 8000 };
// The profiling switches (JN_OWNER_DBG, JN_DENSE_DBG: results WRONG) and the test hooks that send ordinary lists through the long-list routes
// (JN_OWNER_FAST_MAX, JN_OWNER_SCAN_FROM: results unchanged) exist in the hooks build only (hooks.h); the release kernels have neither the
// arguments nor the branches.
#ifdef JN_HOOKS
#define JN_OWNER_HOOK_PARAMS , int dbg, int fast_max, int scan_from
#define JN_OWNER_HOOK_LOCALS
#define JN_DENSE_HOOK_PARAMS , int dbg
#define JN_DENSE_HOOK_LOCALS
#define JN_DENSE_HOOK_PULL , "s"(dbg)
#else
#define JN_OWNER_HOOK_PARAMS
#define JN_OWNER_HOOK_LOCALS constexpr int dbg = 0, fast_max = kBinLds, scan_from = kBinCap;
#define JN_DENSE_HOOK_PARAMS
#define JN_DENSE_HOOK_LOCALS constexpr int dbg = 0;
#define JN_DENSE_HOOK_PULL
#endif
__global__ void __launch_bounds__(256) k_owner(DevParams dp, const FrameInfo* __restrict__ info, const TriRec* __restrict__ recs, int tri_cap,
                                               const int32_t* __restrict__ bin_count, const BinEntry* __restrict__ bin_list, uint16_t* __restrict__ owner JN_OWNER_HOOK_PARAMS) {
  JN_OWNER_HOOK_LOCALS
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tiles_x = (dp.W + kTileW - 1) / kTileW, tiles_y = (dp.H + kTileH - 1) / kTileH;
  const int tx = blockIdx.x * 4 + wave, ty = blockIdx.y, fs = blockIdx.z, frame = fs >> 1, side = fs & 1;
  if (tx >= tiles_x) return;
  if (dbg & 4) return;
  const FrameInfo& fi = info[frame];
  const size_t bin = ((size_t)fs * tiles_y + ty) * tiles_x + tx;
  // What this kernel costs is its vector-memory INSTRUCTIONS (a CU issues one per ~8 cycles whatever it moves; 230 k tiles), so a tile is
  // seven of them: frame flag, list length, the list's first 16 entries as four coalesced dword loads (lane l holds words l, l+64, l+128,
  // l+192 — they always exist: kBinCap entries per tile), one 8-byte store per lane.  An entry's wave-uniform words (triangle, plane, flag)
  // are v_readlane'd out of those registers, its row-mask word for the lane's columns comes by ds_bpermute.
  // Lane (r, g) = (lane >> 3, lane & 7): row r of the tile, columns 4g .. 4g+3 — byte j of mask word 1 + g holds column 4g + j's row bits.
  const int r = lane >> 3, g = lane & 7;
  const int u = tx * kTileW + 4 * g, v = ty * kTileH + r;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(bin_list + bin * kBinCap);
  __shared__ uint32_t s_list[4][kBinLds * kBinWords];        // the list's first 16 entries, per wave
  __shared__ uint4 s_plane[4][kBinLds];                      // their planes by rank
  int vzero = 0;
  asm volatile("" : "+v"(vzero));
  const int ok_v = (&fi.ok)[vzero], cnt_v = bin_count[bin + vzero];
  uint32_t lw[4];
#pragma unroll
  for (int i = 0; i < 4; i++) lw[i] = (dbg & 1) ? (uint32_t)(lane * 77 + i) : src[lane + 64 * i];
  int best_t[4] = {-1, -1, -1, -1};
  float pa[4], pb[4], pc[4]; bool valid[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { pa[j] = pb[j] = pc[j] = 0.f; valid[j] = false; }
  // (the frame flag is looked at last: an early return on it would make the compiler sink every load above behind that wait.  A frame
  // that failed has stale but addressable lists; its overflow scan is skipped.)
  const int cnt = __builtin_amdgcn_readfirstlane(cnt_v);
  const int cnt_l = min(cnt, (int)kBinCap);
  if (cnt_l <= fast_max) {                                   // (fast_max = kBinLds; tests lower it to send ordinary lists through the general form)
    // The usual case (mean 7 entries, 17 the most seen at 720p): what the kernel costs here is its VECTOR instructions (230 k waves x ~400
    // in the general form below = the 0.17 ms it took), so ownership is resolved by cover words (k_dense2's scheme) — the entries are RANKED by triangle number,
    // an entry's hits on the lane's four pixels ((mask word >> row) & 0x01010101: bit 0 of byte j = column 4g + j) are or-ed into the
    // pixels' cover words at the entry's rank (three instructions per entry for four pixels), and a pixel's owner is its highest set bit.
#pragma unroll
    for (int i = 0; i < 4; i++) s_list[wave][lane + 64 * i] = lw[i];
    const int myt = lane < cnt_l ? (int)s_list[wave][(lane & (kBinLds - 1)) * kBinWords] : 0x7FFFFFFF;
    int rank = 0;
    for (int jj = 0; jj < cnt_l; jj++) rank += __builtin_amdgcn_readlane(myt, jj) < myt ? 1 : 0;
    if (lane < cnt_l) {
      const uint32_t* e = &s_list[wave][lane * kBinWords];
      s_plane[wave][rank] = make_uint4(e[9], e[10], e[11], e[12]);
    }
    unsigned acc_lo = 0, acc_hi = 0;                         // byte j: ranks 0..7 / 8..15 that cover pixel (4g + j, r)
    const uint32_t* mrow = &s_list[wave][1 + g];
    for (int c0 = 0; c0 < cnt_l; c0 += 4) {                  // four mask reads in flight (c0 + 3 <= 15: inside the 16 entries)
      unsigned mk[4];
#pragma unroll
      for (int k = 0; k < 4; k++) mk[k] = mrow[(c0 + k) * kBinWords];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (c0 + k >= cnt_l) break;                          // uniform
        const unsigned hit = (mk[k] >> r) & 0x01010101u;
        const int rk = __builtin_amdgcn_readlane(rank, c0 + k);
        if (rk < 8) acc_lo |= hit << rk; else acc_hi |= hit << (rk - 8);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned cover = ((acc_lo >> (8 * j)) & 0xFFu) | (((acc_hi >> (8 * j)) & 0xFFu) << 8);
      if (cover) {
        const uint4 pl = s_plane[wave][31 - __clz(cover)];
        best_t[j] = 0; pa[j] = __uint_as_float(pl.x); pb[j] = __uint_as_float(pl.y); pc[j] = __uint_as_float(pl.z); valid[j] = pl.w & 1u;
      }
    }
  } else
  for (int c0 = 0;;) {                                       // longer lists: the largest covering triangle number entry by entry, sixteen entries a round
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (c0 + k >= cnt_l) break;                            // uniform
      const uint32_t w = lw[k >> 2];
      const int base_lane = 16 * (k & 3);
      const int tc = __builtin_amdgcn_readlane((int)w, base_lane);
      const float epa = __uint_as_float(__builtin_amdgcn_readlane((int)w, base_lane + 9)), epb = __uint_as_float(__builtin_amdgcn_readlane((int)w, base_lane + 10)),
                  epc = __uint_as_float(__builtin_amdgcn_readlane((int)w, base_lane + 11));
      const bool ev = __builtin_amdgcn_readlane((int)w, base_lane + 12) & 1;
      const uint32_t mw = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (base_lane + 1 + g), (int)w);
      const uint32_t hit = (mw >> r) & 0x01010101u;          // byte j: the triangle covers (column 4g + j, row r)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool take = ((hit >> (8 * j)) & 1u) && tc > best_t[j];               // the covering triangle with the LARGEST index
        best_t[j] = take ? tc : best_t[j]; pa[j] = take ? epa : pa[j]; pb[j] = take ? epb : pb[j]; pc[j] = take ? epc : pc[j]; valid[j] = take ? ev : valid[j];
      }
    }
    c0 += 16;
    if (c0 >= cnt_l) break;
#pragma unroll
    for (int i = 0; i < 4; i++) lw[i] = src[c0 * kBinWords + lane + 64 * i];       // the next 16 entries (c0 + 16 <= kBinCap)
  }
  const bool ok = __builtin_amdgcn_readfirstlane(ok_v);
  if (cnt > scan_from && ok) {                               // (scan_from = kBinCap; tests lower it)  overflowing tile: scan every triangle of this side, last one first (overrides the list's answer)
    const TriRec* R = recs + (size_t)fs * tri_cap;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      best_t[j] = -1; pa[j] = pb[j] = pc[j] = 0.f; valid[j] = false;
      for (int c = fi.ntri[side] - 1; c >= 0; c--) {
        const TriRec* q = R + c;
        if (tri_covers(q->Au, q->Bu, q->Cu, q->ACa, q->ACb, q->ABa, q->ABb, q->BCa, q->BCb, u + j, v)) { best_t[j] = c; break; }
      }
      if (best_t[j] >= 0) { const TriRec* tr = R + best_t[j]; pa[j] = tr->pa; pb[j] = tr->pb; pc[j] = tr->pc; valid[j] = tr->flags & 1; }
    }
  }
  if (!ok || v >= dp.H || u >= dp.W) return;
  if ((dbg & 2) && best_t[0] != 12345678) return;
  unsigned code[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int d_plane = (int)__fadd_rn(__fadd_rn(__fmul_rn(pa[j], (float)(u + j)), __fmul_rn(pb[j], (float)v)), pc[j]);   // :722
    code[j] = (best_t[j] >= 0 ? 0x8000u : 0u) | (valid[j] ? 0x4000u : 0u) | (unsigned)(min(max(d_plane, -kOwnerBias), (int)kOwnerMax) + kOwnerBias);
  }
  uint16_t* o = owner + ((size_t)fs * dp.H + v) * dp.W + u;
  if ((dp.W & 3) == 0) *reinterpret_cast<uint2*>(o) = make_uint2(code[0] | (code[1] << 16), code[2] | (code[3] << 16));   // u + 3 < W, 8-byte aligned
  else
#pragma unroll
    for (int j = 0; j < 4; j++) if (u + j < dp.W) o[j] = (uint16_t)code[j];
}

// k_dense_row: one WAVE per strip row of 128 pixels, a lane owns two neighbouring pixels; a workgroup is eight such rows (the 128 x 8 strip
// of one side in the XCD-aware order k_dense describes) only because they share plane rows in the CU's cache — there is NO barrier and
// no LDS shared between waves: wave w assembles window row w and is the only one that reads it.  A wave's life: pull its arguments, issue every
// load (the eight plane rows of its own 128 pixels, the eight of its window, the two ownership words, the grid cell), assemble (v_perm), store
// the window row, match.  Waves of a CU are at different points of that, so the memory round trips of one hide behind the matching of
// the others without software pipelining.
//  * Every plane dword is fetched by ONE lane (4-byte loads) and the second dword a lane needs comes from its neighbour by DPP.  [Measured
//    against 8-byte loads per lane, every dword fetched twice: the same kernel time — what the loads cost is their number, 16 a wave, ~8
//    cycles of the CU's vector-memory path each = 0.1 ms of the kernel — but fewer registers and half the bytes through the cache.]
//  * own descriptors: the two pixels 2l, 2l+1 of lane l lie in the aligned group 4 (l >> 1): even lanes fetch the group's first dword, odd
//    lanes the one behind it, a quad_perm swap hands each the other, and one v_perm with a per-lane selector shifts the odd lanes' view
//    down by two columns; then the C = 0 and C = 1 picks of desc_from_rows.
//  * window: lane l assembles the image-aligned group of columns 4 (g0 + l) .. + 3: its dword, the next lane's (wave_shl:1; behind the last
//    lane: a scalar load), 32 v_perm, four 16-byte LDS stores 64 bytes apart (4-way bank conflicts; slots without them were timed: -2 %).
// Tried and dropped (profiles/r05_dense_forms.txt): k_dense2 reading the planes (own descriptors per thread: 0.61 ms; through a wave-private
// corner of the window block: 0.63 ms — the barrier and the list waves stay); the strip's plane rows fetched once per workgroup into LDS
// behind one barrier (12 loads a workgroup instead of 16 a wave: 0.57 ms — the barrier is back, and the LDS round trip costs what the loads did).
template <int NW>
__global__ void __launch_bounds__(kDenseThreads) k_dense_row(DevParams dp, int n, const FrameInfo* __restrict__ info, const uint32_t* __restrict__ gridbits,
                                                             const uint8_t* __restrict__ planes, int Wp, int16_t* __restrict__ raw, int nbx, int nby,
                                                             int xcd_order, uint32_t nbx_magic, uint32_t nby_magic JN_DENSE_HOOK_PARAMS) {
  JN_DENSE_HOOK_LOCALS
  extern __shared__ uint4 s_Bx[];                            // kDense2Slack + [kTileH][kStripW + disp_max] + kDense2Slack
  // Every kernel argument the wave will need is pulled into scalar registers HERE, behind one wait: left alone the compiler fetches
  // them one basic block at a time, each fetch a scalar-cache round trip in series with the loads below.
  asm volatile("" :: "s"(info), "s"(gridbits), "s"(planes), "s"(Wp), "s"(raw),
               "s"(dp.W), "s"(dp.H), "s"(dp.disp_max), "s"(dp.gw), "s"(dp.gh), "s"(dp.grid_magic), "s"(dp.radius), "s"(dp.match_texture),
               "s"(dp.P[0]), "s"(dp.P[1]), "s"(dp.P[2]), "s"(n), "s"(nbx), "s"(nby), "s"(nbx_magic), "s"(nby_magic) JN_DENSE_HOOK_PULL);
  const int total = nbx * nby * 2 * n;
  int item = blockIdx.x;
  if (xcd_order) { const int per_xcd = (total + 7) / 8; item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3); }   // see k_dense
  if (item >= total) return;
  if (dbg & 64) return;                                     // JN_DENSE_DBG profiling switches (results are then WRONG): 64 = empty blocks,
                                                            // 8 / 16 = no window / own plane loads, 4 = stop after the prologue, 1 / 2 = no grid / plane candidates
  // item -> (side, bx, by, frame): the two divisions by multiply-high with the host's ceil(2^32 / n) (exact while x n < 2^32, which
  // launch_dense() checks) — the compiler's division sequence is ~28 dependent instructions each, in front of the wave's first load
  const int side = item & 1;
  const unsigned rest = (unsigned)item >> 1;
  const unsigned rest2 = nbx == 1 ? rest : __umulhi(rest, nbx_magic);     // ceil(2^32 / 1) does not fit
  const int bx = (int)(rest - rest2 * (unsigned)nbx);
  const int frame = (int)(nby == 1 ? rest2 : __umulhi(rest2, nby_magic));
  const int by = (int)(rest2 - (unsigned)frame * (unsigned)nby);
  const int W = dp.W, H = dp.H;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int u0 = bx * kStripW, v = by * kTileH + wave;
  if (v >= H) return;                                        // (no barrier below: a wave may leave)
  const int fs = frame * 2 + side;
  const int imgA = (side ? n : 0) + frame, imgB = (side ? 0 : n) + frame;                // image being filled, image searched
  const int vr = max(min(v, H - 3), 2);                      // :701
  const int ok = info[frame].ok;                             // (scalar load; looked at after the window row has been stored)

  // ---- every load of the wave, up front ----
  const int up = u0 + 2 * lane;                              // the lane's pixels: up, up + 1
  const int span = kStripW + dp.disp_max;
  const int base = side ? u0 : u0 - dp.disp_max;             // left image looks left (u - d), right image looks right (u + d)
  const int g0 = base >> 2;                                  // arithmetic shift: floor for the negative bases of the left image's first strips
  constexpr int kPasses = NW == 4 ? 1 : 2;                   // 128 + roundup4(disp_max) <= 256 columns are at most 64 groups for NW = 4 (u0 is a multiple of 128), 96 for NW = 8
  const uint8_t* duA = planes + (size_t)(imgA * 2) * H * Wp;
  const uint8_t* duB = planes + (size_t)(imgB * 2) * H * Wp;
  const int dv_off = H * Wp;                                 // the dv plane follows the du plane
  // (buffer loads: the lane's byte offset in the vector operand, the row in the scalar one — no 64-bit address arithmetic per load)
  uint32_t oL[8];
  if (!(dbg & 16)) {
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(duA), 0, 2 * H * Wp, 0x00020000);
    const int o = (vr - 2) * Wp + min((up & ~3) + 4 * (lane & 1), Wp - 4);
#pragma unroll
    for (int k = 0; k < 5; k++) oL[k] = __builtin_amdgcn_raw_buffer_load_b32(rA, o, k * Wp, 0);
#pragma unroll
    for (int k = 0; k < 3; k++) oL[5 + k] = __builtin_amdgcn_raw_buffer_load_b32(rA, o, dv_off + (k + 1) * Wp, 0);
  }
  uint32_t wL[kPasses][8], wS[8];
  if (!(dbg & 8)) {
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(duB), 0, 2 * H * Wp, 0x00020000);
#pragma unroll
    for (int i = 0; i < kPasses; i++) {
      // groups outside the image: any in-range address — their descriptors are zeroed below, and every dword a VALID descriptor taps lies in [0, W + 4)
      const int o = (vr - 2) * Wp + max(min(4 * (g0 + lane + 64 * i), Wp - 4), 0);
#pragma unroll
      for (int k = 0; k < 5; k++) wL[i][k] = __builtin_amdgcn_raw_buffer_load_b32(rB, o, k * Wp, 0);
#pragma unroll
      for (int k = 0; k < 3; k++) wL[i][5 + k] = __builtin_amdgcn_raw_buffer_load_b32(rB, o, dv_off + (k + 1) * Wp, 0);
    }
    const uint8_t* ps = duB + (size_t)(vr - 2) * Wp + max(min(4 * (g0 + 64 * kPasses), Wp - 4), 0);       // wave-uniform address: scalar loads
#pragma unroll
    for (int k = 0; k < 5; k++) wS[k] = *reinterpret_cast<const uint32_t*>(ps + (size_t)k * Wp);
#pragma unroll
    for (int k = 0; k < 3; k++) wS[5 + k] = *reinterpret_cast<const uint32_t*>(ps + dv_off + (size_t)(k + 1) * Wp);
  }
  // ownership words of the two pixels (k_owner left them in the output image), one aligned dword when the rows are dword-aligned
  uint16_t* orow = reinterpret_cast<uint16_t*>(raw) + ((size_t)fs * H + v) * W;
  const bool pair_io = (W & 1) == 0;                         // uniform
  unsigned ow;
  if (pair_io) ow = up < W ? *reinterpret_cast<const uint32_t*>(orow + up) : 0u;
  else ow = (up < W ? (unsigned)orow[up] : 0u) | (up + 1 < W ? (unsigned)orow[up + 1] << 16 : 0u);
  // grid cells of the two pixels: the same one unless a cell boundary falls between them (never for an even grid size).  x / G as one
  // multiply-high (dense_row_applies() guarantees G >= 8, so the magic constant is exact for any pixel index)
  uint32_t cellw[2][NW];
  const uint32_t* cells = gridbits + ((size_t)fs * dp.gw * dp.gh + (size_t)__umulhi((unsigned)v, dp.grid_magic) * dp.gw) * kGridWords;
  const unsigned c0 = __umulhi((unsigned)min(up, W - 1), dp.grid_magic), c1 = __umulhi((unsigned)min(up + 1, W - 1), dp.grid_magic);
#pragma unroll
  for (int w = 0; w < NW; w++) cellw[0][w] = cells[(size_t)c0 * kGridWords + w];
  const bool two_cells = __any(c1 != c0);
  if (two_cells) {
#pragma unroll
    for (int w = 0; w < NW; w++) cellw[1][w] = cells[(size_t)c1 * kGridWords + w];
  } else {
#pragma unroll
    for (int w = 0; w < NW; w++) cellw[1][w] = cellw[0][w];
  }

  // ---- own descriptors ----
  uint4 a4[2];
  if (dbg & 16) { a4[0] = make_uint4(up, up, up, up); a4[1] = a4[0]; }
  else {
    // L = the dword this lane fetched, P = its pair partner's (quad_perm [1,0,3,2]).  Even lanes hold (lo, hi) = (L, P) and want the bytes from
    // 0 on; odd lanes hold (P, L) and want them from byte 2 on (columns 2, 3 of the group): one v_perm with a per-lane selector each for
    // the first four bytes (W) and the ones behind them (W2).
    const bool odd = lane & 1;
    const uint32_t selW = odd ? 0x05040302u : 0x07060504u, selW2 = odd ? 0x0c0c0706u : 0x03020100u;
    PlaneRows po;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t L = oL[k], P = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)L, 0xB1, 0xF, 0xF, false);
      const uint32_t Wd = __builtin_amdgcn_perm(L, P, selW), W2 = __builtin_amdgcn_perm(L, P, selW2);
      if (k < 5) { po.al[k] = Wd; po.ah[k] = W2; } else { po.bl[k - 5] = Wd; po.bh[k - 5] = W2; }
    }
    a4[0] = desc_from_rows<0>(po); a4[1] = desc_from_rows<1>(po);
    const bool rz = vr < 3 || vr > H - 4;
    if (rz || up < 3 || up + 1 > W - 4) {                    // (only the image's rim: zeros outside the descriptor image)
      if (rz || up < 3 || up > W - 4) a4[0] = make_uint4(0, 0, 0, 0);
      if (rz || up + 1 < 3 || up + 1 > W - 4) a4[1] = make_uint4(0, 0, 0, 0);
    }
  }
  // ---- the window row.  The left image's window is stored mirrored, so that on both sides the descriptor matched at disparity d sits d slots
  // after the one matched at d = 0 ----
  uint4* s_B = s_Bx + kDense2Slack;
  uint4* dst = s_B + wave * span;
  {
    const bool row_zero = vr < 3 || vr > H - 4;              // rows 2 and H-3 hold no descriptors: zeros
    const bool rim = row_zero || base < 3 || base + span - 1 > W - 4;      // only strips at the image's rim hold columns outside [3, W-4]
    // the slot of column col + k is side ? c + k : span - 1 - (c + k); as a byte address: one xor-add for k = 0, a scalar step from there
    const int mirror = side - 1, step = side ? 16 : -16;     // (scalars)
    const uint32_t dst_a = lds_addr(dst);
    // RIM as a template argument: left as a uniform `if` around the zeroing, the compiler turns it into four selects and two compares per
    // descriptor, which every strip then pays (the same finding as match_pixel's BORDER)
    PlaneRows pw[kPasses];                                    // (assembled once, in front of the two forms of the stores)
#pragma unroll
    for (int i = 0; i < kPasses; i++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        // the dword behind this lane's: the next lane's (wave_shl:1); the last lane keeps `old` = the next pass's first lane or the scalar load
        uint32_t lo = wL[i][k], behind = (i + 1 < kPasses) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)wL[kPasses - 1][k]) : wS[k];
        if (dbg & 8) { lo = 4 * (g0 + lane + 64 * i) + k; behind = lo - k; }
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)behind, (int)lo, 0x130, 0xF, 0xF, false);
        if (k < 5) { pw[i].al[k] = lo; pw[i].ah[k] = hi; } else { pw[i].bl[k - 5] = lo; pw[i].bh[k - 5] = hi; }
      }
    }
    auto window = [&](auto rim_tag) {
      constexpr bool RIM = decltype(rim_tag)::value;
#pragma unroll
      for (int i = 0; i < kPasses; i++) {
        const int col = 4 * (g0 + lane + 64 * i), c = col - base;           // window slot of the group's first column
        uint32_t a = dst_a + ((uint32_t)((c ^ mirror) + (mirror & span)) << 4);
        auto put = [&](int k, uint4 d) {
          if (RIM && (row_zero || col + k < 3 || col + k > W - 4)) d = make_uint4(0, 0, 0, 0);
          if ((unsigned)(c + k) < (unsigned)span) lds_write16(a, d);
          a += (uint32_t)step;
        };
        put(0, desc_from_rows<0>(pw[i])); put(1, desc_from_rows<1>(pw[i])); put(2, desc_from_rows<2>(pw[i])); put(3, desc_from_rows<3>(pw[i]));
      }
    };
    if (rim) window(std::true_type{}); else window(std::false_type{});
  }
  if (!ok) return;                                           // uniform over the wave (and the block)
  __builtin_amdgcn_wave_barrier();                           // LDS operations of one wave execute in order: the stores above precede the reads below
  if (dbg & 4) return;

  // ---- matching ----
  // the strip needs the image-border mask only if some pixel's disparity range can leave [2, W-2)
  const bool border = side ? (u0 + kStripW - 1 + dp.disp_max > W - 3) : (u0 - dp.disp_max < 2);
  int res[2];
  auto match = [&](auto border_tag) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int u = up + q;
      const bool inw = u < W;
      const unsigned code = (ow >> (16 * q)) & 0xFFFFu;
      const bool valid = (code >> 14) & 1u;
      const int d_plane = (int)(code & 0x3FFFu) - kOwnerBias;
      const uint4 a = a4[q];
      const bool elig = inw && (code >> 15) && u >= 2 && u < W - 2 && texture16(a) >= dp.match_texture;   // :697, :715-719
      const int uc = min(u, W - 1);
      const uint4* Bu = dst + (side ? (uc - base) : (span - 1 + base - uc));
      res[q] = match_pixel<NW, decltype(border_tag)::value>(dp, a, elig, d_plane, valid, u, side, Bu, cellw[q], dbg);
    }
  };
  if (border) match(std::true_type{}); else match(std::false_type{});
  int16_t* out = raw + ((size_t)fs * H + v) * W;             // integer disparity, -1 no match, -10 not visited (:797-798)
  if (pair_io) { if (up < W) *reinterpret_cast<uint32_t*>(out + up) = ((unsigned)res[0] & 0xFFFFu) | ((unsigned)res[1] << 16); }
  else { if (up < W) out[up] = (int16_t)res[0]; if (up + 1 < W) out[up + 1] = (int16_t)res[1]; }
}

// ------------------------------------------------------------------------------------------------
// Left/right consistency (elas.cpp:909-979), out of place: raw -> D1/D2.
__global__ void __launch_bounds__(256) k_lr(DevParams dp, const FrameInfo* __restrict__ info, const int16_t* __restrict__ raw,
                                            float* __restrict__ D1, float* __restrict__ D2) {
  // One workgroup per image row: both raw rows go to LDS with coalesced loads, the data-dependent look-ups
  // (the partner pixel u -/+ d in the other image's row) then hit LDS instead of scattering over the row in memory.
  // The matcher's output is integer valued, so it travels as int16.
  extern __shared__ int16_t s_raw[];                     // [2][W]
  const int v = blockIdx.x, frame = blockIdx.y;
  if (!info[frame].ok) return;
  const int W = dp.W;
  const size_t plane = (size_t)dp.H * W;
  const int16_t* r1 = raw + (size_t)(frame * 2) * plane + (size_t)v * W;
  const int16_t* r2 = r1 + plane;
  int16_t* s1 = s_raw; int16_t* s2 = s_raw + W;
  for (int u = threadIdx.x; u < W; u += 256) { s1[u] = r1[u]; s2[u] = r2[u]; }
  __syncthreads();
  const float thr = (float)dp.lr_threshold;
  float* o1row = D1 + (size_t)frame * plane + (size_t)v * W;
  float* o2row = D2 + (size_t)frame * plane + (size_t)v * W;
  for (int u = threadIdx.x; u < W; u += 256) {
    const float d1 = (float)s1[u], d2 = (float)s2[u];
    float o1 = d1, o2 = d2;
    const float w1 = (float)u - d1, w2 = (float)u + d2;
    if (d1 >= 0 && w1 >= 0 && w1 < (float)W) { if (fabsf((float)s2[(int)w1] - d1) > thr) o1 = -10.0f; } else o1 = -10.0f;
    if (d2 >= 0 && w2 >= 0 && w2 < (float)W) { if (fabsf((float)s1[(int)w2] - d2) > thr) o2 = -10.0f; } else o2 = -10.0f;
    o1row[u] = o1;
    o2row[u] = o2;
  }
}

DEV int uf_find(int32_t* __restrict__ lab, int x) {
  // path halving: every visited node is re-pointed at its grandparent.  Parents only ever move towards the
  // root (smaller index), so the unsynchronised stores are benign; without them a tall region builds a
  // chain of one hop per image row and a single find walks hundreds of dependent loads.
  int p = lab[x];
  while (p != x) {
    const int g = lab[p];
    if (g != p) lab[x] = g;
    x = p; p = g;
  }
  return x;
}
DEV void uf_union(int32_t* __restrict__ lab, int a, int b) {
  for (;;) {
    a = uf_find(lab, a); b = uf_find(lab, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }        // a = larger root, hang it under b
    const int old = atomicMin(&lab[a], b);
    if (old == a) return;
    a = old;
  }
}
// Row pass: every maximal horizontal run of connected pixels gets the flat index of its first pixel
// as label (no atomics: a max-scan of "run starts here" positions along the image row), and the row's runs are
// listed compactly — run k of the row spans columns [starts[k], ends[k]] — so that the later passes that work per
// run (sizes, removal) touch ~4 % as many items as there are pixels and never read the image.
// The lists of frame f live inside frame f's own image-sized block of the scratch buffer ([H][pitch] starts, [H][pitch]
// ends, [H] counts < H*W*4 bytes), so that the scratch may be a caller's output image that is idle at this point: frames
// that fail (and whose outputs must stay untouched) see no write.
struct RunLists {
  uint8_t* base; size_t frame_bytes; int pitch, H;
  __device__ uint16_t* starts(int frame) const { return reinterpret_cast<uint16_t*>(base + (size_t)frame * frame_bytes); }
  __device__ uint16_t* ends(int frame) const { return starts(frame) + (size_t)H * pitch; }
  __device__ int32_t* count(int frame) const { return reinterpret_cast<int32_t*>(ends(frame) + (size_t)H * pitch); }
};
// One 64-pixel chunk of a row's run labelling, on one wave: d = this lane's pixel (u = u0 + lane), nxt = the pixel 64 further.
// Neighbours come by DPP wave shifts (lane 0 / 63 take the carried `left` / the next chunk's first pixel as the shift's `old` operand),
// the latest run start by the DPP prefix maximum (row_shr 1, 2, 4, 8, then row_bcast 15 / 31), the run's ordinal by v_mbcnt: ~45 vector
// instructions a chunk where the ds_bpermute forms of round 4 took ~90 (the post chain is bound by vector issue like the rest of the path).
struct RunCarry { int carry = -1, cnt = 0; float left = -10.0f; };          // latest run start / runs so far / pixel just before the chunk
#define JN_DPP(old, v, ctrl, rmask) __builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, rmask, 0xf, false)
DEV void ccl_chunk(float d, float nxt, int u, int W, int v, float sim, RunCarry& c, int32_t* __restrict__ lab_row, int32_t* __restrict__ sz_row,
                   uint16_t* __restrict__ rs, uint16_t* __restrict__ re) {
  const float nxt0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nxt)));
  const float prev = __int_as_float(JN_DPP(__float_as_int(c.left), __float_as_int(d), 0x138, 0xf));      // wave_shr:1
  const float foll = __int_as_float(JN_DPP(__float_as_int(nxt0), __float_as_int(d), 0x130, 0xf));        // wave_shl:1
  const bool valid = d >= 0;
  const bool conn = valid && prev >= 0 && fabsf(d - prev) <= sim;
  const bool conn_next = valid && foll >= 0 && fabsf(foll - d) <= sim;
  const bool start = valid && !conn, last = valid && !conn_next;
  // (run starts biased by one, unsigned: 0 = none is then the maximum's identity AND what a DPP read beyond the row returns, so each step
  // is ONE v_max_u32 with a DPP operand)
  uint32_t sb = start ? (uint32_t)u + 1u : 0u;
#define JN_DPP0(v, ctrl, rmask) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xf, true)
  sb = max(sb, JN_DPP0(sb, 0x111, 0xf)); sb = max(sb, JN_DPP0(sb, 0x112, 0xf));                         // row_shr:1, :2
  sb = max(sb, JN_DPP0(sb, 0x114, 0xf)); sb = max(sb, JN_DPP0(sb, 0x118, 0xf));                         // row_shr:4, :8
  sb = max(sb, JN_DPP0(sb, 0x142, 0xa));                                                                // row_bcast:15 into rows 1 and 3
  sb = max(sb, JN_DPP0(sb, 0x143, 0xc));                                                                // row_bcast:31 into rows 2 and 3
#undef JN_DPP0
  const int s = max((int)sb - 1, c.carry);
  const unsigned long long starts = __ballot(start);
  const int k = c.cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0u)) + (start ? 0 : -1);   // ordinal of the run this pixel belongs to
  if (u < W) lab_row[u] = valid ? v * W + s : -1;
  if (start) { sz_row[u] = 0; rs[k] = (uint16_t)u; }         // sizes live at roots, and roots are run starts
  if (last) re[k] = (uint16_t)u;
  c.carry = __builtin_amdgcn_readlane(s, 63);
  c.cnt += __popcll(starts);
  c.left = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));
}
// One wave per image row, walking it in 64-pixel chunks: the carries (latest run start, runs so far) are wave-uniform,
// so there is no LDS and no barrier, and the next chunk's pixels are loaded before the current one is processed.
__global__ void __launch_bounds__(256) k_ccl_rows(DevParams dp, const FrameInfo* __restrict__ info, const float* __restrict__ D,
                                                  int32_t* __restrict__ lab, int32_t* __restrict__ sz, RunLists runs) {
  const int v = blockIdx.x * 4 + (threadIdx.x >> 6), frame = blockIdx.y;
  if (v >= dp.H || !info[frame].ok) return;
  const int W = dp.W, lane = threadIdx.x & 63;
  const size_t base = (size_t)frame * dp.H * W + (size_t)v * W;
  const size_t rbase = (size_t)v * runs.pitch;
  uint16_t* r_starts = runs.starts(frame); uint16_t* r_ends = runs.ends(frame);
  const float* row = D + base;
  const float sim = dp.speckle_sim;
  RunCarry c;
  float d = lane < W ? row[lane] : -10.0f;
  for (int u0 = 0; u0 < W; u0 += 64) {
    const int u = u0 + lane;
    const float nxt = u + 64 < W ? row[u + 64] : -10.0f;    // next chunk, in flight while this one is processed
    ccl_chunk(d, nxt, u, W, v, sim, c, lab + base, sz + base, r_starts + rbase, r_ends + rbase);
    d = nxt;
  }
  if (lane == 0) runs.count(frame)[v] = c.cnt;
}
// k_lr and k_ccl_rows in one pass over the row (the route where the left map travels raw -> tmp -> D1): the workgroup that has just formed the
// row's L/R-checked values keeps the left ones in LDS, and its first wave labels their runs from there — the row is not read back from
// memory (4 of the two kernels' 20 bytes per pixel) and one launch is gone.  Same outputs as k_lr (D1 = the checked left row, D2) followed
// by k_ccl_rows on D1.
__global__ void __launch_bounds__(256) k_lr_ccl_rows(DevParams dp, const FrameInfo* __restrict__ info, const int16_t* __restrict__ raw,
                                                     float* __restrict__ D1, float* __restrict__ D2, int32_t* __restrict__ lab, int32_t* __restrict__ sz, RunLists runs) {
  extern __shared__ int16_t s_raw[];                     // [2][W] raw rows, then [W] float: the checked left row
  const int v = blockIdx.x, frame = blockIdx.y;
  if (!info[frame].ok) return;
  const int W = dp.W;
  const size_t plane = (size_t)dp.H * W;
  const int16_t* r1 = raw + (size_t)(frame * 2) * plane + (size_t)v * W;
  const int16_t* r2 = r1 + plane;
  int16_t* s1 = s_raw; int16_t* s2 = s_raw + W;
  float* s_o1 = reinterpret_cast<float*>(s_raw + 2 * ((W + 1) & ~1));
  for (int u = threadIdx.x; u < W; u += 256) { s1[u] = r1[u]; s2[u] = r2[u]; }
  __syncthreads();
  const float thr = (float)dp.lr_threshold;
  float* o1row = D1 + (size_t)frame * plane + (size_t)v * W;
  float* o2row = D2 + (size_t)frame * plane + (size_t)v * W;
  for (int u = threadIdx.x; u < W; u += 256) {           // elas.cpp:909-979, as k_lr
    const float d1 = (float)s1[u], d2 = (float)s2[u];
    float o1 = d1, o2 = d2;
    const float w1 = (float)u - d1, w2 = (float)u + d2;
    if (d1 >= 0 && w1 >= 0 && w1 < (float)W) { if (fabsf((float)s2[(int)w1] - d1) > thr) o1 = -10.0f; } else o1 = -10.0f;
    if (d2 >= 0 && w2 >= 0 && w2 < (float)W) { if (fabsf((float)s1[(int)w2] - d2) > thr) o2 = -10.0f; } else o2 = -10.0f;
    o1row[u] = o1; s_o1[u] = o1;
    o2row[u] = o2;
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  // the row's runs, as k_ccl_rows (one wave, 64 pixels at a time, wave-uniform carries), reading the row from LDS
  const int lane = threadIdx.x;
  const size_t base = (size_t)frame * dp.H * W + (size_t)v * W;
  const size_t rbase = (size_t)v * runs.pitch;
  uint16_t* r_starts = runs.starts(frame); uint16_t* r_ends = runs.ends(frame);
  const float sim = dp.speckle_sim;
  RunCarry c;
  float d = lane < W ? s_o1[lane] : -10.0f;
  for (int u0 = 0; u0 < W; u0 += 64) {
    const int u = u0 + lane;
    const float nxt = u + 64 < W ? s_o1[u + 64] : -10.0f;
    ccl_chunk(d, nxt, u, W, v, sim, c, lab + base, sz + base, r_starts + rbase, r_ends + rbase);
    d = nxt;
  }
  if (lane == 0) runs.count(frame)[v] = c.cnt;
}
// Column pass: unite vertically adjacent runs.  A pixel issues the union only if it is the first
// column of the contact between its run and the run below (the pixel to its left belongs to the
// same two runs otherwise), which removes almost all redundant atomics.
__global__ void __launch_bounds__(256) k_ccl_merge(DevParams dp, const FrameInfo* __restrict__ info, const float* __restrict__ D,
                                                   int32_t* __restrict__ lab, int segs) {
  // one wave per pair of rows (v, v+1) — per column segment of it when a small batch has few rows to offer —
  // 64 columns at a time; left neighbours come from the lane below
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), v = wave / segs, seg = wave - v * segs, frame = blockIdx.y;
  if (v + 1 >= dp.H || !info[frame].ok) return;
  const int W = dp.W, lane = threadIdx.x & 63;
  const int seg_len = ((W + segs - 1) / segs + 63) & ~63, c0 = seg * seg_len, c1 = min(c0 + seg_len, W);
  if (c0 >= W) return;
  const size_t plane = (size_t)dp.H * W;
  const float* r0 = D + frame * plane + (size_t)v * W;
  const float* r1 = r0 + W;
  int32_t* L = lab + frame * plane;
  const float sim = dp.speckle_sim;
  float a_left = c0 > 0 ? r0[c0 - 1] : -10.0f, b_left = c0 > 0 ? r1[c0 - 1] : -10.0f;      // column just before the chunk
  float a = c0 + lane < c1 ? r0[c0 + lane] : -10.0f, b = c0 + lane < c1 ? r1[c0 + lane] : -10.0f;
  for (int u0 = c0; u0 < c1; u0 += 64) {
    const int u = u0 + lane;
    const float an = u + 64 < c1 ? r0[u + 64] : -10.0f, bn = u + 64 < c1 ? r1[u + 64] : -10.0f;   // next chunk in flight
    float a0 = __shfl_up(a, 1), b0 = __shfl_up(b, 1);
    if (lane == 0) { a0 = a_left; b0 = b_left; }
    const bool contact = a >= 0 && b >= 0 && fabsf(a - b) <= sim;
    const bool same_pair = a0 >= 0 && b0 >= 0 && fabsf(a0 - b0) <= sim && fabsf(a - a0) <= sim && fabsf(b - b0) <= sim;
    if (contact && !same_pair) uf_union(L, L[v * W + u], L[(v + 1) * W + u]);
    a_left = __shfl(a, 63); b_left = __shfl(b, 63);
    a = an; b = bn;
  }
}
// One atomic per run: the run length goes to the component root.  Four rows per workgroup, one wave per row
// striding over that row's run list.
__global__ void __launch_bounds__(256) k_ccl_count(DevParams dp, const FrameInfo* __restrict__ info, int32_t* __restrict__ lab,
                                                   int32_t* __restrict__ sz, RunLists runs) {
  const int v = blockIdx.x * 4 + (threadIdx.x >> 6), frame = blockIdx.y;
  if (v >= dp.H || !info[frame].ok) return;
  const int W = dp.W;
  const size_t plane = (size_t)dp.H * W;
  int32_t* L = lab + frame * plane;
  const int cnt = runs.count(frame)[v];
  const uint16_t* r_starts = runs.starts(frame) + (size_t)v * runs.pitch; const uint16_t* r_ends = runs.ends(frame) + (size_t)v * runs.pitch;
  for (int k = threadIdx.x & 63; k < cnt; k += 64) {
    const int start = v * W + r_starts[k], len = r_ends[k] - r_starts[k] + 1;
    const int root = uf_find(L, start);
    if (root != start) L[start] = root;                     // compress: k_ccl_apply then needs at most two hops
    // Only "at least speckle_size or not" matters (elas.cpp:1083): once a component is seen to have reached the
    // threshold, further runs skip the add.  The scene's few huge components would otherwise take ~10^5 atomic adds
    // each on one address.  A stale read only means one add too many.
    int32_t* total = &sz[frame * plane + root];
    if (*total < dp.speckle_size) atomicAdd(total, len);
  }
}
// Runs of components smaller than speckle_size are set to -10 (invalid pixels already are: every invalid value the
// L/R check leaves is -10, the "segment of one" of elas.cpp:1075-1090 changes nothing for them).
__global__ void __launch_bounds__(256) k_ccl_apply(DevParams dp, const FrameInfo* __restrict__ info, float* __restrict__ D,
                                                   const int32_t* __restrict__ lab, const int32_t* __restrict__ sz, RunLists runs) {
  const int v = blockIdx.x * 4 + (threadIdx.x >> 6), frame = blockIdx.y;
  if (v >= dp.H || !info[frame].ok) return;
  const int W = dp.W;
  const size_t plane = (size_t)dp.H * W;
  const int32_t* L = lab + frame * plane;
  const int cnt = runs.count(frame)[v];
  const uint16_t* r_starts = runs.starts(frame) + (size_t)v * runs.pitch; const uint16_t* r_ends = runs.ends(frame) + (size_t)v * runs.pitch;
  for (int k = threadIdx.x & 63; k < cnt; k += 64) {
    const int us = r_starts[k], ue = r_ends[k];
    int x = v * W + us, q = L[x];
    while (q != x) { x = q; q = L[x]; }
    if (sz[frame * plane + x] < dp.speckle_size) {
      float* row = D + frame * plane + (size_t)v * W;
      for (int u = us; u <= ue; u++) row[u] = -10.0f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Gap interpolation (elas.cpp:1101-1284) as a gather: an invalid pixel whose nearest valid
// neighbours along the line are `a` steps before and `b` steps after, with run length a+b-1 <=
// ipol_gap_width, takes (d1+d2)/2 if |d1-d2| < 3 else min(d1,d2).  Equivalent to the sequential
// run scan because fills never become bounds of other runs.
template <bool kRows>
__global__ void __launch_bounds__(256) k_gap(DevParams dp, const FrameInfo* __restrict__ info, const float* __restrict__ in,
                                             float* __restrict__ out) {
  const int u = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y, frame = blockIdx.z;
  if (u >= dp.W || !info[frame].ok) return;
  const size_t plane = (size_t)dp.H * dp.W;
  const float* I = in + frame * plane;
  const int W = dp.W, stride = kRows ? 1 : W, pos = kRows ? u : v, len = kRows ? W : dp.H;
  const size_t p = (size_t)v * W + u;
  float val = I[p];
  if (!(val >= 0)) {
    int a = 0, b = 0;
    for (int k = 1; k <= dp.gap_width && pos - k >= 0; k++) if (I[p - (size_t)k * stride] >= 0) { a = k; break; }
    if (a) {
      for (int k = 1; k <= dp.gap_width - a + 1 && pos + k < len; k++) if (I[p + (size_t)k * stride] >= 0) { b = k; break; }
      if (b) {
        const float d1 = I[p - (size_t)a * stride], d2 = I[p + (size_t)b * stride];
        val = fabsf(d1 - d2) < 3.0f ? __fadd_rn(d1, d2) / 2 : fminf(d1, d2);   // :1149-1150
      }
    }
  }
  out[frame * plane + p] = val;
}

// Four pixels per thread (widths that are multiples of 4): most pixels are valid and just pass through, so the common
// case is one 16-byte load and one 16-byte store; only invalid pixels look along their line.
DEV float gap_fill(const DevParams& dp, const float* I, size_t p, int stride, int pos, int len) {
  int a = 0, b = 0;
  for (int k = 1; k <= dp.gap_width && pos - k >= 0; k++) if (I[p - (size_t)k * stride] >= 0) { a = k; break; }
  if (!a) return I[p];
  for (int k = 1; k <= dp.gap_width - a + 1 && pos + k < len; k++) if (I[p + (size_t)k * stride] >= 0) { b = k; break; }
  if (!b) return I[p];
  const float d1 = I[p - (size_t)a * stride], d2 = I[p + (size_t)b * stride];
  return fabsf(d1 - d2) < 3.0f ? __fadd_rn(d1, d2) / 2 : fminf(d1, d2);     // :1149-1150
}
template <bool kRows>
__global__ void __launch_bounds__(64) k_gap4(DevParams dp, const FrameInfo* __restrict__ info, const float* __restrict__ in,
                                             float* __restrict__ out) {
  const int u0 = (blockIdx.x * 64 + threadIdx.x) * 4, v = blockIdx.y, frame = blockIdx.z;
  if (u0 >= dp.W || !info[frame].ok) return;
  const int W = dp.W;
  const size_t plane = (size_t)dp.H * W;
  const float* I = in + frame * plane;
  const size_t p0 = (size_t)v * W + u0;
  const float4 x = *reinterpret_cast<const float4*>(I + p0);
  float r[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (!(r[i] >= 0)) r[i] = gap_fill(dp, I, p0 + i, kRows ? 1 : W, kRows ? u0 + i : v, kRows ? W : dp.H);
  *reinterpret_cast<float4*>(out + frame * plane + p0) = make_float4(r[0], r[1], r[2], r[3]);
}

// Gap interpolation for any gap width and with the border extrapolation of the add_corners presets
// (elas.cpp:1137-1198, :1218-1282).  Per line: an invalid pixel with valid neighbours at L (before) and R (after) and a run
// R - L - 1 <= gap_width takes (d1+d2)/2 if |d1-d2| < 3 else min(d1,d2); with add_corners the pixels before the line's first
// valid pixel (at most gap_width of them) take its value and the pixels after the last valid one likewise.  Fills never
// become bounds of other runs, so this gather form equals the reference's sequential scan.
// Rows: one wave per image row; the row and, per pixel, the index of the nearest valid pixel at or before / at or after
// it live in LDS (forward max-scan and backward min-scan in 64-pixel chunks with wave-uniform carries).
__global__ void __launch_bounds__(256) k_gap_rows_any(DevParams dp, const FrameInfo* __restrict__ info, const float* __restrict__ in,
                                                      float* __restrict__ out) {
  extern __shared__ float s_gap[];                           // per wave: [W] values | [W] prev index | [W] next index (as int)
  const int W = dp.W, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int v = blockIdx.x * (blockDim.x >> 6) + wave, frame = blockIdx.y;
  if (v >= dp.H || !info[frame].ok) return;
  float* val = s_gap + (size_t)wave * 3 * W;
  int* prev = reinterpret_cast<int*>(val + W);
  int* next = prev + W;
  const size_t row = ((size_t)frame * dp.H + v) * W;
  int carry = -1;
  for (int u0 = 0; u0 < W; u0 += 64) {                       // forward: index of the nearest valid pixel at or before u
    const int u = u0 + lane;
    const float x = u < W ? in[row + u] : -10.0f;
    int p = (u < W && x >= 0) ? u : -1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(p, off); if (lane >= off) p = max(p, o); }
    p = max(p, carry);
    if (u < W) { val[u] = x; prev[u] = p; }
    carry = __shfl(p, 63);
  }
  carry = 1 << 30;
  for (int u0 = ((W - 1) / 64) * 64; u0 >= 0; u0 -= 64) {    // backward: nearest valid pixel at or after u
    const int u = u0 + lane;
    int q = (u < W && val[u] >= 0) ? u : (1 << 30);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_down(q, off); if (lane + off < 64) q = min(q, o); }
    q = min(q, carry);
    if (u < W) next[u] = q;
    carry = __shfl(q, 0);
  }
  const int gw = dp.gap_width;
  for (int u = lane; u < W; u += 64) {
    float x = val[u];
    if (!(x >= 0)) {
      const int L = prev[u], R = next[u];
      if (L >= 0 && R < W) {
        if (R - L - 1 <= gw) { const float d1 = val[L], d2 = val[R]; x = fabsf(d1 - d2) < 3.0f ? __fadd_rn(d1, d2) / 2 : fminf(d1, d2); }   // :1149-1150
      } else if (dp.add_corners) {
        if (L < 0 && R < W) { if (u >= R - gw) x = val[R]; }             // :1172-1183
        else if (L >= 0 && R >= W) { if (u <= L + gw) x = val[L]; }      // :1186-1197
      }
    }
    out[row + u] = x;
  }
}
// Columns: one thread per column walking down; every pixel is passed through as it is read, and when a valid pixel closes a
// run of invalid ones (or the column ends) the run is filled behind it.  Neighbouring threads touch neighbouring addresses.
__global__ void __launch_bounds__(256) k_gap_cols_any(DevParams dp, const FrameInfo* __restrict__ info, const float* __restrict__ in,
                                                      float* __restrict__ out) {
  const int u = blockIdx.x * 256 + threadIdx.x, frame = blockIdx.y;
  if (u >= dp.W || !info[frame].ok) return;
  const int W = dp.W, H = dp.H, gw = dp.gap_width;
  const float* I = in + (size_t)frame * H * W + u;
  float* O = out + (size_t)frame * H * W + u;
  int last = -1;                                             // row of the last valid pixel seen
  float dlast = 0;
  for (int v = 0; v < H; v++) {
    const float x = I[(size_t)v * W];
    O[(size_t)v * W] = x;
    if (x >= 0) {
      const int count = v - last - 1;
      if (count >= 1) {
        if (last >= 0) {
          if (count <= gw) {
            const float d = fabsf(dlast - x) < 3.0f ? __fadd_rn(dlast, x) / 2 : fminf(dlast, x);       // :1230-1231
            for (int w = last + 1; w < v; w++) O[(size_t)w * W] = d;
          }
        } else if (dp.add_corners) {
          for (int w = max(v - gw, 0); w < v; w++) O[(size_t)w * W] = x;                                // :1256-1267
        }
      }
      last = v; dlast = x;
    }
  }
  if (dp.add_corners && last >= 0)
    for (int w = last + 1; w <= min(last + gw, H - 1); w++) O[(size_t)w * W] = dlast;                  // :1270-1281
}

// ------------------------------------------------------------------------------------------------
// Adaptive mean (elas.cpp:1287-1492, full-resolution branch).  The reference's "abs mask" is
// _mm_set1_ps(0x7FFFFFFF) = 2^31 as a float (0x4F000000), so the weight keeps a few exponent
// bits of the difference; reproduced bit for bit, as is the summation order of its 8-slot ring
// (slot = position mod 8; lane l = slot l + slot l+4; total = ((l0+l1)+l2)+l3).
DEV float am_weight(float x, float c) {
  const float m = __uint_as_float(__float_as_uint(__fsub_rn(x, c)) & 0x4F000000u);
  return fmaxf(0.0f, __fsub_rn(4.0f, m));
}
// Two taps at once with packed f32 (v_pk_add_f32 / v_pk_mul_f32: the same IEEE operations, two per instruction; the
// means are bound by vector issue): pw = w(x0) + w(x1), pf = x0 w(x0) + x1 w(x1).
typedef float jn_f2 __attribute__((ext_vector_type(2)));
DEV void am_pair(float x0, float x1, float c, float& pw, float& pf) {
  const jn_f2 X = {x0, x1}, C = {c, c}, four = {4.0f, 4.0f};
  const jn_f2 d = X - C;
  jn_f2 m;
  m.x = __uint_as_float(__float_as_uint(d.x) & 0x4F000000u); m.y = __uint_as_float(__float_as_uint(d.y) & 0x4F000000u);
  jn_f2 w = four - m;
  w.x = fmaxf(0.0f, w.x); w.y = fmaxf(0.0f, w.y);
  const jn_f2 p = X * w;
  pw = w.x + w.y; pf = p.x + p.y;
}
// ---- subsampling = 1 (elas.h:82): half-size maps ------------------------------------------------------------------------
// The dense matching is per pixel (elas.cpp:683-780), so the reference's half-size map is the full one at even (u, v)
// (:693, :877-896): the matcher runs unchanged and this pass picks every second pixel of every second row while it applies
// the left/right check of the half-size maps, whose warps are by d / 2 (:914-941).  One workgroup per half-size row.
__global__ void __launch_bounds__(256) k_lr_sub(DevParams dp, const FrameInfo* __restrict__ info, const int16_t* __restrict__ raw,
                                                float* __restrict__ D1, float* __restrict__ D2) {
  extern __shared__ int16_t s_raw[];                     // [2][W / 2]
  const int v = blockIdx.x, frame = blockIdx.y;
  if (!info[frame].ok) return;
  const int W = dp.W, Wh = dp.W / 2, Hh = dp.H / 2;
  const size_t plane = (size_t)dp.H * W, hplane = (size_t)Hh * Wh;
  const int16_t* r1 = raw + (size_t)(frame * 2) * plane + (size_t)(2 * v) * W;
  const int16_t* r2 = r1 + plane;
  int16_t* s1 = s_raw; int16_t* s2 = s_raw + Wh;
  for (int u = threadIdx.x; u < Wh; u += 256) { s1[u] = r1[2 * u]; s2[u] = r2[2 * u]; }
  __syncthreads();
  const float thr = (float)dp.lr_threshold;
  float* o1row = D1 + (size_t)frame * hplane + (size_t)v * Wh;
  float* o2row = D2 + (size_t)frame * hplane + (size_t)v * Wh;
  for (int u = threadIdx.x; u < Wh; u += 256) {
    const float d1 = (float)s1[u], d2 = (float)s2[u];
    float o1 = d1, o2 = d2;
    const float w1 = __fsub_rn((float)u, d1 / 2), w2 = __fadd_rn((float)u, d2 / 2);
    if (d1 >= 0 && w1 >= 0 && w1 < (float)Wh) { if (fabsf((float)s2[(int)w1] - d1) > thr) o1 = -10.0f; } else o1 = -10.0f;
    if (d2 >= 0 && w2 >= 0 && w2 < (float)Wh) { if (fabsf((float)s1[(int)w2] - d2) > thr) o2 = -10.0f; } else o2 = -10.0f;
    o1row[u] = o1;
    o2row[u] = o2;
  }
}
// The 4-pixel adaptive mean of the half-size map (elas.cpp:1323-1391): the window of output pixel c is {c-2, c-1, c, c+1}, kept by the
// reference in a ring of four (slot = position mod 4) and summed in SLOT order ((s0 + s1) + s2) + s3.  kRows = false: along the row,
// in -> out for every pixel (pixels the pass leaves alone are copied: the reference's D_tmp is -10 where D is invalid, :1304-1309, and
// heap memory elsewhere — a copy of D here, as in the full-size form).  kRows = true: down the column, in = the first pass's output,
// D is written only where a mean forms (:1383-1386).
template <bool kCols>
__global__ void __launch_bounds__(256) k_adaptive_mean_sub(DevParams dp, const FrameInfo* __restrict__ info,
                                                           const float* __restrict__ in, float* __restrict__ out) {
  const int u = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y, frame = blockIdx.z;
  if (u >= dp.W || !info[frame].ok) return;
  const int W = dp.W, H = dp.H;
  const size_t plane = (size_t)H * W;
  const float* I = in + frame * plane;
  const size_t p = (size_t)v * W + u;
  const float c = I[p];
  const bool inside = kCols ? (u >= 3 && u < W - 3 && v >= 2 && v <= H - 2) : (v >= 3 && v < H - 3 && u >= 2 && u <= W - 2);
  float res = c; bool formed = false;
  if (inside) {
    const int stride = kCols ? W : 1, pos = kCols ? v : u;
    float w[4], f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {                          // tap k = pixel pos - 2 + k, ring slot (pos - 2 + k) & 3
      const float x = I[p + (long long)(k - 2) * stride];
      w[k] = am_weight(x, c); f[k] = __fmul_rn(x, w[k]);
    }
    const int r = (pos - 2) & 3;                            // slot j holds tap (j - r) & 3: rotate the taps right by r
    if (r & 1) {
      const float tw = w[3], tf = f[3];
      w[3] = w[2]; w[2] = w[1]; w[1] = w[0]; w[0] = tw;
      f[3] = f[2]; f[2] = f[1]; f[1] = f[0]; f[0] = tf;
    }
    if (r & 2) {
      float t = w[0]; w[0] = w[2]; w[2] = t; t = w[1]; w[1] = w[3]; w[3] = t;
      t = f[0]; f[0] = f[2]; f[2] = t; t = f[1]; f[1] = f[3]; f[3] = t;
    }
    const float ws = __fadd_rn(__fadd_rn(__fadd_rn(w[0], w[1]), w[2]), w[3]);
    const float fs = __fadd_rn(__fadd_rn(__fadd_rn(f[0], f[1]), f[2]), f[3]);
    if (ws > 0) { const float d = fs / ws; if (d >= 0) { res = d; formed = true; } }
  }
  if (!kCols) out[frame * plane + p] = res;
  else if (formed) out[frame * plane + p] = res;
}

// ------------------------------------------------------------------------------------------------
// Speckle removal (elas.cpp:981-1099) as union-find connected components: 4-connected valid
// pixels whose disparities differ by <= speckle_sim_threshold; components smaller than
// speckle_size (and every invalid pixel, a "segment" of one) are set to -10.  The reference's
// flood fill visits the same components, so the result is order-free.
// Horizontal pass: in = D, out = tmp; rows 3..H-4, centres 4..W-4.
__global__ void __launch_bounds__(256) k_adaptive_mean_h(DevParams dp, const FrameInfo* __restrict__ info,
                                                         const float* __restrict__ in, float* __restrict__ out) {
  const int u = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y, frame = blockIdx.z;
  if (u >= dp.W || !info[frame].ok) return;
  const int W = dp.W, H = dp.H;
  const size_t plane = (size_t)H * W;
  const float* I = in + frame * plane;
  const size_t p = (size_t)v * W + u;
  // The reference first copies D with negatives replaced by -10 (elas.cpp:1304-1309).  Here every invalid pixel
  // already holds exactly -10 when this pass runs: k_lr writes -10 for everything it rejects, speckle removal writes
  // -10, gap interpolation only ever turns invalid pixels into valid ones.  So the copy is the identity.
  const float c = I[p];
  float res = c;
  if (v >= 3 && v < H - 3 && u >= 4 && u <= W - 4) {
    // Window [c-4, c+3]; the reference keeps it in an 8-slot ring (slot = position mod 8) and sums lane l =
    // slot l + slot l+4, then ((l0+l1)+l2)+l3.  Window taps k and k+4 always share a lane, so with pair sums
    // P_j = tap j + tap j+4 (addition commutes) lane l holds P[(l - pos) & 3]: a rotation by pos & 3.
    float pw[4], pf[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float x0 = I[p + k - 4], x1 = I[p + k];
      const float w0 = am_weight(x0, c), w1 = am_weight(x1, c);
      pw[k] = __fadd_rn(w0, w1);
      pf[k] = __fadd_rn(__fmul_rn(x0, w0), __fmul_rn(x1, w1));
    }
    const int r = u & 3;
    if (r & 1) {
      const float tw = pw[3], tf = pf[3];
      pw[3] = pw[2]; pw[2] = pw[1]; pw[1] = pw[0]; pw[0] = tw;
      pf[3] = pf[2]; pf[2] = pf[1]; pf[1] = pf[0]; pf[0] = tf;
    }
    if (r & 2) {
      float t = pw[0]; pw[0] = pw[2]; pw[2] = t; t = pw[1]; pw[1] = pw[3]; pw[3] = t;
      t = pf[0]; pf[0] = pf[2]; pf[2] = t; t = pf[1]; pf[1] = pf[3]; pf[3] = t;
    }
    const float ws = __fadd_rn(__fadd_rn(__fadd_rn(pw[0], pw[1]), pw[2]), pw[3]);
    const float fs = __fadd_rn(__fadd_rn(__fadd_rn(pf[0], pf[1]), pf[2]), pf[3]);
    if (ws > 0) { const float d = fs / ws; if (d >= 0) res = d; }
  }
  out[frame * plane + p] = res;
}

// Horizontal pass, four consecutive pixels per thread (widths that are multiples of 4): the 11 values their windows
// span arrive as three 16-byte loads, and because u & 3 is then known at compile time the ring order is a
// straight-line case per pixel instead of a lane-wise rotation.  Same arithmetic as k_adaptive_mean_h.
__global__ void __launch_bounds__(64) k_adaptive_mean_h4(DevParams dp, const FrameInfo* __restrict__ info,
                                                          const float* __restrict__ in, float* __restrict__ out) {
  const int u0 = (blockIdx.x * 64 + threadIdx.x) * 4, v = blockIdx.y, frame = blockIdx.z;
  if (u0 >= dp.W || !info[frame].ok) return;
  const int W = dp.W, H = dp.H;
  const size_t row = ((size_t)frame * H + v) * W;
  const float4* I4 = reinterpret_cast<const float4*>(in + row);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 a = u0 >= 4 ? I4[u0 / 4 - 1] : zero, b = I4[u0 / 4], c4 = u0 + 4 < W ? I4[u0 / 4 + 1] : zero;
  const float x[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c4.x, c4.y, c4.z, c4.w};   // x[k] = pixel u0 - 4 + k
  float res[4] = {b.x, b.y, b.z, b.w};
  if (v >= 3 && v < H - 3) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int u = u0 + i;
      if (u < 4 || u > W - 4) continue;
      const float c = x[4 + i];
      float pw[4], pf[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        am_pair(x[i + k], x[i + k + 4], c, pw[k], pf[k]);
      }
      // ring lane l = P[(l - u) & 3] with u & 3 == i
      const float ws = __fadd_rn(__fadd_rn(__fadd_rn(pw[(0 - i) & 3], pw[(1 - i) & 3]), pw[(2 - i) & 3]), pw[(3 - i) & 3]);
      const float fs = __fadd_rn(__fadd_rn(__fadd_rn(pf[(0 - i) & 3], pf[(1 - i) & 3]), pf[(2 - i) & 3]), pf[(3 - i) & 3]);
      if (ws > 0) { const float d = fs / ws; if (d >= 0) res[i] = d; }
    }
  }
  reinterpret_cast<float4*>(out + row)[u0 / 4] = make_float4(res[0], res[1], res[2], res[3]);
}

// Vertical pass, one thread per column walking kAmRows rows with the 8-tap window in registers: 1.4 row reads per
// output instead of 8 (a thread-per-pixel variant is bound by L2 requests, not by arithmetic).  Same arithmetic as the
// horizontal pass on columns 3..W-4, centres 4..H-4: in = tmp (horizontal result); D keeps its value where no mean forms.
constexpr int kAmRows = 16;
__global__ void __launch_bounds__(256) k_adaptive_mean_v(DevParams dp, const FrameInfo* __restrict__ info,
                                                         const float* __restrict__ in, float* __restrict__ D) {
  const int u = blockIdx.x * 256 + threadIdx.x, v0 = blockIdx.y * kAmRows, frame = blockIdx.z;
  if (u >= dp.W || !info[frame].ok) return;
  const int W = dp.W, H = dp.H;
  const size_t plane = (size_t)H * W;
  const float* I = in + frame * plane + u;
  float* O = D + frame * plane + u;
  if (u < 3 || u >= W - 3) return;                       // columns the pass leaves as they are (out == keep)
  float x[8];
#pragma unroll
  for (int k = 0; k < 7; k++) { const int row = v0 - 4 + k; x[k] = (row >= 0 && row < H) ? I[(size_t)row * W] : 0.0f; }
  const int v1 = min(v0 + kAmRows, H);
  for (int v = v0; v < v1; v++) {
    x[7] = (v + 3 < H) ? I[(size_t)(v + 3) * W] : 0.0f;
    if (v >= 4 && v <= H - 4) {
      const float c = x[4];
      float pw[4], pf[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        am_pair(x[k], x[k + 4], c, pw[k], pf[k]);
      }
      // lane l of the reference's ring holds P[(l - v) & 3]; v is the same for the whole wave, so the four possible
      // orders are four straight-line cases instead of a lane-wise rotation
      float ws, fs;
      switch (v & 3) {
        case 0:  ws = __fadd_rn(__fadd_rn(__fadd_rn(pw[0], pw[1]), pw[2]), pw[3]); fs = __fadd_rn(__fadd_rn(__fadd_rn(pf[0], pf[1]), pf[2]), pf[3]); break;
        case 1:  ws = __fadd_rn(__fadd_rn(__fadd_rn(pw[3], pw[0]), pw[1]), pw[2]); fs = __fadd_rn(__fadd_rn(__fadd_rn(pf[3], pf[0]), pf[1]), pf[2]); break;
        case 2:  ws = __fadd_rn(__fadd_rn(__fadd_rn(pw[2], pw[3]), pw[0]), pw[1]); fs = __fadd_rn(__fadd_rn(__fadd_rn(pf[2], pf[3]), pf[0]), pf[1]); break;
        default: ws = __fadd_rn(__fadd_rn(__fadd_rn(pw[1], pw[2]), pw[3]), pw[0]); fs = __fadd_rn(__fadd_rn(__fadd_rn(pf[1], pf[2]), pf[3]), pf[0]); break;
      }
      if (ws > 0) { const float d = fs / ws; if (d >= 0) O[(size_t)v * W] = d; }
    }
#pragma unroll
    for (int k = 0; k < 7; k++) x[k] = x[k + 1];
  }
}

// ------------------------------------------------------------------------------------------------
// Gap interpolation (rows, then columns) and adaptive mean (horizontal, then vertical) in ONE pass over the image
// (elas.cpp:1101-1284 without the add_corners branches, :1287-1492): the four separate kernels above read and write the
// whole float image four times (0.94 GB per 32-pair batch at ~4 TB/s = 0.24 ms); here a workgroup of 256 threads owns
// 256 columns and walks down the rows, so every pixel is read once and written once.
//   row y arrives -> G1(y)   = row gap fill: neighbours of the same row through an LDS copy of the row
//                 -> G2(y-3) = column gap fill: the thread's own ring of the last 8 G1 values
//                 -> T(y-4)  = horizontal mean of G2(y-4): neighbours through a second LDS row
//                 -> out(y-7) = vertical mean over the thread's ring of the last 8 T values (or G2 where no mean forms)
// One barrier per row (the two LDS rows are double-buffered by row parity).  The rings are indexed by row mod 8 with the
// row loop unrolled by 8, so every ring access is a fixed register — and the reference's summation order, "lane l = ring
// slot l + slot l+4", is simply "the two window rows (columns) congruent to l mod 4": no rotation is needed at all; in
// the horizontal pass the thread reads its taps from LDS in that order (per-lane offsets fixed by u mod 4).
// Halo: 8 columns left / 7 right of the 240 output columns are recomputed by the neighbouring workgroup; 14 rows above /
// below a row band likewise.  Applies for gap widths <= 3 without add_corners; other settings keep the separate kernels.
enum { kPostCols = 240, kPostHaloL = 8 };
DEV float gap_value(float d1, float d2) { return fabsf(d1 - d2) < 3.0f ? __fadd_rn(d1, d2) / 2 : fminf(d1, d2); }   // :1149-1150
// One pixel of a gap pass for gap widths <= 3, without branches: c is kept when valid; otherwise the first valid neighbour
// among b1, b2, b3 (before: left / up, at most gw away) and the first valid one among a1, a2, a3 (after) within what the gap
// width leaves (run length = steps before + steps after - 1 <= gw, :1137-1139) interpolate it.
DEV float gap_fill3(int gw, float c, float b1, float b2, float b3, float a1, float a2, float a3) {
  const bool p1 = gw >= 1 && b1 >= 0, p2 = gw >= 2 && b2 >= 0, p3 = gw >= 3 && b3 >= 0;
  const int steps = p1 ? 1 : (p2 ? 2 : 3);
  const float d1 = p1 ? b1 : (p2 ? b2 : b3);
  const int reach = gw - steps + 1;
  const bool q1 = a1 >= 0, q2 = reach >= 2 && a2 >= 0, q3 = reach >= 3 && a3 >= 0;
  const float d2 = q1 ? a1 : (q2 ? a2 : a3);
  const bool fill = !(c >= 0) && (p1 || p2 || p3) && (q1 || q2 || q3);
  return fill ? gap_value(d1, d2) : c;
}
// mean of the eight taps if it forms (sum of weights > 0, quotient >= 0), else `keep`; branch-free apart from the
// wave-uniform shortcut: in smooth regions all eight taps carry the full weight 4 in every lane of the wave, the divisor
// is exactly 32 and the IEEE quotient is the exact product with 1/32 (no 10-instruction division sequence)
DEV float am_select(const float (&pw)[4], const float (&pf)[4], bool enabled, float keep) {
  const float ws = __fadd_rn(__fadd_rn(__fadd_rn(pw[0], pw[1]), pw[2]), pw[3]);
  const float fs = __fadd_rn(__fadd_rn(__fadd_rn(pf[0], pf[1]), pf[2]), pf[3]);
  float d;
  if (__ballot(ws != 32.0f) == 0ull) d = fs * 0.03125f;
  else d = fs / (ws > 0 ? ws : 1.0f);
  return (enabled && ws > 0 && d >= 0) ? d : keep;
}
// (Round 6, VERDICT r05 #3: a sum-only form for waves in which every tap lies within 2 of the centre — all weights are 4 there, the mean is
// ((s_0 + s_1) + s_2) + s_3 over the pair sums, times 1/8, bit for bit — was built behind a wave-wide test on the window's extremes and
// measured: k_gap_mean_fused 194.5 against 196.7 us, nothing in the pipeline (profiles/r06_adaptive_mean_smooth_ab.txt).  Too few waves are
// smooth in all 64 lanes for the test to pay for itself; removed.)
__global__ void __launch_bounds__(256) k_gap_mean_fused(DevParams dp, const FrameInfo* __restrict__ info, const float* __restrict__ in,
                                                        float* __restrict__ out, int rows_per_band, int do_mean) {
  __shared__ float s_row[2][256 + 8], s_g2[2][256];            // s_row: 4 cells of "invalid" on either side, so the gap search reads without bounds tests
  const int frame = blockIdx.z;
  if (!info[frame].ok) return;
  const int W = dp.W, H = dp.H, gw = dp.gap_width, t = threadIdx.x;
  if (t < 8) { s_row[0][t < 4 ? t : 256 + t] = -10.0f; s_row[1][t < 4 ? t : 256 + t] = -10.0f; }
  const int u = blockIdx.x * kPostCols - kPostHaloL + t;
  const int r0 = blockIdx.y * rows_per_band, r1 = min(r0 + rows_per_band, H);
  const bool col_in = u >= 0 && u < W;
  const size_t plane = (size_t)H * W;
  const float* I = in + frame * plane;
  float* O = out + frame * plane;
  const bool store_col = t >= kPostHaloL && t < kPostHaloL + kPostCols && col_in;
  const bool mean_h_col = u >= 4 && u <= W - 4, mean_v_col = u >= 3 && u < W - 3;
  // horizontal taps in ring order: lane l of the sum takes the window columns congruent to l mod 4
  int tap[4];
#pragma unroll
  for (int l = 0; l < 4; l++) tap[l] = min(max(t - 4 + ((l - u) & 3), 0), 251);
  float g1[8], g2[8], tw[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { g1[k] = -10.0f; g2[k] = -10.0f; tw[k] = -10.0f; }
  const int y_end = r1 + 7;                                    // last row needed + 1
  // (the column's values one row ahead of their use; eight rows ahead was measured in round 5: no change, 0.193 against 0.186 ms — the pass
  // is bound by its ~220 vector instructions per pixel, seven waves a SIMD, not by the loads)
  // What the band's output rows [r0, r1) need: T and G2 of rows >= r0 - 4, G1 of rows >= r0 - 7, nothing above those.  r0 is a multiple
  // of 8 (launch_gap_mean_fused), so the band is walked in blocks of eight rows of three kinds, the phase of a row (its ring slot) a
  // compile-time constant in each: STAGE 0, rows r0-8 .. r0-1: the row fill only (row r0-8 not at all, the column fill for row r0-4 in
  // the last step); STAGE 1, rows r0 .. r0+7: no vertical mean before the last step (output row r0); STAGE 2: everything.  [Round 6:
  // until then every step ran every part, a fifth of a band's steps a mean nobody read.  The same conditions as uniform branches inside
  // one loop cost more than they saved: the rings' registers became conditional copies, 62 -> 76 VGPRs.]  Rows above the image are all
  // "invalid", which is what the rings hold to begin with: the first band starts at row 0.
  float x_next = -10.0f;
  { const int yf = r0 > 0 ? r0 - 7 : 0; if (col_in && yf < H) x_next = I[(size_t)yf * W + u]; }
  auto block = [&](int yb, auto stage_tag) {
    constexpr int STAGE = decltype(stage_tag)::value;
#pragma unroll
    for (int ph = 0; ph < 8; ph++) {
      if (STAGE == 0 && ph == 0) continue;
      const int y = yb + ph;                                   // y & 7 == ph
      const float x = x_next;
      { const int yn = y + 1; x_next = (col_in && yn >= 0 && yn < H) ? I[(size_t)yn * W + u] : -10.0f; }
      float* sr = s_row[ph & 1] + 4; float* sg = s_g2[ph & 1];
      sr[t] = x;
      if (STAGE > 0) sg[t] = g2[(ph + 4) & 7];                 // G2 of row y - 4 (computed in the previous step)
      __syncthreads();
      // ---- G1(y): gap fill along the row (elas.cpp:1122-1166) ----
      // Branch-free (gw <= 3): nearest valid pixel within gw to the left, then within what is left of gw to the right.
      g1[ph] = gap_fill3(gw, x, sr[t - 1], sr[t - 2], sr[t - 3], sr[t + 1], sr[t + 2], sr[t + 3]);
      // ---- G2(y - 3): gap fill along the column (:1204-1247) on the ring of G1 ----
      if (STAGE > 0 || ph == 7)
        g2[(ph + 5) & 7] = gap_fill3(gw, g1[(ph + 5) & 7], g1[(ph + 4) & 7], g1[(ph + 3) & 7], g1[(ph + 2) & 7], g1[(ph + 6) & 7], g1[(ph + 7) & 7], g1[ph]);
      // ---- T(y - 4): horizontal mean (:1394-1433) of the G2 row in LDS ----
      if (STAGE > 0) {
        const int yt = y - 4;
        const float c = sg[t];
        float res = c;
        if (do_mean && yt >= 3 && yt < H - 3) {               // wave-uniform
          float pw[4], pf[4];
#pragma unroll
          for (int l = 0; l < 4; l++) am_pair(sg[tap[l]], sg[tap[l] + 4], c, pw[l], pf[l]);
          res = am_select(pw, pf, mean_h_col, c);
        }
        tw[(ph + 4) & 7] = res;
      }
      // ---- out(y - 7): vertical mean (:1436-1483) over the ring of T; where none forms the pixel keeps G2 ----
      if (STAGE == 2 || (STAGE == 1 && ph == 7)) {
        const int yo = y - 7;
        float res = g2[(ph + 1) & 7];
        if (do_mean && yo >= 4 && yo <= H - 4) {              // wave-uniform
          const float c = tw[(ph + 1) & 7];
          float pw[4], pf[4];
#pragma unroll
          for (int l = 0; l < 4; l++) am_pair(tw[l], tw[l + 4], c, pw[l], pf[l]);
          res = am_select(pw, pf, mean_v_col, res);
        }
        if (store_col && yo < r1) O[(size_t)yo * W + u] = res;
      }
    }
  };
  if (r0 > 0) block(r0 - 8, std::integral_constant<int, 0>{});
  block(r0, std::integral_constant<int, 1>{});
  for (int yb = r0 + 8; yb < y_end; yb += 8) block(yb, std::integral_constant<int, 2>{});
}

// ------------------------------------------------------------------------------------------------
// Median filter (elas.cpp:1494-1560): separable 7-tap median on valid pixels of the interior
// u in [3,W-4], v in [3,H-4].  The horizontal pass writes a calloc'ed scratch image, so the scratch border
// is 0 and takes part in the vertical windows; the vertical pass tests D itself and reads the scratch.
// The reference insertion-sorts the window and takes element 3; the 4th smallest is the same value.
DEV float median7(float x0, float x1, float x2, float x3, float x4, float x5, float x6) {
  float v[7] = {x0, x1, x2, x3, x4, x5, x6};
#pragma unroll
  for (int pass = 0; pass < 4; pass++)                 // four selection passes put the 4 smallest in front
#pragma unroll
    for (int k = 6; k > pass; k--) {
      const float lo = fminf(v[k - 1], v[k]), hi = fmaxf(v[k - 1], v[k]);
      v[k - 1] = lo; v[k] = hi;
    }
  return v[3];
}
template <bool kHorizontal>
__global__ void __launch_bounds__(256) k_median(DevParams dp, const FrameInfo* __restrict__ info, const float* __restrict__ in,
                                                float* __restrict__ D, float* __restrict__ out) {
  // horizontal: in = D, out = tmp (0 outside the interior); vertical: in = tmp, out = D (interior only)
  const int u = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y, frame = blockIdx.z;
  if (u >= dp.W || !info[frame].ok) return;
  const int W = dp.W, H = dp.H;
  const size_t plane = (size_t)H * W, p = (size_t)v * W + u;
  const bool interior = u >= 3 && u < W - 3 && v >= 3 && v < H - 3;
  if (!interior) { if (kHorizontal) out[frame * plane + p] = 0.0f; return; }
  const float c = D[frame * plane + p];
  if (!(c >= 0)) { if (kHorizontal) out[frame * plane + p] = c; return; }
  const float* I = in + frame * plane + p;
  const long long st = kHorizontal ? 1 : W;
  out[frame * plane + p] = median7(I[-3 * st], I[-2 * st], I[-st], I[0], I[st], I[2 * st], I[3 * st]);
}

// ------------------------------------------------------------------------------------------------
// Node side.  convertTo(CV_8U) (point_cloud.cpp:422) = round-half-even + saturate.
DEV uint8_t f32_to_u8(float x) {
  const float r = rintf(x);
  return (uint8_t)(r < 0.f ? 0 : (r > 255.f ? 255 : (int)r));
}
__global__ void __launch_bounds__(256) k_to_u8(const float* __restrict__ D, uint8_t* __restrict__ out, long long count) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < count) out[i] = f32_to_u8(D[i]);
}

struct ScanDev {
  double Q[16], XR[9], XT[3];
  int ox, oy;
  double gp_h, gp_tan, gp_dist, fov, pi;
  int bins;
};
// pos = Q*[i+ox, j+oy, d, 1]; cam = pos.xyz/pos.w; robot = XR*cam + XT (point_cloud.cpp:237-253)
DEV bool reproject(const ScanDev& s, int i, int j, int d, double& X, double& Y, double& Z) {
  const double V0 = (double)(i + s.ox), V1 = (double)(j + s.oy), V2 = (double)d;
  double pos[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    double a = __dmul_rn(s.Q[4 * r], V0);
    a = __dadd_rn(a, __dmul_rn(s.Q[4 * r + 1], V1));
    a = __dadd_rn(a, __dmul_rn(s.Q[4 * r + 2], V2));
    a = __dadd_rn(a, s.Q[4 * r + 3]);
    pos[r] = a;
  }
  if (pos[3] == 0.0) return false;
  const double cx = pos[0] / pos[3], cy = pos[1] / pos[3], cz = pos[2] / pos[3];
  double o[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    double a = __dmul_rn(s.XR[3 * r], cx);
    a = __dadd_rn(a, __dmul_rn(s.XR[3 * r + 1], cy));
    a = __dadd_rn(a, __dmul_rn(s.XR[3 * r + 2], cz));
    o[r] = __dadd_rn(a, s.XT[r]);
  }
  X = o[0]; Y = o[1]; Z = o[2];
  return true;
}
DEV bool is_ground(const ScanDev& s, double X, double Z) {      // point_cloud.cpp:128-137
  if (X < s.gp_dist) return Z < s.gp_h;
  return Z < __dadd_rn(s.gp_h, __dmul_rn(s.gp_tan, X - s.gp_dist));
}

// cacheDisparityValues (point_cloud.cpp:104-147)
__global__ void __launch_bounds__(256) k_valid_lut(ScanDev s, int W, int H, uint8_t* __restrict__ lut) {
  const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
  if (i >= W) return;
  int d;
  for (d = 3; d <= 255; d++) {
    double X, Y, Z;
    if (!reproject(s, i, j, d, X, Y, Z)) continue;
    if (Z < 0.) continue;
    if (is_ground(s, X, Z)) continue;
    break;
  }
  lut[((size_t)j * W + i) * 2] = (uint8_t)d;       // 256 wraps to 0 like the reference's uchar store (:142)
  lut[((size_t)j * W + i) * 2 + 1] = 255;
}

// order-preserving map double -> uint64 so integer atomics implement float min/max
DEV unsigned long long enc(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __host__ inline double dec(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  union { unsigned long long u; double d; } c; c.u = b; return c.d;
}

// publishObstacleScan(Mat&) (point_cloud.cpp:213-296).  Per block: bins and the four extrema are
// reduced in LDS (64-bit integer atomics on order-encoded doubles), then merged into global memory.
// kFromCloud selects the -g flavour (publishPointCloud + publishObstacleScan(vector<Point3d>),
// point_cloud.cpp:321-352, :149-211): every pixel with d >= 2 becomes a point, points on the ground
// model are dropped, the rest are binned — instead of the LUT test of the default path.
constexpr int kScanRows = 16;
// kSgm: the disparities come from the SGM mode's winners (sgm_sweep.hip): the L/R check (k_sw_lr), the int16 map, its mono8 form
// (jn_sgm_disparity_to_u8's rounding) and the scan in ONE pass — the three-kernel tail read the int16 map back twice and the mono8 map once.
// The winners of a thread's 16 rows are requested together (a left winner, then the right image's winner it points at: two dependent
// loads, which one row at a time would pay 16 times).
template <bool kFromCloud, bool kSgm = false>
__global__ void __launch_bounds__(256) k_scan(ScanDev s, const float* __restrict__ dD, uint8_t* __restrict__ dDisp,
                                              const uint8_t* __restrict__ lut, int W, int H, unsigned long long* __restrict__ gbins,
                                              unsigned long long* __restrict__ gmeta, SgmWinners sw = SgmWinners()) {
  extern __shared__ unsigned long long lds[];      // [bins] + 4
  unsigned long long* lbins = lds;
  unsigned long long* lmeta = lds + s.bins;
  const int frame = blockIdx.z;
  for (int k = threadIdx.x; k < s.bins; k += 256) lbins[k] = ~0ull;
  if (threadIdx.x < 4) lmeta[threadIdx.x] = (threadIdx.x & 1) ? 0ull : ~0ull;   // min slots start high, max slots low
  __syncthreads();
  // One thread walks kScanRows rows of one column.  The bearing of a pixel hardly depends on its row or
  // disparity, so consecutive rows fall in the same bin: the running minimum stays in registers and reaches the
  // LDS only when the bin changes (same-address LDS atomics from a whole wave would serialise otherwise).
  const int i = blockIdx.x * 256 + threadIdx.x, j0 = blockIdx.y * kScanRows;
  unsigned long long tmin = ~0ull, tmax = 0ull, rmin = ~0ull, rmax = 0ull;
  int cur_bin = -1;
  unsigned long long cur_min = ~0ull;
  const int jend = min(j0 + kScanRows, H);
  // Phase 1: the inputs of all of the thread's rows are requested TOGETHER (16 independent loads per array; round 4 fetched one row ahead
  // and the kernel ran at one load latency per row), the mono8 map is written, and the rows whose disparity passes the LUT test (or d >= 2
  // for the -g flavour) are noted in a mask.  Phase 2 visits only those rows, in ascending order — the order the running bin minimum expects.
  uint32_t u8pk[kScanRows / 4] = {};                              // the mono8 values of this thread's rows
  uint32_t cand = 0;
  if (i < W) {
    const size_t p0 = ((size_t)frame * H + j0) * W + i;
    int dv[kScanRows];
    if constexpr (kSgm) {
      const int xk = W - 1 - i;                                   // the sweeps work on x-mirrored columns
      uint32_t e[kScanRows], m[kScanRows];
#pragma unroll
      for (int r = 0; r < kScanRows; r++) e[r] = sw.dl[((size_t)frame * H + min(j0 + r, H - 1)) * W + xk];
#pragma unroll
      for (int r = 0; r < kScanRows; r++) m[r] = sw.minr[((size_t)frame * H + min(j0 + r, H - 1)) * W + min(xk + (int)(e[r] & 0xFFFFu), W - 1)];
#pragma unroll
      for (int r = 0; r < kScanRows; r++) {
        const int d = (int)(e[r] & 0xFFFFu);
        const bool ok = sw.lr < 0 || (xk + d < W && abs(d - (int)(m[r] & 0xFFFFu)) <= sw.lr);    // x - d >= 0 and the right image's winner there agrees
        int v = ok ? (sw.subpixel ? (int)(int16_t)(e[r] >> 16) : d) : (sw.subpixel ? -16 : -1);
        if (j0 + r < H) sw.disp[p0 + (size_t)r * W] = (int16_t)v;
        if (v < 0) v = 0;
        else if (sw.subpixel) { const int q = v >> 4, f = v & 15; v = q + ((f > 8 || (f == 8 && (q & 1))) ? 1 : 0); }   // half to even, as jn_sgm_disparity_to_u8
        dv[r] = min(v, 255);
        if (j0 + r < H) dDisp[p0 + (size_t)r * W] = (uint8_t)dv[r];
      }
    } else if (dD) {
      float fd[kScanRows];
#pragma unroll
      for (int r = 0; r < kScanRows; r++) fd[r] = dD[((size_t)frame * H + min(j0 + r, H - 1)) * W + i];
#pragma unroll
      for (int r = 0; r < kScanRows; r++) { dv[r] = f32_to_u8(fd[r]); if (j0 + r < H) dDisp[p0 + (size_t)r * W] = (uint8_t)dv[r]; }
    } else {
#pragma unroll
      for (int r = 0; r < kScanRows; r++) dv[r] = dDisp[((size_t)frame * H + min(j0 + r, H - 1)) * W + i];
    }
    uint32_t lt[kScanRows];
    if (!kFromCloud) {
#pragma unroll
      for (int r = 0; r < kScanRows; r++) lt[r] = reinterpret_cast<const uint16_t*>(lut)[(size_t)min(j0 + r, H - 1) * W + i];       // :234
    }
#pragma unroll
    for (int r = 0; r < kScanRows; r++) {
      u8pk[r >> 2] |= (uint32_t)dv[r] << (8 * (r & 3));
      const bool c = kFromCloud ? dv[r] >= 2 : (dv[r] >= (int)(lt[r] & 0xFF) && dv[r] <= (int)(lt[r] >> 8));
      if (c && j0 + r < jend) cand |= 1u << r;
    }
  }
  for (uint32_t mk = cand; mk; mk &= mk - 1) {
    const int rr = __ffs((int)mk) - 1, j = j0 + rr;
    const uint32_t w = rr < 8 ? (rr < 4 ? u8pk[0] : u8pk[1]) : (rr < 12 ? u8pk[2] : u8pk[3]);
    const int d = (int)((w >> (8 * (rr & 3))) & 255u);
    bool take;
    double X = 0, Y = 0, Z = 0;
    if (kFromCloud) take = reproject(s, i, j, d, X, Y, Z) && !is_ground(s, X, Z);               // :324 (d >= 2: the mask), :166-172
    else take = reproject(s, i, j, d, X, Y, Z);                                                  // (the LUT test: the mask)
    if (take) {
      const double th = atan2(Y, X);
      const double deg = __dmul_rn(th, 180.) / s.pi;
      const double r = sqrt(__dadd_rn(__dmul_rn(Y, Y), __dmul_rn(X, X)));
      const unsigned long long et = enc(th), er = enc(r);
      tmin = min(tmin, et); tmax = max(tmax, et); rmin = min(rmin, er); rmax = max(rmax, er);
      const double kf = floor(__dmul_rn((double)s.bins, __dadd_rn(s.fov / 2., -deg)) / s.fov);   // :263
      if (kf >= 0 && kf < (double)s.bins) {
        const int k = (int)kf;
        if (k != cur_bin) {
          if (cur_bin >= 0) atomicMin(&lbins[cur_bin], cur_min);
          cur_bin = k; cur_min = er;
        } else cur_min = min(cur_min, er);
      }
    }
  }
  if (cur_bin >= 0) atomicMin(&lbins[cur_bin], cur_min);
  // extrema: butterfly inside the wave, then one LDS atomic per wave — skipped by the (many) waves in which
  // no pixel passed the test
  if (__ballot(tmin != ~0ull) != 0ull) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    tmin = min(tmin, __shfl_xor(tmin, off)); tmax = max(tmax, __shfl_xor(tmax, off));
    rmin = min(rmin, __shfl_xor(rmin, off)); rmax = max(rmax, __shfl_xor(rmax, off));
  }
  if ((threadIdx.x & 63) == 0) { atomicMin(&lmeta[0], tmin); atomicMax(&lmeta[1], tmax); atomicMin(&lmeta[2], rmin); atomicMax(&lmeta[3], rmax); }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < s.bins; k += 256)
    if (lbins[k] != ~0ull) atomicMin(&gbins[(size_t)frame * s.bins + k], lbins[k]);
  if (threadIdx.x < 4) {
    const unsigned long long x = lmeta[threadIdx.x];
    if (threadIdx.x & 1) { if (x != 0ull) atomicMax(&gmeta[frame * 4 + threadIdx.x], x); }
    else { if (x != ~0ull) atomicMin(&gmeta[frame * 4 + threadIdx.x], x); }
  }
}
__global__ void k_scan_init(int total_bins, int n, unsigned long long* gbins, unsigned long long* gmeta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < total_bins) gbins[i] = ~0ull;
  if (i < n * 4) gmeta[i] = (i & 1) ? 0ull : ~0ull;
}
__global__ void k_scan_finish(int total_bins, int n, unsigned long long* gbins, const unsigned long long* gmeta, double* meta, double* flat) {
  // flat (may be null): the cross-rig merge's packed buffer [bins of all frames | extrema of all frames, maxima negated] — written here
  // so that a batch with a communicator attached needs no separate pack launch (comm.cpp)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < total_bins) {
    const unsigned long long k = gbins[i];
    const double x = (k == ~0ull) ? 1e9 : dec(k);                              // INF of point_cloud.cpp:54
    reinterpret_cast<double*>(gbins)[i] = x;
    if (flat) flat[i] = x;
  }
  if (i < n * 4) {
    const unsigned long long k = gmeta[i];
    const double init[4] = {400., -400., 1e9, -500.};                          // point_cloud.cpp:219-220
    const bool untouched = (i & 1) ? (k == 0ull) : (k == ~0ull);
    const double x = untouched ? init[i & 3] : dec(k);
    meta[i] = x;
    if (flat) flat[total_bins + i] = (i & 1) ? -x : x;
  }
}

// Cross-rig merge (comm.cpp): bins [n][bins] and extrema [n][4] of a batch <-> one packed buffer, the two maxima of
// every frame negated (exact for doubles) so that the whole merge is a single MIN all-reduce.
__global__ void k_scan_pack(int nb, int nm, double* __restrict__ bins, double* __restrict__ meta, double* __restrict__ flat, int pack) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nb) { if (pack) flat[i] = bins[i]; else bins[i] = flat[i]; }
  else if (i < nb + nm) {
    const int j = i - nb;
    if (pack) { const double x = meta[j]; flat[i] = (j & 1) ? -x : x; }
    else { const double x = flat[i]; meta[j] = (j & 1) ? -x : x; }
  }
}

// initUndistortRectifyMap (point_cloud.cpp:553-554): for every rectified pixel the distorted source
// position.  iR = inverse(P[:, :3] * R) comes from the host; the per-pixel math is OpenCV's, in
// double, stored as float (the column walk is evaluated directly instead of by repeated addition).
struct MapDev { double iR[9], k1, k2, p1, p2, k3, fx, fy, u0, v0; };
__global__ void __launch_bounds__(256) k_undistort_map(MapDev m, int W, int H, float* __restrict__ mapx, float* __restrict__ mapy) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= W) return;
  const double fj = (double)j, fi = (double)i;
  const double _x = __dadd_rn(__dadd_rn(__dmul_rn(fj, m.iR[0]), __dmul_rn(fi, m.iR[1])), m.iR[2]);
  const double _y = __dadd_rn(__dadd_rn(__dmul_rn(fj, m.iR[3]), __dmul_rn(fi, m.iR[4])), m.iR[5]);
  const double _w = __dadd_rn(__dadd_rn(__dmul_rn(fj, m.iR[6]), __dmul_rn(fi, m.iR[7])), m.iR[8]);
  const double w = __ddiv_rn(1.0, _w), x = __dmul_rn(_x, w), y = __dmul_rn(_y, w);
  const double x2 = __dmul_rn(x, x), y2 = __dmul_rn(y, y), r2 = __dadd_rn(x2, y2), _2xy = __dmul_rn(__dmul_rn(2.0, x), y);
  const double kr = __dadd_rn(1.0, __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(m.k3, r2), m.k2), r2), m.k1), r2));
  const double u = __dadd_rn(__dmul_rn(m.fx, __dadd_rn(__dadd_rn(__dmul_rn(x, kr), __dmul_rn(m.p1, _2xy)),
                                                        __dmul_rn(m.p2, __dadd_rn(r2, __dmul_rn(2.0, x2))))), m.u0);
  const double v = __dadd_rn(__dmul_rn(m.fy, __dadd_rn(__dadd_rn(__dmul_rn(y, kr), __dmul_rn(m.p1, __dadd_rn(r2, __dmul_rn(2.0, y2)))),
                                                        __dmul_rn(m.p2, _2xy))), m.v0);
  mapx[(size_t)i * W + j] = (float)u;
  mapy[(size_t)i * W + j] = (float)v;
}

// remap, INTER_LINEAR, BORDER_CONSTANT 0 (point_cloud.cpp:440, :481)
__global__ void __launch_bounds__(256) k_remap(const uint8_t* __restrict__ src, int sw, int sh, int spitch, long long sstride,
                                               const float* __restrict__ mapx, const float* __restrict__ mapy,
                                               uint8_t* __restrict__ dst, int W, int H, int dpitch, long long dstride) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, img = blockIdx.z;
  if (x >= W) return;
  const int sx = (int)rintf(__fmul_rn(mapx[(size_t)y * W + x], 32.0f)), sy = (int)rintf(__fmul_rn(mapy[(size_t)y * W + x], 32.0f));
  const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
  const uint8_t* S = src + (long long)img * sstride;
  auto tap = [&](int xx, int yy) -> int { return (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? S[(size_t)yy * spitch + xx] : 0; };
  const int p00 = tap(ix, iy), p01 = tap(ix + 1, iy), p10 = tap(ix, iy + 1), p11 = tap(ix + 1, iy + 1);
  const int acc = (32 - fx) * (32 - fy) * p00 + fx * (32 - fy) * p01 + (32 - fx) * fy * p10 + fx * fy * p11;
  dst[(long long)img * dstride + (size_t)y * dpitch + x] = (uint8_t)((acc + 512) >> 10);
}

// Point cloud (-g, point_cloud.cpp:321-352): column-major order (i outer, j inner) with d >= 2.
__global__ void __launch_bounds__(256) k_pc_count(const uint8_t* __restrict__ disp, int W, int H, long long* __restrict__ col_count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W) return;
  int c = 0;
  for (int j = 0; j < H; j++) c += disp[(size_t)j * W + i] >= 2;
  col_count[i + 1] = c;
  if (i == 0) col_count[0] = 0;
}
__global__ void k_pc_scan(int W, long long* col_count) {   // tiny: one thread, W <= a few thousand
  if (threadIdx.x == 0 && blockIdx.x == 0) for (int i = 1; i <= W; i++) col_count[i] += col_count[i - 1];
}
__global__ void __launch_bounds__(256) k_pc_scatter(ScanDev s, const uint8_t* __restrict__ disp, int W, int H,
                                                    const long long* __restrict__ col_count, float* __restrict__ xyz) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W) return;
  long long o = col_count[i];
  for (int j = 0; j < H; j++) {
    const int d = disp[(size_t)j * W + i];
    if (d < 2) continue;
    double X, Y, Z;
    if (!reproject(s, i, j, d, X, Y, Z)) { X = Y = Z = 0; }
    xyz[3 * o] = (float)X; xyz[3 * o + 1] = (float)Y; xyz[3 * o + 2] = (float)Z; o++;
  }
}

// ================================================================================================
// launchers
static inline dim3 grid2d(int W, int H, int z) { return dim3((W + 255) / 256, H, z); }
// Kernel attributes (dynamic LDS limits) are per device.  configure_device_kernels() sets every one of them for the
// CURRENT device, once, under a lock, and reports failures; jn_elas_create / jn_device_support_filters call it before
// anything can launch, so the launchers themselves carry no lazy-initialisation state (four slot workers reaching a
// launcher together during warm-up used to race on it).
template <int PITCH> static hipError_t configure_support_pitch() {
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_support_lds<kSupportLanes, PITCH, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_support_lds<kSupportLanes, PITCH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <int FORM>
__global__ void k_arrange(const int16_t* __restrict__ list, const int32_t* __restrict__ count, int list_cap, int step, int arr_cap,
                          int arr_stride, uint16_t* __restrict__ arr, int32_t* __restrict__ arr_ok, unsigned long long* gbuf, int g_cap, ArrBounds bnd);
static bool g_arrange_compact = false;       // the 160 KB form of k_arrange is available on this device (configure_device_kernels)
hipError_t configure_device_kernels() {
  static std::mutex m;
  static uint64_t done = 0;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> guard(m);
  const uint64_t bit = 1ull << (dev & 63);
  if (done & bit) return hipSuccess;
  constexpr int WIN = 5;
  if ((e = configure_support_pitch<320>()) != hipSuccess) return e;
  if ((e = configure_support_pitch<640>()) != hipSuccess) return e;
  if ((e = configure_support_pitch<1280>()) != hipSuccess) return e;
  if ((e = configure_support_pitch<2560>()) != hipSuccess) return e;
  // 152 KB dynamic + 6 KB static: asking for the full 160 KB fails silently and the launch is then refused
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_filter_resolve<WIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_filter_resolve_big<WIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_support_filters<WIN, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_support_filters<WIN, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gap_rows_any), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_arrange<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)) != hipSuccess) return e;
  // the compact form needs 159,760 B dynamic + 72 B static; without it, sides of 8193-12288 vertices take the global-scratch form
  g_arrange_compact = hipFuncSetAttribute(reinterpret_cast<const void*>(k_arrange<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) == hipSuccess;
  (void)hipGetLastError();
  if ((e = configure_delaunay_kernel()) != hipSuccess) return e;
  done |= bit;
  return hipSuccess;
}

static ScanDev to_dev(const jn_scan_params& sp) {
  ScanDev s;
  for (int i = 0; i < 16; i++) s.Q[i] = sp.Q[i];
  for (int i = 0; i < 9; i++) s.XR[i] = sp.XR[i];
  for (int i = 0; i < 3; i++) s.XT[i] = sp.XT[i];
  s.ox = sp.crop_offset_x; s.oy = sp.crop_offset_y;
  s.gp_h = sp.gp_height_thresh; s.gp_tan = tan(sp.gp_angle_thresh); s.gp_dist = sp.gp_dist_thresh;
  s.fov = sp.fov_deg; s.pi = sp.pi_approx; s.bins = sp.bins;
  return s;
}

void launch_descriptor(hipStream_t st, const DevParams& dp, const uint8_t* I1, const uint8_t* I2, int32_t in_pitch,
                       int64_t in_stride, int n, uint4* desc) {
  const int tiles_x = (dp.W + kDescTW - 1) / kDescTW;
  const dim3 grid((tiles_x + kDescTiles - 1) / kDescTiles, (dp.H + kDescTH - 1) / kDescTH, 2 * n);
  hipLaunchKernelGGL(k_descriptor_fused, grid, dim3(256), 0, st, dp, I1, I2, in_pitch, (long long)in_stride, n, desc);
}
int plane_pitch(int W) { return (W + 8 + 63) / 64 * 64; }
size_t plane_bytes(int W, int H, int images) { return (size_t)images * 2 * H * plane_pitch(W); }
void launch_sobel_planes(hipStream_t st, const DevParams& dp, const uint8_t* I1, const uint8_t* I2, int32_t in_pitch, int64_t in_stride, int n, uint8_t* planes) {
  const int Wp = plane_pitch(dp.W);
  const dim3 grid((Wp / 4 + 63) / 64, (dp.H + 4 * kPlaneRows - 1) / (4 * kPlaneRows), 2 * n);
  hipLaunchKernelGGL(k_sobel_planes, grid, dim3(256), 0, st, dp, I1, I2, in_pitch, (long long)in_stride, n, planes, Wp);
}
template <int PITCH>
static void launch_support_pitch(hipStream_t st, const DevParams& dp, int n, const DescSrc& src, int16_t* d_can, int nseg) {
  const int per_seg = (dp.cw + nseg - 1) / nseg;
  const int threads = std::min(1024, (per_seg * kSupportLanes + 63) / 64 * 64);           // one pass over the segment's candidates when they fit
  const int blocks = (dp.ch * nseg * n + 7) / 8 * 8;      // (k_support_lds decodes its item from a flat index, XCD-aware)
  if (src.planes)
    hipLaunchKernelGGL((k_support_lds<kSupportLanes, PITCH, true>), dim3(blocks), dim3(threads), (size_t)4 * PITCH * sizeof(uint4), st, dp, n, src.ptr, src.Wp, d_can, nseg);
  else
    hipLaunchKernelGGL((k_support_lds<kSupportLanes, PITCH, false>), dim3(blocks), dim3(threads), (size_t)4 * PITCH * sizeof(uint4), st, dp, n, src.ptr, 0, d_can, nseg);
}
// columns of both images a workgroup stages for one of nseg segments of a lattice row
static int support_window(const DevParams& dp, int nseg) {
  return nseg == 1 ? dp.W : ((dp.cw + nseg - 1) / nseg) * dp.step + 2 * dp.disp_max + 8;
}
static bool launch_support_bucket(hipStream_t st, const DevParams& dp, int n, const DescSrc& desc, int16_t* d_can, int nseg, bool dry = false) {
  const int win = support_window(dp, nseg);
  if (win > 2560) return false;
  if (dry) return true;
  if (win <= 320) launch_support_pitch<320>(st, dp, n, desc, d_can, nseg);
  else if (win <= 640) launch_support_pitch<640>(st, dp, n, desc, d_can, nseg);
  else if (win <= 1280) launch_support_pitch<1280>(st, dp, n, desc, d_can, nseg);
  else launch_support_pitch<2560>(st, dp, n, desc, d_can, nseg);
  return true;
}
// dry: nothing is launched; the return value says whether an LDS bucket takes these parameters (false: the global-memory kernel would run,
// which reads materialised descriptors only)
bool launch_support(hipStream_t st, const DevParams& dp, int n, const DescSrc& desc, int16_t* d_can, bool dry) {
  // The LDS row pitch is a template constant (immediate tap offsets): the smallest bucket that holds the staged window;
  // 1280 columns = 80 KB, 2560 = the whole 160 KB.  A lattice row is cut into column segments, one workgroup each (windows
  // overlap by 2 disp_max): segments of 64 candidates (256 threads, a 320 / 640-column bucket) while the overlap stays below
  // the segment itself, else of 128.  Alone, one workgroup per row is a little quicker (it stages every descriptor once);
  // among the other slots' kernels the small workgroups find room at once instead of queueing for 80 KB of LDS, and the
  // pipelined rate gains 1.5 % (profiles/r02_d_support_split_ab.txt).  JN_SUPPORT_SPLIT=k forces k segments (1 = one per row).
  const int split = JN_HOOK_ENV("JN_SUPPORT_SPLIT") ? atoi(JN_HOOK_ENV("JN_SUPPORT_SPLIT")) : 0;
  if (split >= 1 && launch_support_bucket(st, dp, n, desc, d_can, split, dry)) return true;
  const int per = 2 * dp.disp_max + 8 <= 64 * dp.step ? 64 : 128;
  static const int max_seg = JN_HOOK_ENV("JN_SUPPORT_SEGMENTS") ? atoi(JN_HOOK_ENV("JN_SUPPORT_SEGMENTS")) : 8;
  for (int nseg = std::min(std::max(1, (dp.cw + per - 1) / per), std::max(1, max_seg)); nseg <= std::max(1, max_seg); nseg++)
    if (launch_support_bucket(st, dp, n, desc, d_can, nseg, dry)) return true;       // more segments until the window fits a bucket
  if (launch_support_bucket(st, dp, n, desc, d_can, 1, dry)) return true;
  if (!dry && !desc.planes) hipLaunchKernelGGL(k_support, dim3((dp.cw * dp.ch + 3) / 4, n), dim3(256), 0, st, dp, n, static_cast<const uint4*>(desc.ptr), d_can);
  return false;
}
// classify + resolve applies when lattice (with border) and codes fit the LDS together; JN_FILTER_WAVEFRONT=1 keeps the
// skewed-wavefront kernel (A/B and test hook), which also serves lattices that need streaming
// 2: lattice and codes fit the LDS together (k_filter_resolve), 1: only the codes do (k_filter_resolve_big), 0: neither
static int support_filters_form(const DevParams& dp, int win, int min_support) {
  static const bool wavefront_only = getenv("JN_FILTER_WAVEFRONT") != nullptr && atoi(getenv("JN_FILTER_WAVEFRONT")) != 0;
  const char* kb = JN_HOOK_ENV("JN_FILTER_LDS_KB");
  const size_t budget_bytes = (size_t)(kb ? atoi(kb) : 150) * 1024;
  const size_t codes = (size_t)dp.cw * dp.ch, both = (size_t)(dp.cw + 10) * (dp.ch + 10) * sizeof(int16_t) + codes;
  if (win != 5 || wavefront_only || min_support < 1 || min_support > 254) return 0;
  return both <= budget_bytes ? 2 : (codes <= budget_bytes ? 1 : 0);
}
bool support_filters_fast(const DevParams& dp, int win, int min_support) { return support_filters_form(dp, win, min_support) != 0; }
bool launch_support_filters(hipStream_t st, const DevParams& dp, int n, int win, int tol, int min_support, int16_t* d_can,
                            void* scratch, int16_t* list, int32_t* count, int list_cap, bool* listed) {
  if (listed) *listed = false;
  constexpr int WIN = 5, K = WIN + 1;                               // the reference's incon_window_size (elas.h:97)
  if (win != WIN) return false;                                     // other window sizes: the host stage filters
  const int form = scratch ? support_filters_form(dp, win, min_support) : 0;
  {
    const size_t need = (size_t)(dp.cw + 2 * WIN) * (dp.ch + 2 * WIN) * sizeof(int16_t) + (size_t)dp.cw * dp.ch;
    if (form == 2) {
      uint8_t* code = reinterpret_cast<uint8_t*>(scratch);          // [n][cw*ch], column-major
      hipLaunchKernelGGL(k_filter_classify<WIN>, dim3((dp.cw + 15) / 16, (dp.ch + 15) / 16, n), dim3(256), 0, st, dp, tol, min_support, d_can, code);
      static const bool fuse_list = !(JN_HOOK_ENV("JN_FUSE_LIST") && atoi(JN_HOOK_ENV("JN_FUSE_LIST")) == 0);
      const bool with_list = fuse_list && list && count && listed;
      const int rounds = JN_HOOK_ENV("JN_FILTER_ROUNDS") ? atoi(JN_HOOK_ENV("JN_FILTER_ROUNDS")) : 1;      // read per launch (A/B, tests)
      hipLaunchKernelGGL(k_filter_resolve<WIN>, dim3(n), dim3(kFilterThreads), need, st, dp, tol, min_support, d_can, code,
                         with_list ? list : static_cast<int16_t*>(nullptr), count, list_cap, rounds);
      if (with_list) *listed = true;
      return true;
    }
  }
  // LDS budget in int16 cells (JN_FILTER_LDS_KB shrinks it: a test hook that forces the streamed variant)
  const char* env = JN_HOOK_ENV("JN_FILTER_LDS_KB");
  const int budget = (env ? atoi(env) : 150) * 1024 / (int)sizeof(int16_t);
  const int cw = dp.cw, ch = dp.ch, ph = ch + 2 * WIN, pwr = cw + 2 * WIN;
  int seg_c = 0, seg_r = 0, cells = pwr * ph;
  if (cells > budget) {                                             // stream the lattice through the LDS in pieces
    seg_c = budget / ph - 2 * WIN;
    seg_r = budget / pwr;
    if (seg_c < 8 || seg_r < 1) return false;
    const int nc = (cw + seg_c - 1) / seg_c, nr = (ch + seg_r - 1) / seg_r;
    seg_c = (cw + nc - 1) / nc; seg_r = (ch + nr - 1) / nr;         // balanced pieces
    cells = max((seg_c + 2 * WIN) * ph, pwr * seg_r);
  }
  const int points = (ch + K - 1) / K;                              // points per wavefront step
  const int lanes = points <= kFilterThreads / 16 ? 16 : (points <= kFilterThreads / 8 ? 8 : 0);
  if (!lanes) return false;
  const size_t lds = (size_t)cells * sizeof(int16_t);
  int sweep = 1;
  if (form == 1) {                                                  // classify + resolve from memory; the kernel below only runs the redundancy passes
    uint8_t* code = reinterpret_cast<uint8_t*>(scratch);
    hipLaunchKernelGGL(k_filter_classify<WIN>, dim3((dp.cw + 15) / 16, (dp.ch + 15) / 16, n), dim3(256), 0, st, dp, tol, min_support, d_can, code);
    hipLaunchKernelGGL(k_filter_resolve_big<WIN>, dim3(n), dim3(kFilterThreads), ((size_t)cw * ch + 3) & ~(size_t)3, st, dp, tol, min_support, d_can, code);
    sweep = 0;
  }
  if (lanes == 16) hipLaunchKernelGGL((k_support_filters<WIN, 16>), dim3(n), dim3(kFilterThreads), lds, st, dp, tol, min_support, d_can, seg_c, seg_r, sweep);
  else             hipLaunchKernelGGL((k_support_filters<WIN, 8>), dim3(n), dim3(kFilterThreads), lds, st, dp, tol, min_support, d_can, seg_c, seg_r, sweep);
  return true;
}
// ------------------------------------------------------------------------------------------------
// The alternating-cut arrangement of a frame side's support points, on the GPU.
// Triangle's divide-and-conquer first re-partitions the (x, y)-sorted vertices by alternating median cuts
// (triangle.cpp:5514-5606, :6197-6206); csrc/delaunay.cpp computes the same array kd-style (Delaunay::arrange / split): a
// quarter of the triangulation's time on the host, whose cores are the scarce resource of the path.  The result
// depends on the coordinates alone, so it is computed here, one workgroup per frame side, and only the recursion over hulls
// stays on the host.  Level by level instead of recursively: all ranges of one depth are cut along the same axis; a cut
// keeps the first half of the defining order (x-sorted array for axis 0, y-sorted for axis 1) and stably partitions the
// other array to follow — one flag pass, one workgroup-wide prefix sum, one scatter per level.  Ranges of <= 3 vertices are
// final (Delaunay::split).  Sides whose vertices are not all distinct (two support points can share (u - d, v) in the right
// image) are left to the host, which must replay Triangle's randomised sort for them (delaunay.cpp): ok = 0.
#ifndef JN_AB_ARR_THREADS
#define JN_AB_ARR_THREADS 512
#endif
// 512 threads, not 1024 (round 6, with the orders as ranks): 75 against 85 us alone, and an eight-wave workgroup finds room among the other
// slots' kernels where a sixteen-wave one waits for half a CU to drain: host route 24.7 -> 25.5 k pairs/s (profiles/r06_arr_threads_ab.txt).
constexpr int kArrThreads = JN_AB_ARR_THREADS;
DEV void arr_sort(unsigned long long* keys, int N, int tid) {      // bitonic, N a power of two, ascending
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < N; i += kArrThreads) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = keys[i], b = keys[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[l] = a; }
        }
      }
      __syncthreads();
    }
}
DEV void arr_sort32(uint32_t* keys, uint16_t* vals, int N, int tid) {   // the same network over 32-bit keys; vals (may be null) move with them
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < N; i += kArrThreads) {
        const int l = i ^ j;
        if (l > i) {
          const uint32_t a = keys[i], b = keys[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            keys[i] = b; keys[l] = a;
            if (vals) { const uint16_t t = vals[i]; vals[i] = vals[l]; vals[l] = t; }
          }
        }
      }
      __syncthreads();
    }
}
// FORM 1: working arrays in LDS, 64-bit sort keys (sides of <= 8192 vertices: every frame of the 1280x720 workload).
// FORM 2: in LDS with 32-bit keys and a 16-bit payload array, 6 instead of 8 bytes per sorted element: 12288 vertices in 160 KB
//         (a 1920x1080 side has 11 k).  FORM 0: in this side's slice of a global scratch buffer, up to g_cap vertices.
constexpr int kArrCompactCap = 12288;
template <int FORM>
__global__ void __launch_bounds__(kArrThreads) k_arrange(const int16_t* __restrict__ list, const int32_t* __restrict__ count, int list_cap, int step,
                                                         int arr_cap, int arr_stride, uint16_t* __restrict__ arr, int32_t* __restrict__ arr_ok,
                                                         unsigned long long* gbuf, int g_cap, ArrBounds bnd) {
  extern __shared__ unsigned long long s_lds[];
  const int side = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x;
  const int n = count[frame];
  int32_t* ok = arr_ok + frame * 2 + side;
  // Working arrays: in LDS (FORM 1: sides whose vertices fit this launch's LDS, arr_cap; the others are marked for the host)
  // or — further instantiations, launched only on request for frames with more support points than that (1920x1080: 11 k) —
  // in LDS with compact sort keys (FORM 2) or in this side's slice of a global scratch buffer (FORM 0): the same code, every
  // pass then goes through L2.  Instantiations rather than a run-time choice: a pointer that may be either makes every
  // access a flat one.  FORM 0 / 2 take the sides with arr_cap < n <= g_cap: a smaller form has dealt with the others.
  if (FORM == 1) {
    if (n < 3 || n > list_cap || n > arr_cap) { if (tid == 0) *ok = 0; return; }
  } else {
    if (n <= arr_cap || n > list_cap || n > g_cap) return;
    arr_cap = g_cap;
  }
  int N = 1; while (N < n) N <<= 1;
  // layout: [sort keys, N u64 (FORM 2: N u32 + N u16); afterwards tmp, rlo, rn, sx, arr_cap u16 each] | ord, byy [arr_cap] u16 | isleft [arr_cap] u8
  int Ncap = 1; while (Ncap < arr_cap) Ncap <<= 1;
  const size_t head8 = FORM == 2 ? max((size_t)Ncap * 6, (size_t)arr_cap * 8) / 8 : (size_t)Ncap;
  unsigned long long* s_arr = FORM ? s_lds : gbuf + ((size_t)frame * 2 + side) * (((size_t)Ncap * 8 + (size_t)arr_cap * 5 + 15) / 8);
  unsigned long long* keys = s_arr;
  uint32_t* keys32 = reinterpret_cast<uint32_t*>(s_arr); uint16_t* vals = reinterpret_cast<uint16_t*>(keys32 + Ncap);
  uint16_t* tmp = reinterpret_cast<uint16_t*>(s_arr); uint16_t* rlo = tmp + arr_cap; uint16_t* rn = rlo + arr_cap; uint16_t* sx = rn + arr_cap;
  uint16_t* ord = reinterpret_cast<uint16_t*>(s_arr + head8);
  uint16_t* byy = ord + arr_cap;
  uint8_t* isleft = reinterpret_cast<uint8_t*>(byy + arr_cap);
  __shared__ int s_wave[kArrThreads / 64 + 1];
  __shared__ int s_flag;
  const int16_t* t = list + (size_t)frame * list_cap * 3;
  if (tid == 0) s_flag = 0;
  // The two orders as RANKS instead of sorts (round 6), where the caller has said what the coordinates range over (bnd) and the bitmaps
  // below fit the space the sort keys would take: a vertex's place in the (y, x) order is the number of vertices in the lattice rows above
  // its own plus those of its row to its left — with one bit per (row, column) that is a prefix population count: set the bits, count
  // the words, one workgroup-wide exclusive scan over the word counts, and every vertex reads its rank off its word.  The (x, y) order of a
  // right side (x = u - d: not on the lattice) is the same with rows and columns exchanged.  The two bitonic sorts were 55 % of the
  // kernel's time for a left side (one sort), 70 % for a right one (133 k of 237 k cycles at 3 400 vertices; 0.5 M of 0.8 M at 11 200).
  // A bit that is already set = two vertices coincide = the side goes to the host, as with the sorts.
  bool ranked = false;
  if (FORM != 0 && bnd.ny > 0) {
    const int wY = (bnd.ny + 31) >> 5, wXl = (bnd.nxl + 31) >> 5, wXr = (bnd.nxr + 31) >> 5;
    const long long NWx = (long long)bnd.nxr * wY, NWy = (long long)bnd.ny * wXr, NWl = (long long)bnd.ny * wXl;
    const long long NWmax = side == 0 ? NWl : max(NWx, NWy);
    if (NWmax * 6 + 8 <= (long long)head8 * 8) {               // uniform: bitmap words + their 16-bit prefixes in front of ord
      ranked = true;
      uint32_t* bm = reinterpret_cast<uint32_t*>(s_arr);
      // kind 0: left side, (y, x) order; 1: right side, (x, y) order; 2: right side, (y, x) order.  dst[rank] = list index.
      auto rank_pass = [&](int kind, int NW, uint16_t* dst) {
        uint16_t* pre = reinterpret_cast<uint16_t*>(bm + NW);
        auto where = [&](int i, int& w, int& b) -> bool {
          const int uc = t[3 * i], vc = t[3 * i + 1], xi = uc * step - t[3 * i + 2] - bnd.xmin;
          if (kind == 0) { w = vc * wXl + (uc >> 5); b = uc & 31; return (unsigned)vc < (unsigned)bnd.ny && (unsigned)uc < (unsigned)bnd.nxl; }
          if (kind == 1) { w = xi * wY + (vc >> 5); b = vc & 31; } else { w = vc * wXr + (xi >> 5); b = xi & 31; }
          return (unsigned)vc < (unsigned)bnd.ny && (unsigned)xi < (unsigned)bnd.nxr;
        };
        for (int k = tid; k < NW; k += kArrThreads) bm[k] = 0u;
        __syncthreads();
        for (int i = tid; i < n; i += kArrThreads) {
          int w, b;
          if (!where(i, w, b)) { s_flag = 1; continue; }       // outside what the caller declared: the host takes the side
          const uint32_t old = atomicOr(&bm[w], 1u << b);
          if ((old >> b) & 1u) s_flag = 1;                     // two vertices coincide
        }
        __syncthreads();
        if (s_flag) return false;
        const int per_w = (NW + kArrThreads - 1) / kArrThreads, w_lo = min(tid * per_w, NW), w_hi = min(w_lo + per_w, NW);
        int mine = 0;
        for (int k = w_lo; k < w_hi; k++) mine += __popc(bm[k]);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if ((tid & 63) >= off) incl += o; }
        if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
        __syncthreads();
        if (tid < 64) {
          const int wv = tid < kArrThreads / 64 ? s_wave[tid] : 0;
          int acc = wv;
#pragma unroll
          for (int off = 1; off < 16; off <<= 1) { const int o = __shfl_up(acc, off); if (tid >= off) acc += o; }
          if (tid < kArrThreads / 64) s_wave[tid] = acc - wv;  // exclusive over waves
        }
        __syncthreads();
        int run = s_wave[tid >> 6] + incl - mine;
        for (int k = w_lo; k < w_hi; k++) { pre[k] = (uint16_t)run; run += __popc(bm[k]); }
        __syncthreads();
        for (int i = tid; i < n; i += kArrThreads) {
          int w, b;
          where(i, w, b);
          dst[pre[w] + __popc(bm[w] & ((1u << b) - 1u))] = (uint16_t)i;
        }
        __syncthreads();
        return true;
      };
      bool fine;
      if (side == 0) {
        for (int i = tid; i < n; i += kArrThreads) ord[i] = (uint16_t)i;       // u-major, v ascending: the list's own order
        fine = rank_pass(0, (int)NWl, byy);
      } else {
        fine = rank_pass(1, (int)NWx, ord);
        if (fine) fine = rank_pass(2, (int)NWy, byy);
      }
      if (!fine) { if (tid == 0) *ok = 0; return; }
    }
  }
  if (!ranked) {
  // (x, y) order.  Left image: the list is written u-major, v ascending = already sorted.  Right image: x = u - d.
  if (side == 0) {
    for (int i = tid; i < n; i += kArrThreads) ord[i] = (uint16_t)i;
    __syncthreads();
  } else if (FORM == 2) {                                      // key (x, y), payload the list index; x + 32768 < 65535 always, so padding sorts last
    for (int i = tid; i < N; i += kArrThreads) {
      uint32_t k = ~0u;
      if (i < n) { const int x = t[3 * i] * step - t[3 * i + 2], vc = t[3 * i + 1]; k = ((uint32_t)(x + 32768) << 16) | (uint32_t)vc; }
      keys32[i] = k; vals[i] = (uint16_t)i;
    }
    __syncthreads();
    arr_sort32(keys32, vals, N, tid);
    for (int i = tid; i < n; i += kArrThreads) {
      ord[i] = vals[i];
      if (i > 0 && keys32[i] == keys32[i - 1]) s_flag = 1;
    }
    __syncthreads();
    if (s_flag) { if (tid == 0) *ok = 0; return; }
  } else {
    for (int i = tid; i < N; i += kArrThreads) {
      unsigned long long k = ~0ull;
      if (i < n) { const int x = t[3 * i] * step - t[3 * i + 2], vc = t[3 * i + 1]; k = ((unsigned long long)(unsigned)(x + 32768) << 32) | ((unsigned)vc << 16) | (unsigned)i; }
      keys[i] = k;
    }
    __syncthreads();
    arr_sort(keys, N, tid);
    for (int i = tid; i < n; i += kArrThreads) {
      ord[i] = (uint16_t)(keys[i] & 0xFFFFu);
      if (i > 0 && (keys[i] >> 16) == (keys[i - 1] >> 16)) s_flag = 1;          // two vertices coincide
    }
    __syncthreads();
    if (s_flag) { if (tid == 0) *ok = 0; return; }
  }
  // (y, x) order: stable sort of the x-sorted array by y
  if (FORM == 2) {                                             // key (y, position in the x order): no payload, the vertex is ord[position]
    for (int i = tid; i < N; i += kArrThreads) {
      uint32_t k = ~0u;
      if (i < n) k = ((uint32_t)(unsigned)t[3 * ord[i] + 1] << 16) | (uint32_t)i;
      keys32[i] = k;
    }
    __syncthreads();
    arr_sort32(keys32, nullptr, N, tid);
    for (int i = tid; i < n; i += kArrThreads) byy[i] = ord[keys32[i] & 0xFFFFu];
  } else {
    for (int i = tid; i < N; i += kArrThreads) {
      unsigned long long k = ~0ull;
      if (i < n) { const int v = ord[i]; k = ((unsigned long long)(unsigned)t[3 * v + 1] << 32) | ((unsigned)i << 16) | (unsigned)v; }
      keys[i] = k;
    }
    __syncthreads();
    arr_sort(keys, N, tid);
    for (int i = tid; i < n; i += kArrThreads) byy[i] = (uint16_t)(keys[i] & 0xFFFFu);
  }
  }
  __syncthreads();                                             // the key space is free now: it holds tmp, rlo, rn, sx from here on
  for (int i = tid; i < n; i += kArrThreads) { rlo[i] = 0; rn[i] = (uint16_t)n; }
  __syncthreads();
  const int per = (n + kArrThreads - 1) / kArrThreads, p_lo = min(tid * per, n), p_hi = min(p_lo + per, n);
  for (int level = 0; level < 32; level++) {
    uint16_t* def = (level & 1) ? byy : ord;
    uint16_t* oth = (level & 1) ? ord : byy;
    // flags: the first half of every active range in the defining order goes left
    int active = 0;
    for (int i = p_lo; i < p_hi; i++) {
      const int len = rn[i];
      if (len > 3) { isleft[def[i]] = (uint8_t)((i - rlo[i]) < (len >> 1)); active = 1; }
    }
    if (tid == 0) s_flag = 0;
    __syncthreads();
    if (active) s_flag = 1;
    __syncthreads();
    if (!s_flag) break;
    // exclusive prefix sum of the flags over the other order (sx), workgroup-wide
    int mine = 0;
    for (int i = p_lo; i < p_hi; i++) { const int f = rn[i] > 3 ? isleft[oth[i]] : 0; sx[i] = (uint16_t)mine; mine += f; }
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if ((tid & 63) >= off) incl += o; }
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    if (tid < 64) {
      const int w = tid < kArrThreads / 64 ? s_wave[tid] : 0;
      int acc = w;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) { const int o = __shfl_up(acc, off); if (tid >= off) acc += o; }
      if (tid < kArrThreads / 64) s_wave[tid] = acc - w;      // exclusive over waves
    }
    __syncthreads();
    const int base = s_wave[tid >> 6] + incl - mine;
    for (int i = p_lo; i < p_hi; i++) sx[i] = (uint16_t)(sx[i] + base);
    __syncthreads();
    // scatter: left vertices keep their order at the front of the range, the others follow
    for (int i = p_lo; i < p_hi; i++) {
      const int len = rn[i], lo = rlo[i];
      int to = i;
      if (len > 3) {
        const int before = sx[i] - sx[lo];                    // left vertices of this range ahead of position i
        to = isleft[oth[i]] ? lo + before : lo + (len >> 1) + (i - lo - before);
      }
      tmp[to] = oth[i];
    }
    __syncthreads();
    for (int i = p_lo; i < p_hi; i++) {
      oth[i] = tmp[i];
      const int len = rn[i], lo = rlo[i];
      if (len > 3) {                                           // children of the cut
        const int half = len >> 1;
        if (i - lo < half) rn[i] = (uint16_t)half; else { rlo[i] = (uint16_t)(lo + half); rn[i] = (uint16_t)(len - half); }
      }
    }
    __syncthreads();
  }
  uint16_t* out = arr + ((size_t)frame * 2 + side) * arr_stride;
  for (int i = tid; i < n; i += kArrThreads) out[i] = ord[i];
  if (tid == 0) *ok = 1;
}
size_t arrange_lds_bytes(int arr_cap) {
  int N = 1; while (N < arr_cap) N <<= 1;
  return (size_t)N * 8 + (size_t)arr_cap * (2 * 2 + 1) + 16;
}
static size_t arrange_compact_lds_bytes() { return (size_t)16384 * 6 + (size_t)kArrCompactCap * 5 + 16; }
size_t arrange_scratch_bytes(int n, int g_cap) { return (size_t)n * 2 * ((arrange_lds_bytes(g_cap) + 7) / 8 * 8); }
void launch_arrange(hipStream_t st, int n, const int16_t* list, const int32_t* count, int list_cap, int step, int arr_cap, int arr_stride, uint16_t* arr,
                    int32_t* arr_ok, void* gbuf, int g_cap, ArrBounds bnd) {
  hipLaunchKernelGGL(k_arrange<1>, dim3(2, n), dim3(kArrThreads), arrange_lds_bytes(arr_cap), st, list, count, list_cap, step, arr_cap, arr_stride, arr,
                     arr_ok, static_cast<unsigned long long*>(nullptr), 0, bnd);
  if (!gbuf || g_cap <= arr_cap) return;
  // sides beyond that capacity (requested by the caller: the launches overwrite their "handed back" mark): up to 12288 vertices
  // with compact keys in LDS, the rest on global scratch
  int done_to = arr_cap;
  if (g_arrange_compact && arr_cap < kArrCompactCap) {
    done_to = std::min(g_cap, kArrCompactCap);
    hipLaunchKernelGGL(k_arrange<2>, dim3(2, n), dim3(kArrThreads), arrange_compact_lds_bytes(), st, list, count, list_cap, step, arr_cap, arr_stride, arr, arr_ok,
                       static_cast<unsigned long long*>(nullptr), done_to, bnd);
  }
  if (g_cap > done_to)
    hipLaunchKernelGGL(k_arrange<0>, dim3(2, n), dim3(kArrThreads), 0, st, list, count, list_cap, step, done_to, arr_stride, arr, arr_ok,
                       static_cast<unsigned long long*>(gbuf), g_cap, bnd);
}
void launch_support_list(hipStream_t st, const DevParams& dp, int n, const int16_t* d_can, int16_t* list, int32_t* count, int cap) {
  hipLaunchKernelGGL(k_support_list, dim3(n), dim3(kFilterThreads), 0, st, dp, d_can, list, count, cap);
}
void launch_grid_clear(hipStream_t st, const DevParams& dp, int n, uint32_t* mark) {
  hipMemsetAsync(mark, 0, (size_t)n * 2 * dp.gw * dp.gh * kGridWords * sizeof(uint32_t), st);
}
void launch_grid_from_list(hipStream_t st, const DevParams& dp, int n, const int16_t* list, const int32_t* count, int cap, uint32_t* mark, uint32_t* gridbits) {
  launch_grid_clear(st, dp, n, mark);
  hipLaunchKernelGGL(k_grid_mark_list, dim3((cap + 255) / 256, n), dim3(256), 0, st, dp, list, count, cap, mark);
  hipLaunchKernelGGL(k_grid_dilate, dim3((dp.gw * dp.gh * kGridWords + 255) / 256, 2 * n), dim3(256), 0, st, dp, static_cast<const FrameInfo*>(nullptr), mark, gridbits);
}
void launch_grid(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const uint8_t* payload,
                 int64_t payload_stride, int max_sup, uint32_t* mark, uint32_t* gridbits, bool clear) {
  if (clear) launch_grid_clear(st, dp, n, mark);
  if (max_sup > 0)
    hipLaunchKernelGGL(k_grid_mark, dim3((max_sup + 255) / 256, n), dim3(256), 0, st, dp, info, payload, (long long)payload_stride, mark);
  hipLaunchKernelGGL(k_grid_dilate, dim3((dp.gw * dp.gh * kGridWords + 255) / 256, 2 * n), dim3(256), 0, st, dp, info, mark, gridbits);
}
void launch_tri_setup(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const uint8_t* payload,
                      int64_t payload_stride, int max_tri, int tri_cap, TriRec* recs) {
  if (max_tri <= 0) return;
  hipLaunchKernelGGL(k_tri_setup, dim3((max_tri + 255) / 256, n, 2), dim3(256), 0, st, dp, info, payload, (long long)payload_stride, tri_cap, recs);
}
void launch_bin_clear(hipStream_t st, const DevParams& dp, int n, int32_t* bin_count) {
  const int tiles = ((dp.W + kTileW - 1) / kTileW) * ((dp.H + kTileH - 1) / kTileH);
  hipMemsetAsync(bin_count, 0, (size_t)n * 2 * tiles * sizeof(int32_t), st);
}
void launch_bin(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, TriRec* recs, int tri_cap,
                int max_tri, int32_t* bin_count, BinEntry* bin_list, bool clear, const uint8_t* payload, int64_t payload_stride) {
  if (clear) launch_bin_clear(st, dp, n, bin_count);
  if (max_tri <= 0) return;
  const dim3 grid((max_tri + kBinTris - 1) / kBinTris, n, 2);
  // One launch less matters on a lone pair's critical path (0.283-0.296 against 0.301-0.302 ms per 640x480 pair, same box); in a batch the
  // records' FP64 plane fits on one of a workgroup's four waves make the binning workgroups longer and the pipelined rate 2 % lower
  // (22.3 against 22.8 k pairs/s): small batches only.  JN_BIN_SETUP=0 / 1 forces one form (A/B).
  static const int fuse_env = JN_HOOK_ENV("JN_BIN_SETUP") ? atoi(JN_HOOK_ENV("JN_BIN_SETUP")) : -1;
  const bool fuse = fuse_env >= 0 ? fuse_env != 0 : n <= 2;
  if (payload && fuse) { hipLaunchKernelGGL(k_bin<true>, grid, dim3(256), 0, st, dp, info, recs, tri_cap, bin_count, bin_list, payload, (long long)payload_stride); return; }
  if (payload) launch_tri_setup(st, dp, n, info, payload, payload_stride, max_tri, tri_cap, recs);
  hipLaunchKernelGGL(k_bin<false>, grid, dim3(256), 0, st, dp, info, recs, tri_cap, bin_count, bin_list, nullptr, 0ll);
}
bool dense_row_applies(const DevParams& dp) {
  if (dp.grid_size < 8) return false;                 // tiny grids: a wave would touch many cells
  for (int k = 0; k <= dp.radius; k++) if (dp.P[k] < -kCellPriorMax || dp.P[k] > kCellPriorMax) return false;   // 16-bit cost field of the keys
  return true;
}
// Planes: k_owner + k_dense_row (false when they do not take these parameters: nothing is launched then).  Materialised descriptors: k_dense.
// dry: nothing is launched; the return value says whether the plane form takes these parameters.
bool launch_dense(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const TriRec* recs, int tri_cap,
                  const int32_t* bin_count, const BinEntry* bin_list, const uint32_t* gridbits, const DescSrc& desc, int16_t* raw, bool dry, hipEvent_t ev_owner) {
  static const int xcd_order = JN_HOOK_ENV("JN_DENSE_XCD_ORDER") ? atoi(JN_HOOK_ENV("JN_DENSE_XCD_ORDER")) : 1;
  const int nbx = (dp.W + kStripW - 1) / kStripW, nby = (dp.H + kTileH - 1) / kTileH;
  const int total = nbx * nby * 2 * n;
  const int blocks = xcd_order ? (total + 7) / 8 * 8 : total;
  const size_t lds = (size_t)kTileH * (kStripW + dp.disp_max) * sizeof(uint4);   // 32.6 KB at disp_max 127, 49 KB at 255
  const unsigned long long items_half = (unsigned long long)nbx * nby * n;          // the largest dividend of k_dense_row's item decode
  const bool magic_ok = items_half * (unsigned long long)max(nbx, nby) < (1ull << 32);
  if (desc.planes) {
    if (!(dense_row_applies(dp) && magic_ok)) return false;
    if (dry) return true;
    const uint32_t nbx_magic = (uint32_t)(((1ull << 32) + nbx - 1) / nbx), nby_magic = (uint32_t)(((1ull << 32) + nby - 1) / nby);
    const size_t lds2 = lds + 2 * kDense2Slack * sizeof(uint4);
    const int tiles_x = (dp.W + kTileW - 1) / kTileW, tiles_y = (dp.H + kTileH - 1) / kTileH;
#ifdef JN_HOOKS
    static const int dbg = JN_HOOK_ENV("JN_DENSE_DBG") ? atoi(JN_HOOK_ENV("JN_DENSE_DBG")) : 0;
    static const int odbg = JN_HOOK_ENV("JN_OWNER_DBG") ? atoi(JN_HOOK_ENV("JN_OWNER_DBG")) : 0;          // profiling switches (results wrong): 1 no list loads, 2 no stores, 4 empty
    // test hooks (read per launch; results stay right): lists longer than JN_OWNER_FAST_MAX take the general loop, longer than JN_OWNER_SCAN_FROM the all-triangles scan
    const char* e1 = JN_HOOK_ENV("JN_OWNER_FAST_MAX"); const char* e2 = JN_HOOK_ENV("JN_OWNER_SCAN_FROM");
    const int fast_max = e1 ? std::min(atoi(e1), (int)kBinLds) : (int)kBinLds, scan_from = e2 ? std::min(atoi(e2), (int)kBinCap) : (int)kBinCap;
#define JN_OWNER_HOOK_ARGS , odbg, fast_max, scan_from
#define JN_DENSE_HOOK_ARGS , dbg
#else
#define JN_OWNER_HOOK_ARGS
#define JN_DENSE_HOOK_ARGS
#endif
    hipLaunchKernelGGL(k_owner, dim3((tiles_x + 3) / 4, tiles_y, 2 * n), dim3(256), 0, st, dp, info, recs, tri_cap, bin_count, bin_list, reinterpret_cast<uint16_t*>(raw) JN_OWNER_HOOK_ARGS);
    if (ev_owner) hipEventRecord(ev_owner, st);                // (timing of the matcher proper, jn_elas_kernel_time)
    const uint8_t* pl = static_cast<const uint8_t*>(desc.ptr);
    if (dp.disp_max < 128)
      hipLaunchKernelGGL(k_dense_row<4>, dim3(blocks), dim3(kDenseThreads), lds2, st, dp, n, info, gridbits, pl, desc.Wp, raw, nbx, nby, xcd_order, nbx_magic, nby_magic JN_DENSE_HOOK_ARGS);
    else
      hipLaunchKernelGGL(k_dense_row<8>, dim3(blocks), dim3(kDenseThreads), lds2, st, dp, n, info, gridbits, pl, desc.Wp, raw, nbx, nby, xcd_order, nbx_magic, nby_magic JN_DENSE_HOOK_ARGS);
    return true;
  }
  if (!dry && !desc.planes)
    hipLaunchKernelGGL(k_dense, dim3(blocks), dim3(kDenseThreads), lds, st, dp, n, info, recs, tri_cap, bin_count, bin_list, gridbits,
                       static_cast<const uint4*>(desc.ptr), raw, nbx, nby, xcd_order);
  return false;
}
void launch_lr(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const int16_t* raw, float* D1, float* D2) {
  hipLaunchKernelGGL(k_lr, dim3(dp.H, n), dim3(256), (size_t)2 * dp.W * sizeof(int16_t), st, dp, info, raw, D1, D2);
}
void launch_speckle(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, float* D, int32_t* label, int32_t* size,
                    void* scratch) {
  // run lists live in the (currently idle) float scratch image: 2 x uint16 [n*H][W/2 + 2] + int32 [n*H] < n*H*W*4 bytes
  RunLists runs;
  runs.pitch = (dp.W / 2 + 2) & ~1;                          // a row of W pixels holds at most ceil(W/2) runs
  runs.H = dp.H;
  runs.base = reinterpret_cast<uint8_t*>(scratch);
  runs.frame_bytes = (size_t)dp.H * dp.W * sizeof(float);    // 2 x uint16 [H][W/2 + 2] + int32 [H] < H*W*4 bytes
  const dim3 gr((dp.H + 3) / 4, n);                          // one wave per image row
  hipLaunchKernelGGL(k_ccl_rows, gr, dim3(256), 0, st, dp, info, D, label, size, runs);
  // enough waves to fill the GPU even for a lone small pair: split the rows of the merge pass into column segments
  const int row_waves = dp.H * n, chunks = (dp.W + 63) / 64;
  const int segs = row_waves >= 4096 ? 1 : max(1, min(chunks, 4096 / max(row_waves, 1)));
  hipLaunchKernelGGL(k_ccl_merge, dim3((dp.H * segs + 3) / 4, n), dim3(256), 0, st, dp, info, D, label, segs);
  hipLaunchKernelGGL(k_ccl_count, gr, dim3(256), 0, st, dp, info, label, size, runs);
  hipLaunchKernelGGL(k_ccl_apply, gr, dim3(256), 0, st, dp, info, D, label, size, runs);
}
// launch_lr(raw -> D, D2) followed by launch_speckle(D) with the first two kernels as one (k_lr_ccl_rows)
void launch_lr_speckle(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const int16_t* raw, float* D, float* D2, int32_t* label, int32_t* size,
                       void* scratch) {
  static const int fuse = getenv("JN_LR_CCL_FUSED") ? atoi(getenv("JN_LR_CCL_FUSED")) : 1;       // 0: the two kernels (A/B)
  if (!fuse) { launch_lr(st, dp, n, info, raw, D, D2); launch_speckle(st, dp, n, info, D, label, size, scratch); return; }
  RunLists runs;
  runs.pitch = (dp.W / 2 + 2) & ~1;
  runs.H = dp.H;
  runs.base = reinterpret_cast<uint8_t*>(scratch);
  runs.frame_bytes = (size_t)dp.H * dp.W * sizeof(float);
  const size_t lds = (size_t)2 * ((dp.W + 1) & ~1) * sizeof(int16_t) + (size_t)dp.W * sizeof(float);
  hipLaunchKernelGGL(k_lr_ccl_rows, dim3(dp.H, n), dim3(256), lds, st, dp, info, raw, D, D2, label, size, runs);
  const dim3 gr((dp.H + 3) / 4, n);
  const int row_waves = dp.H * n, chunks = (dp.W + 63) / 64;
  const int segs = row_waves >= 4096 ? 1 : max(1, min(chunks, 4096 / max(row_waves, 1)));
  hipLaunchKernelGGL(k_ccl_merge, dim3((dp.H * segs + 3) / 4, n), dim3(256), 0, st, dp, info, D, label, segs);
  hipLaunchKernelGGL(k_ccl_count, gr, dim3(256), 0, st, dp, info, label, size, runs);
  hipLaunchKernelGGL(k_ccl_apply, gr, dim3(256), 0, st, dp, info, D, label, size, runs);
}
void launch_gap(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, float* D, float* tmp) {
  const dim3 g = grid2d(dp.W, dp.H, n);
  if (dp.add_corners || dp.gap_width > 64) {                 // any width, border extrapolation (MIDDLEBURY preset)
    const int wpb = (size_t)4 * 3 * dp.W * sizeof(float) <= 150 * 1024 ? 4 : 1;     // rows per workgroup: three W-sized LDS arrays each
    hipLaunchKernelGGL(k_gap_rows_any, dim3((dp.H + wpb - 1) / wpb, n), dim3(64 * wpb), (size_t)wpb * 3 * dp.W * sizeof(float), st, dp, info, D, tmp);
    hipLaunchKernelGGL(k_gap_cols_any, dim3((dp.W + 255) / 256, n), dim3(256), 0, st, dp, info, tmp, D);
    return;
  }
  if ((dp.W & 3) == 0) {
    const dim3 g4((dp.W / 4 + 63) / 64, dp.H, n);
    hipLaunchKernelGGL(k_gap4<true>, g4, dim3(64), 0, st, dp, info, D, tmp);
    hipLaunchKernelGGL(k_gap4<false>, g4, dim3(64), 0, st, dp, info, tmp, D);
    return;
  }
  hipLaunchKernelGGL(k_gap<true>, g, dim3(256), 0, st, dp, info, D, tmp);
  hipLaunchKernelGGL(k_gap<false>, g, dim3(256), 0, st, dp, info, tmp, D);
}
void launch_adaptive_mean(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, float* D, float* tmp) {
  const dim3 g = grid2d(dp.W, dp.H, n);
  if ((dp.W & 3) == 0) hipLaunchKernelGGL(k_adaptive_mean_h4, dim3((dp.W / 4 + 63) / 64, dp.H, n), dim3(64), 0, st, dp, info, D, tmp);
  else hipLaunchKernelGGL(k_adaptive_mean_h, g, dim3(256), 0, st, dp, info, D, tmp);
  hipLaunchKernelGGL(k_adaptive_mean_v, dim3((dp.W + 255) / 256, (dp.H + kAmRows - 1) / kAmRows, n), dim3(256), 0, st, dp, info, tmp, D);
}
__global__ void __launch_bounds__(256) k_copy_ok(const FrameInfo* __restrict__ info, const float4* __restrict__ src, float4* __restrict__ dst, long long per_frame4) {
  const int frame = blockIdx.y;
  if (!info[frame].ok) return;                               // frames that failed keep the caller's values
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < per_frame4) dst[frame * per_frame4 + i] = src[frame * per_frame4 + i];
}
void launch_copy_ok(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const float* src, float* dst) {
  const long long px = (long long)dp.W * dp.H;
  if ((px & 3) == 0) {
    hipLaunchKernelGGL(k_copy_ok, dim3((unsigned)((px / 4 + 255) / 256), n), dim3(256), 0, st, info, reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), px / 4);
    return;
  }
  for (int f = 0; f < n; f++) hipMemcpyAsync(dst + f * px, src + f * px, px * sizeof(float), hipMemcpyDeviceToDevice, st);   // odd sizes: plain copies (all frames)
}
bool gap_mean_fusable(const DevParams& dp, int n) {
  const int enabled = getenv("JN_POST_FUSED") ? atoi(getenv("JN_POST_FUSED")) : 1;      // read per batch: tests and A/B runs switch it
  // The fused pass walks a band of rows serially (~100 us whatever the batch); a lone pair or a small batch is quicker
  // through the four short full-width kernels.  Same results either way.
  const long long min_pixels = getenv("JN_POST_FUSED_MIN_PIXELS") ? atoll(getenv("JN_POST_FUSED_MIN_PIXELS")) : 10000000ll;
  return enabled && !dp.add_corners && dp.gap_width <= 3 && dp.W >= 16 && dp.H >= 16 && (long long)n * dp.W * dp.H >= min_pixels;
}
void launch_gap_mean_fused(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const float* in, float* out, bool mean) {
  // rows per band: flat between 40 and 120 (0.209 / 0.206 / 0.205 / 0.202 / 0.197 / 0.205 / 0.214 ms at 40 / 48 / 60 / 72 / 80 / 90 / 120 rows, 720p batch 32,
  // scripts/post_band_sweep.sh): fewer halo rows against fewer workgroups
  static const int band_rows = JN_HOOK_ENV("JN_POST_BAND") ? atoi(JN_HOOK_ENV("JN_POST_BAND")) : 80;
  // (a band's first row is a multiple of 8: the kernel's blocks of eight rows start there)
  const int bands0 = (dp.H + band_rows - 1) / band_rows, rows = ((dp.H + bands0 - 1) / bands0 + 7) & ~7, bands = (dp.H + rows - 1) / rows;
  hipLaunchKernelGGL(k_gap_mean_fused, dim3((dp.W + kPostCols - 1) / kPostCols, bands, n), dim3(256), 0, st, dp, info, in, out, rows, mean ? 1 : 0);
}
void launch_lr_sub(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const int16_t* raw, float* D1, float* D2) {
  hipLaunchKernelGGL(k_lr_sub, dim3(dp.H / 2, n), dim3(256), (size_t)2 * (dp.W / 2) * sizeof(int16_t), st, dp, info, raw, D1, D2);
}
void launch_adaptive_mean_sub(hipStream_t st, const DevParams& dph, int n, const FrameInfo* info, float* D, float* tmp) {
  const dim3 g = grid2d(dph.W, dph.H, n);
  hipLaunchKernelGGL(k_adaptive_mean_sub<false>, g, dim3(256), 0, st, dph, info, D, tmp);
  hipLaunchKernelGGL(k_adaptive_mean_sub<true>, g, dim3(256), 0, st, dph, info, tmp, D);
}
void launch_median(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, float* D, float* tmp) {
  const dim3 g = grid2d(dp.W, dp.H, n);
  hipLaunchKernelGGL(k_median<true>, g, dim3(256), 0, st, dp, info, D, D, tmp);
  hipLaunchKernelGGL(k_median<false>, g, dim3(256), 0, st, dp, info, tmp, D, D);
}
void launch_to_u8(hipStream_t st, const float* D, uint8_t* out, int64_t count) {
  hipLaunchKernelGGL(k_to_u8, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, D, out, (long long)count);
}
void launch_valid_lut(hipStream_t st, const jn_scan_params& sp, int W, int H, uint8_t* lut) {
  hipLaunchKernelGGL(k_valid_lut, grid2d(W, H, 1), dim3(256), 0, st, to_dev(sp), W, H, lut);
}
void launch_scan(hipStream_t st, const jn_scan_params& sp, int n, const float* dD, uint8_t* dDisp, const uint8_t* lut,
                 int W, int H, double* bins, double* meta, unsigned long long* scratch, double* flat, const SgmWinners* sgm) {
  const ScanDev s = to_dev(sp);
  unsigned long long* gb = reinterpret_cast<unsigned long long*>(bins);
  const int total = n * s.bins, m = total > n * 4 ? total : n * 4;
  hipLaunchKernelGGL(k_scan_init, dim3((m + 255) / 256), dim3(256), 0, st, total, n, gb, scratch);
  const dim3 sg((W + 255) / 256, (H + kScanRows - 1) / kScanRows, n);
  if (sgm) hipLaunchKernelGGL((k_scan<false, true>), sg, dim3(256), (s.bins + 4) * sizeof(unsigned long long), st, s, nullptr, dDisp, lut, W, H, gb, scratch, *sgm);
  else if (lut) hipLaunchKernelGGL((k_scan<false>), sg, dim3(256), (s.bins + 4) * sizeof(unsigned long long), st, s, dD, dDisp, lut, W, H, gb, scratch, SgmWinners());
  else     hipLaunchKernelGGL((k_scan<true>), sg, dim3(256), (s.bins + 4) * sizeof(unsigned long long), st, s, dD, dDisp, lut, W, H, gb, scratch, SgmWinners());
  hipLaunchKernelGGL(k_scan_finish, dim3((m + 255) / 256), dim3(256), 0, st, total, n, gb, scratch, meta, flat);
}
void launch_scan_pack(hipStream_t st, int n, int bins, double* dBins, double* dMeta, double* flat, bool pack) {
  const int nb = n * bins, nm = n * 4;
  hipLaunchKernelGGL(k_scan_pack, dim3((nb + nm + 255) / 256), dim3(256), 0, st, nb, nm, dBins, dMeta, flat, pack ? 1 : 0);
}
void launch_undistort_map(hipStream_t st, const double iR[9], const double K[9], const double D[5], int W, int H, float* mapx, float* mapy) {
  MapDev m;
  for (int i = 0; i < 9; i++) m.iR[i] = iR[i];
  m.k1 = D[0]; m.k2 = D[1]; m.p1 = D[2]; m.p2 = D[3]; m.k3 = D[4];
  m.fx = K[0]; m.fy = K[4]; m.u0 = K[2]; m.v0 = K[5];
  hipLaunchKernelGGL(k_undistort_map, grid2d(W, H, 1), dim3(256), 0, st, m, W, H, mapx, mapy);
}
void launch_remap(hipStream_t st, int n, const uint8_t* src, int sw, int sh, int spitch, int64_t sstride, const float* mapx,
                  const float* mapy, uint8_t* dst, int W, int H, int dpitch, int64_t dstride) {
  hipLaunchKernelGGL(k_remap, grid2d(W, H, n), dim3(256), 0, st, src, sw, sh, spitch, (long long)sstride, mapx, mapy, dst, W, H, dpitch,
                     (long long)dstride);
}
void launch_point_cloud(hipStream_t st, const jn_scan_params& sp, const uint8_t* disp, int W, int H, float* xyz, long long* col_count) {
  const ScanDev s = to_dev(sp);
  hipLaunchKernelGGL(k_pc_count, dim3((W + 255) / 256), dim3(256), 0, st, disp, W, H, col_count);
  hipLaunchKernelGGL(k_pc_scan, dim3(1), dim3(64), 0, st, W, col_count);
  hipLaunchKernelGGL(k_pc_scatter, dim3((W + 255) / 256), dim3(256), 0, st, s, disp, W, H, col_count, xyz);
}

}  // namespace jnav
