// Probe for gfx950: what happens to LDS stores WIDER than their alignment?  k_delaunay's triangle records were 6 bytes
// (three halfwords), so odd records start in the middle of a dword; without `volatile` the compiler merged the three halfword stores of a
// fresh record into ds_write_b32 / ds_write_b64 at 2-byte alignment and the two-vertex leaves came out wrong on the device.  This probe
// issues such stores by hand (inline asm, so the compiler neither splits nor reorders them) at byte offsets 0..15 from a 16-byte boundary,
// reads the bytes back with byte loads and reports, per width and offset, whether memory holds what an unaligned store should leave.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/lds_misaligned_store_probe.hip -o /tmp/lds_mis && /tmp/lds_mis
// It also compiles the pattern the kernel had (three adjacent uint16 stores through a non-volatile LDS pointer at an odd record) and shows
// what the compiler made of it and whether the result is right: hardware or compiler.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef __attribute__((address_space(3))) uint8_t lds_u8;
__global__ void __launch_bounds__(64) k_store(uint8_t* out, int width, int off) {
  __shared__ __attribute__((aligned(16))) uint8_t sm[64];
  if (threadIdx.x < 64) sm[threadIdx.x] = 0xEE;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = (uint32_t)(uintptr_t)(lds_u8*)sm + 16 + off;
    const uint32_t d0 = 0x44332211u, d1 = 0x88776655u, d2 = 0xccbbaa99u, d3 = 0x00ffeeddu;
    if (width == 2) asm volatile("ds_write_b16 %0, %1\ns_waitcnt lgkmcnt(0)" :: "v"(a), "v"(d0) : "memory");
    if (width == 4) asm volatile("ds_write_b32 %0, %1\ns_waitcnt lgkmcnt(0)" :: "v"(a), "v"(d0) : "memory");
    if (width == 8) { const uint64_t d = ((uint64_t)d1 << 32) | d0; asm volatile("ds_write_b64 %0, %1\ns_waitcnt lgkmcnt(0)" :: "v"(a), "v"(d) : "memory"); }
    if (width == 12) {
      typedef uint32_t u3 __attribute__((ext_vector_type(3)));
      const u3 d = {d0, d1, d2};
      asm volatile("ds_write_b96 %0, %1\ns_waitcnt lgkmcnt(0)" :: "v"(a), "v"(d) : "memory");
    }
    if (width == 16) {
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      const u4 d = {d0, d1, d2, d3};
      asm volatile("ds_write_b128 %0, %1\ns_waitcnt lgkmcnt(0)" :: "v"(a), "v"(d) : "memory");
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) out[threadIdx.x] = sm[threadIdx.x];
}
// the kernel's pattern: record t of 6 bytes, three halfword stores through a plain (non-volatile) LDS pointer; noinline keeps it visible in the ISA
struct Rec { uint16_t a, b, c; };
__global__ void __launch_bounds__(64) k_pattern(uint8_t* out, int t, int v0, int v1, int v2) {
  __shared__ __attribute__((aligned(16))) uint16_t rec[32];
  if (threadIdx.x < 32) rec[threadIdx.x] = 0xEEEE;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint16_t* r = rec + 3 * t;
    r[0] = (uint16_t)v0; r[1] = (uint16_t)v1; r[2] = (uint16_t)v2;
    uint16_t* q = rec + 3 * (t + 1);                      // the neighbouring record right behind it, as fresh() + fresh() of a leaf
    q[0] = (uint16_t)(v0 + 1); q[1] = (uint16_t)(v1 + 1); q[2] = (uint16_t)(v2 + 1);
  }
  __syncthreads();
  if (threadIdx.x < 64) out[threadIdx.x] = reinterpret_cast<uint8_t*>(rec)[threadIdx.x];
}
int main() {
  uint8_t* d; (void)hipMalloc(&d, 64);
  uint8_t h[64];
  const uint8_t pat[16] = {0x11, 0x22, 0x33, 0x44, 0x55, 0x66, 0x77, 0x88, 0x99, 0xaa, 0xbb, 0xcc, 0xdd, 0xee, 0xff, 0x00};
  for (int width : {2, 4, 8, 12, 16}) {
    printf("ds_write_b%-3d offset:", width * 8);
    for (int off = 0; off < 16; off++) {
      k_store<<<1, 64>>>(d, width, off); (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
      uint8_t want[64]; memset(want, 0xEE, 64); memcpy(want + 16 + off, pat, width);
      const bool ok = memcmp(h, want, 64) == 0;
      int wrote = 0; for (int i = 0; i < 64; i++) wrote += h[i] != 0xEE;
      // where did the bytes land, if not where an unaligned store would put them
      int first = -1; for (int i = 0; i < 64; i++) if (h[i] != 0xEE) { first = i - 16; break; }
      if (ok) printf(" %2d:ok", off); else printf(" %2d:WRONG(%dB@%d)", off, wrote, first);
    }
    printf("\n");
  }
  for (int t : {0, 1, 2, 3}) {
    k_pattern<<<1, 64>>>(d, t, 0x1111, 0x2222, 0x3333); (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    uint16_t want[32]; for (auto& w : want) w = 0xEEEE;
    want[3 * t] = 0x1111; want[3 * t + 1] = 0x2222; want[3 * t + 2] = 0x3333; want[3 * t + 3] = 0x1112; want[3 * t + 4] = 0x2223; want[3 * t + 5] = 0x3334;
    printf("compiler-merged halfword stores, 6-byte records %d and %d (byte offset %2d): %s\n", t, t + 1, 6 * t, memcmp(h, want, 64) == 0 ? "right" : "WRONG");
  }
  return 0;
}
