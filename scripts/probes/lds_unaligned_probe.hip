// Probe for gfx950: what does a misaligned ds_read_b64 cost?  Lane l reads 8 bytes at byte offset 32 * (l >> 4) + (l & 15) * STEP + MIS + 4 k
// (the access pattern of the row sweeps' cost operands when the input row sits in LDS).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/lds_unaligned_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, int step, int mis) {
  __shared__ __attribute__((aligned(16))) uint8_t sm[4][1024];
  for (int i = threadIdx.x; i < 4096; i += 256) (&sm[0][0])[i] = (uint8_t)(i * 7);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&sm[w][0] + 32 * (lane >> 4) + (lane & 15) * step + mis;
  uint64_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
  u32x4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0;
  uint64_t acc = 0;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0)
      asm volatile("ds_read_b64 %0, %8\nds_read_b64 %1, %8 offset:4\nds_read_b64 %2, %8 offset:8\nds_read_b64 %3, %8 offset:12\n"
                   "ds_read_b64 %4, %8 offset:16\nds_read_b64 %5, %8 offset:20\nds_read_b64 %6, %8 offset:24\nds_read_b64 %7, %8 offset:28\ns_waitcnt lgkmcnt(0)"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a) : "memory");
    if (MODE == 1)
      asm volatile("ds_read2_b32 %0, %8 offset1:1\nds_read2_b32 %1, %8 offset0:1 offset1:2\nds_read2_b32 %2, %8 offset0:2 offset1:3\nds_read2_b32 %3, %8 offset0:3 offset1:4\n"
                   "ds_read2_b32 %4, %8 offset0:4 offset1:5\nds_read2_b32 %5, %8 offset0:5 offset1:6\nds_read2_b32 %6, %8 offset0:6 offset1:7\nds_read2_b32 %7, %8 offset0:7 offset1:8\ns_waitcnt lgkmcnt(0)"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a) : "memory");
    if (MODE == 2)      // 3 x b128 (aligned windows), the lane would shift afterwards
      asm volatile("ds_read_b128 %0, %3\nds_read_b128 %1, %3 offset:16\nds_read_b128 %2, %3 offset:32\ns_waitcnt lgkmcnt(0)"
                   : "=v"(q0), "=v"(q1), "=v"(q2) : "v"(a & ~15u) : "memory");
    acc += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + q0.x + q1.y + q2.z;
  }
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)acc + (uint32_t)(acc >> 32);
}
template <int MODE>
static void run(const char* name, uint32_t* o, int step, int mis, int per) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2048, blocks = 256 * 8;
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) { (void)hipEventRecord(e0); k<MODE><<<blocks, 256>>>(o, iters, step, mis); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); }
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double winst = (double)blocks * 4 * iters * per;
  printf("%-50s step %d mis %d: %.3f ms = %.1f cycles per LDS instruction per CU at 2.4 GHz\n", name, step, mis, ms, 256.0 * 2.4e9 / (winst / ms * 1e3));
}
int main() {
  uint32_t* o; (void)hipMalloc(&o, 4 * 256 * 2048 * 4);
  run<0>("ds_read_b64 x8 (lanes 8 bytes apart, aligned)", o, 8, 0, 8);
  run<0>("ds_read_b64 x8 (lanes 4 bytes apart)", o, 4, 0, 8);
  run<0>("ds_read_b64 x8 (lanes 1 byte apart)", o, 1, 0, 8);
  run<0>("ds_read_b64 x8 (lanes 1 byte apart, +1)", o, 1, 1, 8);
  run<1>("ds_read2_b32 x8 (lanes 4 bytes apart)", o, 4, 0, 8);
  run<1>("ds_read2_b32 x8 (all lanes of a row same dword)", o, 0, 0, 8);
  run<2>("ds_read_b128 x3 (16-byte aligned windows)", o, 1, 0, 3);
  return 0;
}
