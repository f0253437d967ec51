// Probe for gfx950: what does FETCH_SIZE count for loads of 4, 8 and 16 bytes per lane?  The guide's HBM section says the
// counter has to be DOUBLED on gfx950 (it was written for 16-byte-per-lane streaming loads); the ELAS kernels issue 4-byte-per-lane buffer
// loads.  Each kernel below streams the same 1 GiB buffer exactly once (every byte by exactly one lane, fully coalesced) with
// raw_buffer_load of one width and keeps a checksum alive; run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` the counted bytes over the
// known bytes IS the correction factor for that width.   scripts/probes/fetch_size_probe.sh runs it and prints the ratios.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <int BYTES>
__global__ void __launch_bounds__(256) k_stream(const uint8_t* __restrict__ src, size_t bytes_per_block, uint32_t* __restrict__ out) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src + (size_t)blockIdx.x * bytes_per_block), 0, (int)bytes_per_block, 0x00020000);
  uint32_t acc = 0;
  for (int off = threadIdx.x * BYTES; off < (int)bytes_per_block; off += 256 * BYTES) {
    if (BYTES == 4) acc += __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    if (BYTES == 8) { const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); acc += v.x ^ v.y; }
    if (BYTES == 16) { const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); acc += v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;            // (keeps the loads; practically never true)
}
int main() {
  const size_t total = 1ull << 30, per_block = 1 << 20;      // 1 GiB, 1 MiB a block: 1024 blocks
  uint8_t* d; uint32_t* o;
  if (hipMalloc(&d, total) != hipSuccess || hipMalloc(&o, 4096 * 4) != hipSuccess) return 1;
  (void)hipMemset(d, 1, total);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++) {
    k_stream<4><<<total / per_block, 256>>>(d, per_block, o);
    k_stream<8><<<total / per_block, 256>>>(d, per_block, o);
    k_stream<16><<<total / per_block, 256>>>(d, per_block, o);
    (void)hipDeviceSynchronize();
  }
  printf("streamed %zu bytes per kernel\n", total);
  return 0;
}
