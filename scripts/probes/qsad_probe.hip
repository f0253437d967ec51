// Probe for gfx950: semantics of v_qsad_pk_u16_u8 / v_mqsad_pk_u16_u8 and their issue rate next to v_sad_u8.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/qsad_probe.hip -o gpurun_out/qsad_probe && gpurun_out/qsad_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_sem(uint64_t* out) {
  const uint64_t s0 = 0x0807060504030201ull;          // bytes 1..8
  const uint32_t ref = 0x0A000005u;                    // bytes 5,0,0,10
  out[0] = __builtin_amdgcn_qsad_pk_u16_u8(s0, ref, 0ull);
  out[1] = __builtin_amdgcn_mqsad_pk_u16_u8(s0, ref, 0ull);
  out[2] = __builtin_amdgcn_mqsad_pk_u16_u8(0x0000000000000000ull, 0x01010101u, 0x0001000200030004ull);
}
template <int MODE>
__global__ void k_rate(uint32_t* out, int iters) {
  uint64_t a0 = threadIdx.x, a1 = threadIdx.x * 3, a2 = 7, a3 = 9;
  uint32_t b0 = threadIdx.x, b1 = 1, b2 = 2, b3 = 3;
  const uint64_t s = 0x1122334455667788ull + threadIdx.x;
  const uint32_t r = 0x01020304u + blockIdx.x;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {
      a0 = __builtin_amdgcn_qsad_pk_u16_u8(s, r, a0); a1 = __builtin_amdgcn_qsad_pk_u16_u8(s, r, a1);
      a2 = __builtin_amdgcn_qsad_pk_u16_u8(s, r, a2); a3 = __builtin_amdgcn_qsad_pk_u16_u8(s, r, a3);
    } else if (MODE == 1) {
      a0 = __builtin_amdgcn_mqsad_pk_u16_u8(s, r, a0); a1 = __builtin_amdgcn_mqsad_pk_u16_u8(s, r, a1);
      a2 = __builtin_amdgcn_mqsad_pk_u16_u8(s, r, a2); a3 = __builtin_amdgcn_mqsad_pk_u16_u8(s, r, a3);
    } else {
      b0 = __builtin_amdgcn_sad_u8((uint32_t)s, r, b0); b1 = __builtin_amdgcn_sad_u8((uint32_t)s, r, b1);
      b2 = __builtin_amdgcn_sad_u8((uint32_t)s, r, b2); b3 = __builtin_amdgcn_sad_u8((uint32_t)s, r, b3);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3) + b0 + b1 + b2 + b3;
}
int main() {
  uint64_t* d; hipMalloc(&d, 64); k_sem<<<1, 1>>>(d); uint64_t h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("qsad  %016llx   (window bytes i..i+3 of 1..8 vs 5,0,0,10)\nmqsad %016llx\nmqsad zero-src/acc %016llx\n", (unsigned long long)h[0], (unsigned long long)h[1], (unsigned long long)h[2]);
  uint32_t* o; hipMalloc(&o, 4 * 256 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096, blocks = 256 * 16;
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (mode == 0) k_rate<0><<<blocks, 256>>>(o, iters); else if (mode == 1) k_rate<1><<<blocks, 256>>>(o, iters); else k_rate<2><<<blocks, 256>>>(o, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * 4 * iters * 4;    // wave instructions
    printf("%s: %.3f ms, %.1f G wave-instr/s (614 G/s = one per 4 cycles per SIMD at 2.4 GHz)\n", mode == 0 ? "qsad_pk_u16_u8 " : mode == 1 ? "mqsad_pk_u16_u8" : "sad_u8         ", ms, winst / ms / 1e6);
  }
  return 0;
}
