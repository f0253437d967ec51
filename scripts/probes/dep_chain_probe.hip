// Probe for gfx950: how many cycles between DEPENDENT vector instructions of one wave?  One wave per SIMD (or W waves per SIMD), C independent
// chains of v_pk_max_u16 / v_pk_maximum3_f16 / v_pk_sub_u16 per wave: time per instruction as a function of C and W tells the latency a
// dependent instruction waits for and how many waves / chains it takes to keep a SIMD's vector pipeline full.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/dep_chain_probe.hip -o /tmp/dep_probe && /tmp/dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int CHAINS, int SALU_MIX>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
  uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 + 7, a3 = a0 ^ 9, a4 = a0 + 11, a5 = a0 * 5, a6 = a0 + 13, a7 = a0 ^ 21;
  const uint32_t b = 0x00030001u + blockIdx.x;
  int sacc = blockIdx.x;
  for (int i = 0; i < iters; i++) {
    // 8 instructions per iteration, spread over CHAINS independent chains
    if (CHAINS == 1)
      asm volatile("v_pk_max_u16 %0, %0, %1\nv_pk_sub_u16 %0, %0, %1\nv_pk_max_u16 %0, %0, %1\nv_pk_sub_u16 %0, %0, %1\n"
                   "v_pk_max_u16 %0, %0, %1\nv_pk_sub_u16 %0, %0, %1\nv_pk_max_u16 %0, %0, %1\nv_pk_sub_u16 %0, %0, %1" : "+v"(a0) : "v"(b));
    if (CHAINS == 2)
      asm volatile("v_pk_max_u16 %0, %0, %2\nv_pk_max_u16 %1, %1, %2\nv_pk_sub_u16 %0, %0, %2\nv_pk_sub_u16 %1, %1, %2\n"
                   "v_pk_max_u16 %0, %0, %2\nv_pk_max_u16 %1, %1, %2\nv_pk_sub_u16 %0, %0, %2\nv_pk_sub_u16 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(b));
    if (CHAINS == 4)
      asm volatile("v_pk_max_u16 %0, %0, %4\nv_pk_max_u16 %1, %1, %4\nv_pk_max_u16 %2, %2, %4\nv_pk_max_u16 %3, %3, %4\n"
                   "v_pk_sub_u16 %0, %0, %4\nv_pk_sub_u16 %1, %1, %4\nv_pk_sub_u16 %2, %2, %4\nv_pk_sub_u16 %3, %3, %4"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    if (CHAINS == 8)
      asm volatile("v_pk_max_u16 %0, %0, %8\nv_pk_max_u16 %1, %1, %8\nv_pk_max_u16 %2, %2, %8\nv_pk_max_u16 %3, %3, %8\n"
                   "v_pk_max_u16 %4, %4, %8\nv_pk_max_u16 %5, %5, %8\nv_pk_max_u16 %6, %6, %8\nv_pk_max_u16 %7, %7, %8"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (SALU_MIX) {     // + 4 dependent scalar instructions per 8 vector ones (the row sweeps carry ~1 scalar per 3 vector instructions)
      asm volatile("s_add_i32 %0, %0, 1\ns_xor_b32 %0, %0, 5\ns_add_i32 %0, %0, 3\ns_and_b32 %0, %0, 0xffff" : "+s"(sacc));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + sacc;
}
template <int CHAINS, int SALU_MIX>
static void run(uint32_t* o, int waves_per_simd) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 1 << 15, blocks = 256 * waves_per_simd;       // one block = 4 waves = one wave per SIMD of a CU
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) { (void)hipEventRecord(e0); k<CHAINS, SALU_MIX><<<blocks, 256>>>(o, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); }
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)waves_per_simd * iters * 8;      // vector instructions one SIMD issued
  printf("chains %d, waves/SIMD %d%s: %.3f ms, %.2f cycles per vector instruction per SIMD at 2.4 GHz\n", CHAINS, waves_per_simd, SALU_MIX ? ", +4 SALU per 8 VALU" : "", ms,
         ms * 1e-3 * 2.4e9 / per_simd);
}
int main() {
  uint32_t* o; (void)hipMalloc(&o, 4 * 256 * 4096 * 4);
  for (int w : {1, 2, 3, 4}) { run<1, 0>(o, w); run<2, 0>(o, w); run<4, 0>(o, w); run<8, 0>(o, w); }
  for (int w : {1, 2, 3, 4}) { run<2, 1>(o, w); run<8, 1>(o, w); }
  return 0;
}
