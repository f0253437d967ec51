// Probe for gfx950: issue rate of the VALU forms the SGM recurrence could be built from.  8 independent chains per lane,
// 16 waves per CU, inline asm so that the compiler neither folds nor re-associates.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/valu_rate_probe.hip -o gpurun_out/valu_rate_probe && gpurun_out/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OP8(INS)                                                                                         \
  asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n"           \
               INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n"           \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
#define OP8_3(INS)                                                                                       \
  asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" \
               INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n" \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c))
template <int MODE>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, int iters) {
  uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 + 7, a3 = a0 ^ 9, a4 = a0 + 11, a5 = a0 * 5, a6 = a0 + 13, a7 = a0 ^ 21;
  const uint32_t b = 0x00030001u + blockIdx.x, c = 0x00010002u;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) OP8("v_pk_min_u16");
    if (MODE == 1) OP8("v_pk_add_u16");
    if (MODE == 2) OP8("v_pk_min_f16");
    if (MODE == 3) OP8("v_pk_add_f16");
    if (MODE == 4) OP8("v_min_u32");
    if (MODE == 5) OP8("v_min_f32");
    if (MODE == 6) OP8("v_add_f32");
    if (MODE == 7) OP8_3("v_perm_b32");
    if (MODE == 8) OP8_3("v_alignbit_b32");
    if (MODE == 9) OP8_3("v_min3_u32");
    if (MODE == 10) OP8_3("v_pk_mad_u16");
    if (MODE == 11) OP8_3("v_min3_f32");
    if (MODE == 12) OP8("v_pk_max_f16");
    if (MODE == 13) OP8_3("v_pk_fma_f16");
    if (MODE == 14) OP8("v_pk_sub_u16");
    if (MODE == 15) OP8_3("v_min3_u16");
    if (MODE == 16) OP8_3("v_min3_f16");
    if (MODE == 17) OP8_3("v_add3_u32");
    if (MODE == 18) OP8("v_add_u32");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE>
static void run(const char* name, uint32_t* o) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096, blocks = 256 * 8;
  for (int rep = 0; rep < 2; rep++) { hipEventRecord(e0); k_rate<MODE><<<blocks, 256>>>(o, iters); hipEventRecord(e1); hipEventSynchronize(e1); }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winst = (double)blocks * 4 * iters * 8;
  printf("%-16s %.3f ms  %.0f G wave-instr/s  = %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, ms, winst / ms / 1e6, 1024.0 * 2.4e9 / (winst / ms * 1e3));
}
int main() {
  uint32_t* o; hipMalloc(&o, 4 * 256 * 1024 * 4);
  run<0>("v_pk_min_u16", o); run<1>("v_pk_add_u16", o); run<14>("v_pk_sub_u16", o); run<10>("v_pk_mad_u16", o);
  run<2>("v_pk_min_f16", o); run<12>("v_pk_max_f16", o); run<3>("v_pk_add_f16", o); run<13>("v_pk_fma_f16", o);
  run<4>("v_min_u32", o); run<18>("v_add_u32", o); run<9>("v_min3_u32", o); run<17>("v_add3_u32", o); run<15>("v_min3_u16", o);
  run<5>("v_min_f32", o); run<6>("v_add_f32", o); run<11>("v_min3_f32", o); run<16>("v_min3_f16", o);
  run<7>("v_perm_b32", o); run<8>("v_alignbit_b32", o);
  return 0;
}
