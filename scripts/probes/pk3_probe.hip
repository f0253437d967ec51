// Probe for gfx950: can v_pk_maximum3_f16 / v_pk_minimum3_f16 stand in for unsigned 16-bit max / min of three?
// (1) semantics: for u16 patterns below 0x7C00 (positive finite f16, denormals included) the f16 order is the unsigned order;
//     checks every pattern against two pseudo-random partners;  (2) issue rate next to the two-input forms;
// (3) a few more rates the row sweeps depend on (DPP move, permlane swaps, v_mqsad, ds_min_u32 / ds_bpermute round trips).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/pk3_probe.hip -o gpurun_out/pk3_probe && gpurun_out/pk3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_sem(uint32_t* bad, uint32_t limit) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // i = a | b << 16 pattern pair
  const uint32_t a = i & 0xFFFFu, b = i >> 16;
  if (a >= limit || b >= limit) return;
  const uint32_t c = (a * 2654435761u + b * 40503u) % limit;
  const uint32_t pa = a | (b << 16), pb = b | (c << 16), pc = c | (a << 16);
  uint32_t mx, mn;
  asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(mx) : "v"(pa), "v"(pb), "v"(pc));
  asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(mn) : "v"(pa), "v"(pb), "v"(pc));
  const uint32_t mxl = max(a, max(b, c)), mxh = max(b, max(c, a));
  const uint32_t mnl = min(a, min(b, c)), mnh = min(b, min(c, a));
  if (mx != (mxl | (mxh << 16))) atomicAdd(bad, 1u);
  if (mn != (mnl | (mnh << 16))) atomicAdd(bad + 1, 1u);
}

#define OP8(INS)                                                                                         \
  asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n"           \
               INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n"           \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
#define OP8_3(INS)                                                                                       \
  asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" \
               INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n" \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c))
#define OP8_DPP(INS, CTRL)                                                                               \
  asm volatile(INS " %0, %0 " CTRL "\n" INS " %1, %1 " CTRL "\n" INS " %2, %2 " CTRL "\n" INS " %3, %3 " CTRL "\n" \
               INS " %4, %4 " CTRL "\n" INS " %5, %5 " CTRL "\n" INS " %6, %6 " CTRL "\n" INS " %7, %7 " CTRL "\n" \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
template <int MODE>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, int iters) {
  __shared__ uint32_t sm[4096];
  uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 + 7, a3 = a0 ^ 9, a4 = a0 + 11, a5 = a0 * 5, a6 = a0 + 13, a7 = a0 ^ 21;
  const uint32_t b = 0x00030001u + blockIdx.x, c = 0x00010002u;
  for (int k = threadIdx.x; k < 4096; k += 256) sm[k] = 0xFFFFFFFFu;
  __syncthreads();
  const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&sm[threadIdx.x];
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) OP8("v_pk_max_u16");
    if (MODE == 1) OP8_3("v_pk_maximum3_f16");
    if (MODE == 2) OP8_3("v_pk_minimum3_f16");
    if (MODE == 3) OP8_DPP("v_mov_b32_dpp", "row_shl:1 row_mask:0xf bank_mask:0xf");
    if (MODE == 4) {
      asm volatile("v_permlane16_swap_b32 %0, %1\nv_permlane16_swap_b32 %2, %3\nv_permlane16_swap_b32 %4, %5\nv_permlane16_swap_b32 %6, %7\n"
                   "v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %4, %5\nv_permlane32_swap_b32 %6, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    if (MODE == 5) {     // 4 v_mqsad (64-bit in / out) = "8 register results"
      uint64_t x0 = a0 | ((uint64_t)a1 << 32), x1 = a2 | ((uint64_t)a3 << 32), x2 = a4 | ((uint64_t)a5 << 32), x3 = a6 | ((uint64_t)a7 << 32);
      asm volatile("v_mqsad_pk_u16_u8 %0, %0, %4, %0\nv_mqsad_pk_u16_u8 %1, %1, %4, %1\nv_mqsad_pk_u16_u8 %2, %2, %4, %2\nv_mqsad_pk_u16_u8 %3, %3, %4, %3\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));
      a0 = (uint32_t)x0; a1 = (uint32_t)(x0 >> 32); a2 = (uint32_t)x1; a3 = (uint32_t)(x1 >> 32); a4 = (uint32_t)x2; a5 = (uint32_t)(x2 >> 32); a6 = (uint32_t)x3; a7 = (uint32_t)(x3 >> 32);
    }
    if (MODE == 6) {     // 8 LDS atomic minima without return, conflict-free addresses
      asm volatile("ds_min_u32 %8, %0\nds_min_u32 %8, %1 offset:1024\nds_min_u32 %8, %2 offset:2048\nds_min_u32 %8, %3 offset:3072\n"
                   "ds_min_u32 %8, %4 offset:4096\nds_min_u32 %8, %5 offset:5120\nds_min_u32 %8, %6 offset:6144\nds_min_u32 %8, %7 offset:7168\n"
                   : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(la) : "memory");
      a0 += b;
    }
    if (MODE == 7) {     // 8 dependent-free ds_bpermute + one wait
      asm volatile("ds_bpermute_b32 %0, %8, %0\nds_bpermute_b32 %1, %8, %1\nds_bpermute_b32 %2, %8, %2\nds_bpermute_b32 %3, %8, %3\n"
                   "ds_bpermute_b32 %4, %8, %4\nds_bpermute_b32 %5, %8, %5\nds_bpermute_b32 %6, %8, %6\nds_bpermute_b32 %7, %8, %7\ns_waitcnt lgkmcnt(0)\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(la) : "memory");
    }
    if (MODE == 8) OP8("v_pk_sub_u16");
    if (MODE == 9) {     // 8 ds_write_b32 + wait
      asm volatile("ds_write_b32 %8, %0\nds_write_b32 %8, %1 offset:1024\nds_write_b32 %8, %2 offset:2048\nds_write_b32 %8, %3 offset:3072\n"
                   "ds_write_b32 %8, %4 offset:4096\nds_write_b32 %8, %5 offset:5120\nds_write_b32 %8, %6 offset:6144\nds_write_b32 %8, %7 offset:7168\n"
                   : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(la) : "memory");
      a0 += b;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + sm[threadIdx.x];
}
template <int MODE>
static void run(const char* name, uint32_t* o, int per_iter = 8) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096, blocks = 256 * 8;
  for (int rep = 0; rep < 2; rep++) { hipEventRecord(e0); k_rate<MODE><<<blocks, 256>>>(o, iters); hipEventRecord(e1); hipEventSynchronize(e1); }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winst = (double)blocks * 4 * iters * per_iter;
  printf("%-22s %.3f ms  %.0f G wave-instr/s  = %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, ms, winst / ms / 1e6, 1024.0 * 2.4e9 / (winst / ms * 1e3));
}
int main() {
  uint32_t* bad; hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
  const uint32_t limit = 0x7C00u;
  k_sem<<<(1u << 32) / 256, 256>>>(bad, limit);
  uint32_t h[2]; hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
  printf("semantics on u16 patterns < 0x%x (all pairs + a third): maximum3 mismatches %u, minimum3 mismatches %u  -> %s\n", limit, h[0], h[1],
         (h[0] | h[1]) ? "NOT USABLE" : "usable as unsigned max3 / min3");
  uint32_t* o; hipMalloc(&o, 4 * 256 * 1024 * 4);
  run<0>("v_pk_max_u16", o); run<8>("v_pk_sub_u16", o); run<1>("v_pk_maximum3_f16", o); run<2>("v_pk_minimum3_f16", o);
  run<3>("v_mov_b32_dpp row_shl", o); run<4>("v_permlane16/32_swap", o); run<5>("v_mqsad_pk_u16_u8", o, 4);
  run<6>("ds_min_u32 (no rtn)", o); run<7>("ds_bpermute_b32", o); run<9>("ds_write_b32", o);
  return 0;
}
