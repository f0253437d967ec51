// Probe for gfx950: what ONE wave alone on its SIMD pays per instruction, by kind — the regime of k_delaunay's top merges (one thread walks a
// seam; nothing else runs on its SIMD).  Dependent chains of 64 instructions, timed with s_memtime around 4096 rounds:
//   VALU int (v_add_u32), VALU f64 (v_fma_f64), SALU (s_add_u32 / s_mul_i32 / s_mul_hi_i32), v_readfirstlane, a scalar compare + branch,
//   an LDS round trip whose address comes from the previous one's data (ds_read_b32 -> wait -> next), the same through v_readfirstlane and
//   v_mov (what a merge whose state lives in SGPRs pays per dependent read), and ds_read_b128 against ds_read_u16.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/lone_wave_probe.hip -o /tmp/lone_wave_probe && /tmp/lone_wave_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R8(x) x x x x x x x x
#define R64(x) R8(R8(x))
template <int MODE>
__global__ void __launch_bounds__(64) k(uint64_t* out, int rounds) {
  __shared__ uint32_t sm[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = ((i * 4 + 64) & 4095) & ~15u;       // a chain of byte addresses inside sm
  __syncthreads();
  uint32_t v = threadIdx.x; double d = threadIdx.x, e = 1.0; uint32_t s = blockIdx.x + 1, s2 = 3;
  uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)sm;
  uint32_t base = a;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  u4 q = {0, 0, 0, 0};
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < rounds; i++) {
    if (MODE == 0) asm volatile(R64("v_add_u32 %0, %0, 1\n") : "+v"(v));
    if (MODE == 1) asm volatile(R64("v_fma_f64 %0, %0, %1, %1\n") : "+v"(d) : "v"(e));
    if (MODE == 2) asm volatile(R64("s_add_u32 %0, %0, 1\n") : "+s"(s) :: "scc");
    if (MODE == 3) asm volatile(R64("s_mul_i32 %0, %0, %1\n") : "+s"(s) : "s"(s2));
    if (MODE == 4) asm volatile(R64("s_mul_hi_i32 %0, %0, %1\n") : "+s"(s) : "s"(s2));
    if (MODE == 5) asm volatile(R64("v_readfirstlane_b32 %0, %1\nv_mov_b32 %1, %0\n") : "+s"(s), "+v"(v));          // 2 instructions per unit
    if (MODE == 6) asm volatile(R64("s_cmp_lg_u32 %0, 0\ns_cbranch_scc0 1f\ns_add_u32 %0, %0, 1\n1:\n") : "+s"(s) :: "scc");   // 3 per unit, branch not taken
    if (MODE == 7) asm volatile(R64("ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\nv_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(base) : "memory");   // dependent LDS reads
    if (MODE == 8) asm volatile(R64("v_mov_b32 %1, %0\nds_read_b32 %1, %1\ns_waitcnt lgkmcnt(0)\nv_readfirstlane_b32 %0, %1\ns_add_u32 %0, %0, %2\n") : "+s"(s), "+v"(v) : "s"(base) : "memory", "scc");
    if (MODE == 9) asm volatile(R64("ds_read_b128 %0, %1\ns_waitcnt lgkmcnt(0)\nv_xor_b32 %1, %1, 16\n") : "=v"(q), "+v"(a) :: "memory");   // 16-byte reads one after the other (each waited for)
    if (MODE == 10) asm volatile(R64("s_add_u32 %0, %0, 1\nv_add_u32 %1, %1, 1\n") : "+s"(s), "+v"(v) :: "scc");   // alternating SALU / VALU, independent
    if (MODE == 11) asm volatile(R64("s_add_u32 %0, %0, 1\ns_add_u32 %2, %2, 1\ns_add_u32 %0, %0, 1\nv_add_u32 %1, %1, 1\n") : "+s"(s), "+v"(v), "+s"(s2) :: "scc");   // 3 SALU per VALU
    if (MODE == 8) s = base + (s & 0xff0);
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = v + (uint64_t)d + s + a + q.x + s2; }
}
template <int MODE>
static void run(const char* name, uint64_t* o, int per_unit) {
  const int rounds = 4096;
  k<MODE><<<1, 64>>>(o, rounds); (void)hipDeviceSynchronize();
  k<MODE><<<1, 64>>>(o, rounds); (void)hipDeviceSynchronize();
  uint64_t h[2]; (void)hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
  // s_memtime counts at 100 MHz on this part: convert with the event-timed duration instead
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); k<MODE><<<1, 64>>>(o, rounds * 8); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double units = (double)rounds * 8 * 64;
  printf("%-72s %7.1f ns per unit = %6.1f cycles at 2.4 GHz (%d instruction%s per unit: %.1f cycles each)\n", name, ms * 1e6 / units, ms * 1e-3 * 2.4e9 / units, per_unit, per_unit > 1 ? "s" : "",
         ms * 1e-3 * 2.4e9 / units / per_unit);
}
int main() {
  uint64_t* o; (void)hipMalloc(&o, 1024);
  run<0>("v_add_u32, dependent", o, 1);
  run<1>("v_fma_f64, dependent", o, 1);
  run<2>("s_add_u32, dependent", o, 1);
  run<3>("s_mul_i32, dependent", o, 1);
  run<4>("s_mul_hi_i32, dependent", o, 1);
  run<5>("v_readfirstlane_b32 + v_mov_b32 (SGPR <-> VGPR round trip)", o, 2);
  run<6>("s_cmp + s_cbranch (not taken) + s_add", o, 3);
  run<7>("ds_read_b32 -> wait -> v_add: dependent LDS reads, address in a VGPR", o, 3);
  run<8>("v_mov, ds_read_b32, wait, v_readfirstlane, s_add: state in SGPRs", o, 5);
  run<9>("ds_read_b128 -> wait -> v_add", o, 3);
  run<10>("s_add_u32 + v_add_u32, independent of each other", o, 2);
  run<11>("3 s_add_u32 + 1 v_add_u32", o, 4);
  return 0;
}
