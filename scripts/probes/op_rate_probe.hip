// Probe for gfx950: issue cost of single vector instructions with the SIMD full (8 waves per SIMD, 8 independent instructions per iteration).
// Which of the candidates for k_dense2's mask / key arithmetic run at full rate (4 cycles per wave instruction)?
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/op_rate_probe.hip -o /tmp/op_probe && /tmp/op_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define KERNEL(NAME, BODY)                                                                                  \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters) {                                   \
    uint32_t a[8]; uint64_t q[8];                                                                           \
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * (i + 3); q[i] = (uint64_t)a[i] << 7; }               \
    uint32_t b = 5 + (threadIdx.x & 7), c = 0x01020304u + blockIdx.x;                                       \
    for (int it = 0; it < iters; it++) { BODY }                                                             \
    uint32_t s = 0; for (int i = 0; i < 8; i++) s += a[i] + (uint32_t)q[i] + (uint32_t)(q[i] >> 32);        \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                                \
  }
#define S_LSHL32(i) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define S_LSHL64(i) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q[i]) : "v"(b));
#define S_SADHI(i) asm volatile("v_sad_hi_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_SAD(i) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define S_MED3(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_BFE(i) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(a[i]));
#define S_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_BFM(i) asm volatile("v_bfm_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define S_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
#define S_FFBL(i) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]));
#define S_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(a[i]) : "v"(b));
#define S_ADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
#define S_BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x30" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_READLANE(i) { uint32_t t; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(t) : "v"(a[i])); asm volatile("" :: "s"(t)); }
#define S_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define S_CMP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
#define S_MIN3(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_MQSAD(i) asm volatile("v_mqsad_pk_u16_u8 %0, %0, %1, %0" : "+v"(q[i]) : "v"(b));
#define S_QSAD(i) asm volatile("v_qsad_pk_u16_u8 %0, %0, %1, %0" : "+v"(q[i]) : "v"(b));
#define S_PKADD(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_PKMAX3(i) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_CVTPK(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(c));
KERNEL(k_lshl32, REP8(S_LSHL32)) KERNEL(k_lshl64, REP8(S_LSHL64)) KERNEL(k_sadhi, REP8(S_SADHI)) KERNEL(k_sad, REP8(S_SAD))
KERNEL(k_mullo, REP8(S_MULLO)) KERNEL(k_mulhi, REP8(S_MULHI)) KERNEL(k_med3, REP8(S_MED3)) KERNEL(k_bfe, REP8(S_BFE))
KERNEL(k_andor, REP8(S_ANDOR)) KERNEL(k_perm, REP8(S_PERM)) KERNEL(k_bfm, REP8(S_BFM)) KERNEL(k_align, REP8(S_ALIGN))
KERNEL(k_ffbl, REP8(S_FFBL)) KERNEL(k_lshladd, REP8(S_LSHLADD)) KERNEL(k_add64, REP8(S_ADD64)) KERNEL(k_bitop3, REP8(S_BITOP3))
KERNEL(k_mqsad, REP8(S_MQSAD)) KERNEL(k_qsad, REP8(S_QSAD)) KERNEL(k_pkadd, REP8(S_PKADD)) KERNEL(k_pkmax3, REP8(S_PKMAX3))
KERNEL(k_readlane, REP8(S_READLANE)) KERNEL(k_cndmask, REP8(S_CNDMASK)) KERNEL(k_cmp, REP8(S_CMP)) KERNEL(k_min3, REP8(S_MIN3))
template <typename K> static void run(const char* name, K kern, uint32_t* o) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 1 << 13, wps = 8, blocks = 256 * wps;
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) { (void)hipEventRecord(e0); kern<<<blocks, 256>>>(o, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); }
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-14s %.2f cycles per wave instruction per SIMD at 2.4 GHz\n", name, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 8));
}
int main() {
  uint32_t* o; (void)hipMalloc(&o, 4 * 256 * 4096 * 4);
#define R(k) run(#k, k, o);
  R(k_lshl32) R(k_lshl64) R(k_sadhi) R(k_sad) R(k_mullo) R(k_mulhi) R(k_med3) R(k_bfe) R(k_andor) R(k_perm) R(k_bfm) R(k_align) R(k_ffbl) R(k_lshladd)
  R(k_add64) R(k_bitop3) R(k_mqsad) R(k_qsad) R(k_pkadd) R(k_pkmax3) R(k_readlane) R(k_cndmask) R(k_cmp) R(k_min3)
  return 0;
}
