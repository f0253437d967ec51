#include <hip/hip_runtime.h>
__global__ void k(int* o) {
  int x = threadIdx.x * 3 + 1;
  int y = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false);   // wave_shl:1
  int z = __builtin_amdgcn_update_dpp(-1, x, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
  o[threadIdx.x] = y; o[64 + threadIdx.x] = z;
}
int main() {
  int* d; hipMalloc(&d, 128 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); int h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i++) printf("%d:%d,%d ", i, h[i], h[64 + i]); printf("\n"); return 0;
}
