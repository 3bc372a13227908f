// What v_permlane32_swap / v_permlane16_swap and the DPP row rotations do, lane by lane (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  int a = 1000 + lane, b = 2000 + lane;
  auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[0 * 64 + lane] = r32[0]; out[1 * 64 + lane] = r32[1];
  out[2 * 64 + lane] = r16[0]; out[3 * 64 + lane] = r16[1];
  out[4 * 64 + lane] = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xC, false);   // row_ror:8, banks 2,3
  out[5 * 64 + lane] = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false);   // row_ror:8, banks 0,1
  out[6 * 64 + lane] = __builtin_amdgcn_update_dpp(a, b, 0x124, 0xf, 0xA, false);   // row_ror:4, banks 1,3
  out[7 * 64 + lane] = __builtin_amdgcn_update_dpp(b, a, 0x12C, 0xf, 0x5, false);   // row_ror:12, banks 0,2
  out[8 * 64 + lane] = __builtin_amdgcn_update_dpp(0, a, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
  out[9 * 64 + lane] = __builtin_amdgcn_update_dpp(0, a, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
}
int main() {
  int* d; hipMalloc(&d, 10 * 64 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[640]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[10] = {"swap32.a", "swap32.b", "swap16.a", "swap16.b", "ror8 old=a src=b banks23", "ror8 old=b src=a banks01",
                        "ror4 old=a src=b banks13", "ror12 old=b src=a banks02", "quad[2,3,0,1](a)", "quad[1,0,3,2](a)"};
  for (int r = 0; r < 10; ++r) { printf("%-28s", nm[r]); for (int l = 0; l < 64; l += (r < 4 ? 8 : 1)) { if (r >= 4 && l >= 20) break; printf(" %d", h[r * 64 + l]); } printf("\n"); }
  return 0;
}
