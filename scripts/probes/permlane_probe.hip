// What v_permlane16_swap / v_permlane32_swap do on gfx950, by experiment: every lane holds its lane id; prints which lane's value each lane ends with.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/permlane_probe.hip -o gpurun_out/permlane_probe && gpurun_out/permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) unsigned u2;
__global__ void k(unsigned* p) {
  const unsigned l = threadIdx.x;
  u2 a = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);          // first operand = lane id, second = 100 + lane id
  u2 b = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
  u2 eo = __builtin_amdgcn_permlane16_swap(l, l, false, false);
  u2 e = __builtin_amdgcn_permlane32_swap(eo[0], eo[0], false, false), o = __builtin_amdgcn_permlane32_swap(eo[1], eo[1], false, false);
  p[l] = a[0]; p[64 + l] = a[1]; p[128 + l] = b[0]; p[192 + l] = b[1];
  p[256 + l] = e[0]; p[320 + l] = o[0]; p[384 + l] = e[1]; p[448 + l] = o[1];
}
int main() {
  unsigned *d, h[512];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[8] = {"swap16 first ", "swap16 second", "swap32 first ", "swap32 second", "gather e[0]  ", "gather o[0]  ", "gather e[1]  ", "gather o[1]  "};
  for (int r = 0; r < 8; ++r) {
    printf("%s:", names[r]);
    for (int l = 0; l < 64; l += 8) printf(" [%2d]=%3u", l, h[r * 64 + l]);
    printf("\n");
  }
  return 0;
}
