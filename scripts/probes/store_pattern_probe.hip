// How much the LANE ORDER inside a store instruction costs on gfx950: every wave instruction writes the same contiguous 1 KB (16-byte stores) or
// 512 B (8-byte stores); only the lane -> piece assignment differs.  "mfma" = the order an MFMA accumulator gives a conv / GEMM epilogue: lane
// (g = lane / 16, j = lane % 16) holds piece g of voxel j, i.e. the 16 lanes of one pass write 16 pieces that lie 64 (32) bytes apart.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/store_pattern_probe.hip -o scripts/probes/store_pattern_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
typedef __attribute__((ext_vector_type(2))) unsigned u2;
template <int MODE> __global__ void __launch_bounds__(256) k(char* p, size_t bytes) {
  extern __shared__ char occupancy_limiter[];   // dynamic LDS only to bound the resident workgroups per CU (second table)
  const unsigned lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
  constexpr unsigned PER = (MODE < 2) ? 1024u : 512u;
  const unsigned off = MODE == 0 ? lane * 16u : MODE == 1 ? j * 64u + g * 16u : MODE == 2 ? lane * 8u : j * 32u + g * 8u;
  for (size_t c = wave; c * PER < bytes; c += nw) {
    if (MODE < 2) *reinterpret_cast<u4*>(p + c * PER + off) = u4{lane, 1u, 2u, 3u};
    else *reinterpret_cast<u2*>(p + c * PER + off) = u2{lane, 1u};
  }
}
int main() {
  const size_t bytes = 537ull << 20;
  char* d; hipMalloc(&d, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"16 B, lane-contiguous      ", "16 B, mfma order (4 x 16 B) ", "8 B, lane-contiguous       ", "8 B, mfma order (4 x 8 B)   "};
  for (int grid : {1024, 2048, 8192}) for (int m = 0; m < 4; ++m) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      if (m == 0) k<0><<<grid, 256>>>(d, bytes); else if (m == 1) k<1><<<grid, 256>>>(d, bytes); else if (m == 2) k<2><<<grid, 256>>>(d, bytes); else k<3><<<grid, 256>>>(d, bytes);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    printf("grid %5d  %s %7.1f us  %6.2f TB/s\n", grid, names[m], best * 1e3f, bytes / best / 1e9f);
  }
  // the same stores from FEW resident waves: 256-thread workgroups held to 1 / 2 / 4 per CU by their LDS size
  for (int per_cu : {1, 2, 4}) for (int m = 0; m < 2; ++m) {
    const int lds = per_cu == 1 ? 96 * 1024 : per_cu == 2 ? 64 * 1024 : 36 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(m == 0 ? k<0> : k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      if (m == 0) k<0><<<256 * per_cu, 256, lds>>>(d, bytes); else k<1><<<256 * per_cu, 256, lds>>>(d, bytes);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    printf("%d workgroup(s) of 4 waves per CU  %s %7.1f us  %6.2f TB/s\n", per_cu, names[m], best * 1e3f, bytes / best / 1e9f);
  }
  return 0;
}
