// The MFMA phase of the windowed "shift-dy" weight-gradient kernels, shared by wgrad.hip (wgrad_sdm_kernel) and bwd_fused.hip (the fused
// dgrad + wgrad kernel): one staged activation tile [MC][TV][32 B] and one staged dy halo [HV][32 B] -> 27 taps x MC input-channel chunks.
#pragma once
#include "bpx_common.h"

namespace bpxwg {

#ifndef BPX_WGRAD_HC
#define BPX_WGRAD_HC 2
#endif
#ifndef BPX_WGRAD_REREAD
#define BPX_WGRAD_REREAD 2
#endif
// Bias gradient (column sums of dy over the tile's own voxels) on the matrix unit: wave 3 owns six taps, its seventh accumulator is
// free.  One MFMA per K-chunk of an all-ones A operand with the UN-shifted dy fragment (halo offset (1, 1, 1)) leaves sum_v dy[v][co]
// in every row of acc[6][0] (bf16 1.0 x dy is exact, the sums are fp32 like the weight gradients) - no extra registers, and it replaces
// a 16-iteration scalar-load loop over the dy tile that every wave of the workgroup ran per tile (~150 of the kernel's ~540 VALU
// instructions per tile and wave; the kernel is VALU-bound).
template <int W, int MC, int HY, int HX, int VBA, int VBG, int TV, int NKC>
__device__ __forceinline__ void sd_mfma_phase(const unsigned char* sA, const unsigned char* sG, int a_base, int g_lane, f32x4_t (&acc)[7][MC], bool want_b) {
  constexpr int T0 = 7 * W, T1 = (T0 + 7 < 27) ? T0 + 7 : 27;
  typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) {
    const int ka = kc * 32 * VBA;
    const int kg = (((kc >> 1) * HY + (kc & 1) * 2) * HX) * VBG;
    u32x4_t af[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
      const unsigned char* q = sA + c * TV * VBA + a_base + ka;
      u32x2_t l2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q)));
      u32x2_t h2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 4 * VBA)));
      af[c] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
    }
#pragma unroll
    for (int row = T0 / 3; row <= (T1 - 1) / 3; ++row) {
      const int d0 = (T0 > 3 * row ? T0 : 3 * row) - 3 * row, d1 = (T1 < 3 * row + 3 ? T1 : 3 * row + 3) - 1 - 3 * row;   // dx range of this segment
      const int smin = 2 - d1, nt = d1 - d0 + 1;
      const int dz = row / 3, dyy = row % 3;
      const unsigned char* q = sG + g_lane + kg + ((((2 - dz) * HY + (2 - dyy)) * HX + smin) * VBG);
      uint32_t w[6];
      {
        u32x2_t r0 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q)));
        u32x2_t r1 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 4 * VBG)));
        w[0] = r0[0]; w[1] = r0[1]; w[2] = r1[0]; w[3] = r1[1]; w[4] = 0u; w[5] = 0u;
        if (nt > 1 && BPX_WGRAD_REREAD != 2) {
          u32x2_t r2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 8 * VBG)));
          w[4] = r2[0]; w[5] = r2[1];
        }
      }
#pragma unroll
      for (int dx = d1; dx >= d0; --dx) {
        const int rs = (2 - dx) - smin;   // window shift of this tap in voxels: 0, 1 or 2
        u32x4_t gf;
        if (rs == 0) gf = u32x4_t{w[0], w[1], w[2], w[3]};
        else if (rs == 2) {
          if (BPX_WGRAD_REREAD) {
            u32x2_t s0 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 2 * VBG)));
            u32x2_t s1 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 6 * VBG)));
            gf = u32x4_t{s0[0], s0[1], s1[0], s1[1]};
          } else {
            gf = u32x4_t{w[1], w[2], w[3], w[4]};
          }
        }
        else if (BPX_WGRAD_REREAD == 2) {   // the one-voxel shift from LDS as well (two reads instead of a third window read + four v_alignbit)
          u32x2_t s0 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 1 * VBG)));
          u32x2_t s1 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 5 * VBG)));
          gf = u32x4_t{s0[0], s0[1], s1[0], s1[1]};
        }
        else gf = u32x4_t{__builtin_amdgcn_alignbit(w[1], w[0], 16), __builtin_amdgcn_alignbit(w[2], w[1], 16),
                          __builtin_amdgcn_alignbit(w[3], w[2], 16), __builtin_amdgcn_alignbit(w[4], w[3], 16)};
        const int a = 3 * row + dx - T0;
#pragma unroll
        for (int c = 0; c < MC; ++c)
          acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[c]), __builtin_bit_cast(bf16x8_t, gf), acc[a][c], 0, 0, 0);
      }
    }
    if (W == 3 && want_b) {
      const unsigned char* q = sG + g_lane + kg + (((1 * HY + 1) * HX + 1) * VBG);
      u32x2_t r0 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q)));
      u32x2_t r1 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 4 * VBG)));
      const u32x4_t gf = u32x4_t{r0[0], r0[1], r1[0], r1[1]};
      const u32x4_t ones = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
      acc[6][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ones), __builtin_bit_cast(bf16x8_t, gf), acc[6][0], 0, 0, 0);
    }
  }
}


}  // namespace bpxwg
