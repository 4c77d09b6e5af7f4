// Voxel-wise GEMMs on the matrix cores: Conv3d k=1, ConvTranspose3d k=s=2 (forward and dgrad).
//
// All three are D[col][v] = sum_k Wp[col][k] * A[k][v] with no spatial reuse, so the activation
// operand goes straight from global memory to the MFMA "B" registers (each element is read exactly
// once from HBM; LDS would be a pure round trip) and the weights come from the dense packed layout
// [kgroup q][cols][KPL] through L1/L2.  What differs is only how a voxel index maps to addresses:
//   PW_CONV1  : A = x[v], store y[v]                       (+ bias, + IN-backward affine, + addend)
//   PW_CONVT  : A = x[v], column block -> (sub-position, co), scatter to y[2v+sub]   (+ bias, stats)
//   PW_CONVTD : A gathered from dy[2v+sub] over the 8 sub-positions (K = 8*Cout), store dx[v]
#include <type_traits>

#include <cstdlib>

#include "bpx_common.h"

// transposed-conv forward, K1STEP instance: voxel blocks whose operands are in flight ahead of the one being computed (pw_kernel)
#ifndef BPX_PW_MS_CT
#define BPX_PW_MS_CT 2
#endif
#ifndef BPX_CONVT_OCC
#define BPX_CONVT_OCC 2                    // waves per SIMD the convt_k1_kernel is compiled for
#endif
#ifndef BPX_CONVT_PD
#define BPX_CONVT_PD 2
#endif

namespace {

enum { PW_CONV1 = 0, PW_CONVT = 1, PW_CONVTD = 2 };

// four consecutive channels of one voxel with a single 8-byte (bf16) / 16-byte (f32) load
template <typename T> __device__ __forceinline__ void load4(const T* p, float* f);
template <> __device__ __forceinline__ void load4<uint16_t>(const uint16_t* p, float* f) {
  u32x2_t v = *reinterpret_cast<const u32x2_t*>(p);
  f[0] = bf16lo(v[0]); f[1] = bf16hi(v[0]); f[2] = bf16lo(v[1]); f[3] = bf16hi(v[1]);
}
template <> __device__ __forceinline__ void load4<float>(const float* p, float* f) {
  f32x4_t v = *reinterpret_cast<const f32x4_t*>(p);
  f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
}

// 4*NQ consecutive channels of one voxel: 16-byte accesses for pairs of quads, one 8-byte access for an odd last quad
// (fp32: one 16-byte access per quad).  The addresses are 8*NQ-byte (bf16) aligned, which global memory accepts.
template <typename T, int NQ> __device__ __forceinline__ void loadq(const T* p, float (*f)[4]);
template <typename T, int NQ> __device__ __forceinline__ void storeq(T* p, const float (*f)[4]);
template <> __device__ __forceinline__ void loadq<float, 1>(const float* p, float (*f)[4]) { load4<float>(p, f[0]); }
template <> __device__ __forceinline__ void loadq<float, 2>(const float* p, float (*f)[4]) { load4<float>(p, f[0]); load4<float>(p + 4, f[1]); }
template <> __device__ __forceinline__ void loadq<float, 3>(const float* p, float (*f)[4]) { load4<float>(p, f[0]); load4<float>(p + 4, f[1]); load4<float>(p + 8, f[2]); }
template <> __device__ __forceinline__ void loadq<float, 4>(const float* p, float (*f)[4]) { load4<float>(p, f[0]); load4<float>(p + 4, f[1]); load4<float>(p + 8, f[2]); load4<float>(p + 12, f[3]); }
__device__ __forceinline__ void load8_bf16(const uint16_t* p, float* f0, float* f1) {
  u32x4_t v = *reinterpret_cast<const u32x4_t*>(p);
  f0[0] = bf16lo(v[0]); f0[1] = bf16hi(v[0]); f0[2] = bf16lo(v[1]); f0[3] = bf16hi(v[1]);
  f1[0] = bf16lo(v[2]); f1[1] = bf16hi(v[2]); f1[2] = bf16lo(v[3]); f1[3] = bf16hi(v[3]);
}
template <> __device__ __forceinline__ void loadq<uint16_t, 1>(const uint16_t* p, float (*f)[4]) { load4<uint16_t>(p, f[0]); }
template <> __device__ __forceinline__ void loadq<uint16_t, 2>(const uint16_t* p, float (*f)[4]) { load8_bf16(p, f[0], f[1]); }
template <> __device__ __forceinline__ void loadq<uint16_t, 3>(const uint16_t* p, float (*f)[4]) { load8_bf16(p, f[0], f[1]); load4<uint16_t>(p + 8, f[2]); }
template <> __device__ __forceinline__ void loadq<uint16_t, 4>(const uint16_t* p, float (*f)[4]) { load8_bf16(p, f[0], f[1]); load8_bf16(p + 8, f[2], f[3]); }
template <int NQ> __device__ __forceinline__ void storeq_f32(float* p, const float (*f)[4]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4_t*>(p + 4 * q) = f32x4_t{f[q][0], f[q][1], f[q][2], f[q][3]};
}
template <> __device__ __forceinline__ void storeq<float, 1>(float* p, const float (*f)[4]) { storeq_f32<1>(p, f); }
template <> __device__ __forceinline__ void storeq<float, 2>(float* p, const float (*f)[4]) { storeq_f32<2>(p, f); }
template <> __device__ __forceinline__ void storeq<float, 3>(float* p, const float (*f)[4]) { storeq_f32<3>(p, f); }
template <> __device__ __forceinline__ void storeq<float, 4>(float* p, const float (*f)[4]) { storeq_f32<4>(p, f); }
__device__ __forceinline__ void store4_bf16(uint16_t* p, const float* f) {
  *reinterpret_cast<u32x2_t*>(p) = u32x2_t{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3])};
}
__device__ __forceinline__ void store8_bf16(uint16_t* p, const float* f0, const float* f1) {
  *reinterpret_cast<u32x4_t*>(p) = u32x4_t{pack_bf16x2(f0[0], f0[1]), pack_bf16x2(f0[2], f0[3]), pack_bf16x2(f1[0], f1[1]), pack_bf16x2(f1[2], f1[3])};
}
template <> __device__ __forceinline__ void storeq<uint16_t, 1>(uint16_t* p, const float (*f)[4]) { store4_bf16(p, f[0]); }
template <> __device__ __forceinline__ void storeq<uint16_t, 2>(uint16_t* p, const float (*f)[4]) { store8_bf16(p, f[0], f[1]); }
template <> __device__ __forceinline__ void storeq<uint16_t, 3>(uint16_t* p, const float (*f)[4]) { store8_bf16(p, f[0], f[1]); store4_bf16(p + 8, f[2]); }
template <> __device__ __forceinline__ void storeq<uint16_t, 4>(uint16_t* p, const float (*f)[4]) { store8_bf16(p, f[0], f[1]); store8_bf16(p + 8, f[2], f[3]); }

// fp16 storage (forward kernels only): same access widths as bf16
template <> __device__ __forceinline__ void load4<f16_t>(const f16_t* p, float* f) {
  u32x2_t v = *reinterpret_cast<const u32x2_t*>(p);
  f[0] = lo16<f16_t>(v[0]); f[1] = hi16<f16_t>(v[0]); f[2] = lo16<f16_t>(v[1]); f[3] = hi16<f16_t>(v[1]);
}
__device__ __forceinline__ void load8_f16(const f16_t* p, float* f0, float* f1) {
  u32x4_t v = *reinterpret_cast<const u32x4_t*>(p);
  f0[0] = lo16<f16_t>(v[0]); f0[1] = hi16<f16_t>(v[0]); f0[2] = lo16<f16_t>(v[1]); f0[3] = hi16<f16_t>(v[1]);
  f1[0] = lo16<f16_t>(v[2]); f1[1] = hi16<f16_t>(v[2]); f1[2] = lo16<f16_t>(v[3]); f1[3] = hi16<f16_t>(v[3]);
}
template <> __device__ __forceinline__ void loadq<f16_t, 1>(const f16_t* p, float (*f)[4]) { load4<f16_t>(p, f[0]); }
template <> __device__ __forceinline__ void loadq<f16_t, 2>(const f16_t* p, float (*f)[4]) { load8_f16(p, f[0], f[1]); }
template <> __device__ __forceinline__ void loadq<f16_t, 3>(const f16_t* p, float (*f)[4]) { load8_f16(p, f[0], f[1]); load4<f16_t>(p + 8, f[2]); }
template <> __device__ __forceinline__ void loadq<f16_t, 4>(const f16_t* p, float (*f)[4]) { load8_f16(p, f[0], f[1]); load8_f16(p + 8, f[2], f[3]); }
__device__ __forceinline__ void store4_f16(f16_t* p, const float* f) {
  *reinterpret_cast<u32x2_t*>(p) = u32x2_t{pk16s<f16_t>(f[0], f[1]), pk16s<f16_t>(f[2], f[3])};   // saturating: raw outputs (bpx_common.h)
}
__device__ __forceinline__ void store8_f16(f16_t* p, const float* f0, const float* f1) {
  *reinterpret_cast<u32x4_t*>(p) = u32x4_t{pk16s<f16_t>(f0[0], f0[1]), pk16s<f16_t>(f0[2], f0[3]), pk16s<f16_t>(f1[0], f1[1]), pk16s<f16_t>(f1[2], f1[3])};
}
template <> __device__ __forceinline__ void storeq<f16_t, 1>(f16_t* p, const float (*f)[4]) { store4_f16(p, f[0]); }
template <> __device__ __forceinline__ void storeq<f16_t, 2>(f16_t* p, const float (*f)[4]) { store8_f16(p, f[0], f[1]); }
template <> __device__ __forceinline__ void storeq<f16_t, 3>(f16_t* p, const float (*f)[4]) { store8_f16(p, f[0], f[1]); store4_f16(p + 8, f[2]); }
template <> __device__ __forceinline__ void storeq<f16_t, 4>(f16_t* p, const float (*f)[4]) { store8_f16(p, f[0], f[1]); store8_f16(p + 8, f[2], f[3]); }

// The same 4*NQ consecutive channels c0.. of a voxel of a CHUNK-PLANAR tensor (channel c of a voxel at (c/16)*cs + c%16 from the voxel's
// base): the widest accesses that stay inside one 16-channel chunk.  c0 is a multiple of 4*NQ for NQ = 1, 2, 4 (never straddles); for
// NQ = 3 the 12 channels split as 8 + 4 or 4 + 8 depending on the lane.
template <typename T, int NQ> __device__ __forceinline__ void loadq_planar(const T* vbase, int c0, int cs, float (*f)[4]) {
  auto at = [&](int c) { return vbase + (size_t)(c >> 4) * cs + (c & 15); };
  if (NQ == 1) loadq<T, 1>(at(c0), f);
  else if (NQ == 2) loadq<T, 2>(at(c0), f);
  else if (NQ == 4) { loadq<T, 2>(at(c0), f); loadq<T, 2>(at(c0 + 8), f + 2); }
  else if (sizeof(T) == 4) { loadq<T, 1>(at(c0), f); loadq<T, 1>(at(c0 + 4), f + 1); loadq<T, 1>(at(c0 + 8), f + 2); }   // fp32 quads are 16 bytes
  else {
    // bf16, 12 channels from c0 = 12 g: every lane reads the TWO aligned 8-channel blocks that hold them (same two instructions for the
    // whole wave, no divergence) and keeps channels [0, 12) of the 16 (c0 % 8 == 0) or [4, 16) (c0 % 8 == 4)
    const int b0 = c0 & ~7;
    float a[2][4], b[2][4];
    loadq<T, 2>(at(b0), a);
    loadq<T, 2>(at(b0 + 8), b);
    const bool odd = (c0 & 7) != 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f[0][r] = odd ? a[1][r] : a[0][r];
      f[1][r] = odd ? b[0][r] : a[1][r];
      f[2][r] = odd ? b[1][r] : b[0][r];
    }
  }
}
template <typename T, int NQ> __device__ __forceinline__ void storeq_planar(T* vbase, int c0, int cs, const float (*f)[4]) {
  auto at = [&](int c) { return vbase + (size_t)(c >> 4) * cs + (c & 15); };
  if (NQ == 1) storeq<T, 1>(at(c0), f);
  else if (NQ == 2) storeq<T, 2>(at(c0), f);
  else if (NQ == 4) { storeq<T, 2>(at(c0), f); storeq<T, 2>(at(c0 + 8), f + 2); }
  else if ((c0 & 7) == 0) { storeq<T, 2>(at(c0), f); storeq<T, 1>(at(c0 + 8), f + 2); }
  else { storeq<T, 1>(at(c0), f); storeq<T, 2>(at(c0 + 4), f + 1); }
}

struct PwParams {
  int N, D, H, W;        // voxel grid of v (the low-res grid for the transposed conv)
  int sz;                // transposed conv: kernel = stride = (sz,2,2), sz = 1 or 2 -> 4*sz sub-positions, sub = (a*2+b)*2+c
  int64_t vps;           // voxels per sample = D*H*W
  const void* x; int x_ld; int K;          // K = reduction length in channels (8*Cout for CONVTD)
  int Csub;              // CONVT: Cout (columns per sub-position); CONVTD: channels per sub-position of dy
  const void* wp; const float* bias;
  void* y; int y_ld; int Ncols;            // Ncols = total columns (8*Cout for CONVT)
  void* y2; int y2_ld; int ysplit;         // CONV1, optional: columns >= ysplit go to y2 (column - ysplit); ysplit % 4 == 0
  // CONV1 extras
  const void* g; int g_ld; const void* t; int t_ld; const bpx_nbwd_coef* coef;
  int t_cs, y_cs;                          // elements between 16-channel chunks of t / y: 16, or the plane of a chunk-planar tensor
  const void* addend; int addend_ld;
  float* part; int mblocks;                // voxel blocks (64 * MS voxels each) per sample
  uint32_t y_bytes;                        // convt_k1_kernel: extent of y in bytes from p.y (buffer range)
  int mgroups;                             // CONVT: persistent workgroups per (sample, column block), each walking blocks grp, grp + mgroups, ..;
                                           // stats: [N][mgroups * 4 sz][2][Csub]; the other modes: mgroups = mblocks (one block per workgroup)
};

// PL: chunk-planar t (PW_CONV1) / y (PW_CONVT) operand.  A compile-time switch on purpose: with a run-time test the interleaved instance
// of the 48-column GEMM went from 116 to 128 VGPRs and from 0.62 to 0.75 ms (cfg 2, level 0).
// TT: element type of the t operand of PW_CONV1's fused InstanceNorm-backward affine (BPX_MIX16: the forward pass's fp16 activations beside
// bf16 gradients), else T
// K1STEP (transposed-conv forward into a chunk-planar buffer with Cin = 32, i.e. ONE K step: level 0 of cfg 2): weights kept in registers and the next
// block's operand requested before this block's stores.  An instance of its own: the 40 extra VGPRs cost the deeper levels a workgroup per CU
// (58 -> 73 us at 64 -> 64 channels) when every transposed-conv forward carried them.
template <typename T, int MS, int NS, int MODE, bool PL, typename TT = T, bool K1STEP = false>
__global__ void __launch_bounds__(256) pw_kernel(const PwParams p) {
  using Tr = ElemTraits<T>;
  constexpr int KPL = Tr::KPL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int nbk = p.Ncols / (16 * NS);
  // (Round 5, measured and dropped: an XCD swizzle - workgroup i as logical id (i % 8) * (grid / 8) + i / 8, so that the 4 column blocks of a voxel
  // group, which read the same input voxels, share one L2 instead of four.  Level 0 of cfg 2: 202 -> 225 us; levels 1-3: 56 -> 52, 28 -> 27, 17 -> 16 us.
  // The input is 1/8 of the traffic and comes from the MALL either way; the writes of an XCD concentrated on 1/8 of the output cost more.)
  const unsigned bid = blockIdx.x;
  const int nb = bid % nbk;
  const int grp = (bid / nbk) % p.mgroups;
  const int n = bid / (nbk * p.mgroups);
  const int col_base = nb * 16 * NS;
  // PERM (transposed conv into a chunk-planar buffer, 64 columns per block, Cout % 32 == 0): the block is the x pair of sub-positions
  // (2 sp, 2 sp + 1) of the 32 channels [32 pp, 32 pp + 32), and lane row g owns 8 channels of EACH of the two 16-channel planes at
  // sub-position 2 sp + (g >> 1): channels 32 pp + (ns >> 1) * 16 + (g & 1) * 8 + (ns & 1) * 4 + r.  One store instruction then writes, per
  // input voxel, 64 contiguous bytes of one plane (lanes g = 0..3: two voxels x 32 bytes) and 1 KB per 16-voxel lane row - full 64-byte
  // requests - where the plain binding (lane row g = 16 consecutive channels) leaves every request half empty until the second store.
  constexpr bool PERM = MODE == PW_CONVT && PL && NS == 4;
  const int ppb = PERM ? p.Csub / 32 : 1;
  const int sp = nb / ppb, pp = nb - sp * ppb;

  // statistics of the transposed conv accumulate over every voxel block this workgroup walks (one reduction and one partial row per
  // workgroup: with a row per 128-voxel block the 16-lane reductions, the LDS exchange and the barrier cost more than the block's 8 MFMAs)
  float s1[NS][4], s2[NS][4];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[ns][r] = s2[ns][r] = 0.f;
  T* __restrict__ yout = reinterpret_cast<T*>(p.y);
  const T* __restrict__ xin = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wp = reinterpret_cast<const T*>(p.wp);

  int mb = grp;
  // Transposed-conv forward (the persistent loop), round 4: a wave's VMEM operations retire in order, so the wait for block i + 1's operand - and for
  // its weight fragments, re-loaded per block - also waited for the write acknowledgement of block i's stores: every iteration was
  // [load latency + store latency], 2.3 TB/s written.  With one K step (Cin = 32, the level-0 / cfg-2 case) the weights now stay in registers and the
  // operand of the NEXT block is requested BEFORE this block's stores: its wait leaves the stores outstanding.
  // Round 5: the operands of the next PD blocks, not of one.  VMEM retires in order, so the wait for block i + 1's operand also waits for every store
  // issued before its request: with one block ahead a wave had the stores of ONE block (4 KB) in flight, a CU of two 176-VGPR workgroups 32 KB, the
  // chip 8 MB - at ~3 us of write latency 2.7 TB/s, which is what the launch measured (537 MB in 204 us) while a plain store loop reaches 4.8-5.2 TB/s
  // in any lane order (scripts/probes/store_pattern_probe.hip).  The operand is 8 registers per block.
  constexpr bool PRE = MODE == PW_CONVT && K1STEP;
  constexpr bool pre = PRE;                        // (the launcher picks the instance for K <= 32)
  constexpr int PD = PRE ? BPX_CONVT_PD : 1;
  u32x4_t af_pre[PD][PRE ? MS : 1], wf_keep[PRE ? NS : 1];
  auto prefetch = [&](int mbn, auto slot_c) {
    constexpr int SL = decltype(slot_c)::value;
#pragma unroll
    for (int ms = 0; ms < (PRE ? MS : 1); ++ms) {
      const uint32_t vn = ((uint32_t)mbn * 4u + (uint32_t)wave) * (uint32_t)(MS * 16) + (uint32_t)(ms * 16 + j);
      af_pre[SL][ms] = u32x4_t{0u, 0u, 0u, 0u};
      if (g * KPL < p.K && vn < (uint32_t)p.vps) af_pre[SL][ms] = *reinterpret_cast<const u32x4_t*>(xin + ((size_t)n * p.vps + vn) * (size_t)p.x_ld + g * KPL);
    }
  };
  if (pre) {
    prefetch(mb, std::integral_constant<int, 0>{});
    if (PD > 1 && mb + p.mgroups < p.mblocks) prefetch(mb + p.mgroups, std::integral_constant<int, (PD > 1 ? 1 : 0)>{});
    if (PD > 2 && mb + 2 * p.mgroups < p.mblocks) prefetch(mb + 2 * p.mgroups, std::integral_constant<int, (PD > 2 ? 2 : 0)>{});
    if (PD > 3 && mb + 3 * p.mgroups < p.mblocks) prefetch(mb + 3 * p.mgroups, std::integral_constant<int, (PD > 3 ? 3 : 0)>{});
#pragma unroll
    for (int ns = 0; ns < (PRE ? NS : 1); ++ns) {
      const int gq = j >> 2;
      const int colw = PERM ? (2 * sp + (gq >> 1)) * p.Csub + pp * 32 + (ns >> 1) * 16 + (gq & 1) * 8 + (ns & 1) * 4 + (j & 3)
                            : col_base + gq * (4 * NS) + ns * 4 + (j & 3);
      wf_keep[ns] = *reinterpret_cast<const u32x4_t*>(wp + ((size_t)g * p.Ncols + colw) * KPL);
    }
  }
  // ---- column binding of the epilogue and bias -------------------------------------------------------------------------------------
  // MFMA row i of column block ns is bound to column col_base + (i/4)*4NS + ns*4 + i%4 (see the weight load above), so
  // lane (g, j) ends up with the 4*NS CONSECUTIVE columns col_base + g*4NS .. of voxel j: its g / t / addend operands and
  // its results move as 16-byte (two column quads) + 8-byte accesses, and the 4 lanes of a voxel cover 16*NS contiguous
  // channels.
  // Transposed-conv forward (round 5): set up ONCE, in front of the block loop.  Inside the loop the bias loads were re-issued per block - the
  // output stores may alias them as far as the compiler knows - and sat in the VMEM queue BEHIND the prefetch of the next block's operand: the
  // epilogue's wait for its 16 bias values was a wait for that prefetch, a full memory latency per block, whatever the depth of the operand
  // ring.  (The single-pass modes keep the setup behind their MFMA loop: in front of it the values cost the 1x1x1 GEMMs up to a wave per SIMD.)
  const int col0 = col_base + g * 4 * NS;        // first of this lane's columns
  int sub = 0, co0 = col0;
  if (MODE == PW_CONVT) { sub = col0 / p.Csub; co0 = col0 % p.Csub; }
  if (PERM) { sub = 2 * sp + (g >> 1); co0 = pp * 32 + (g & 1) * 8; }
  auto cof = [&](int ns) { return PERM ? co0 + (ns >> 1) * 16 + (ns & 1) * 4 : co0 + ns * 4; };   // first channel of column quad ns
  float add[NS][4];
  auto load_bias = [&]() {
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 4; ++r) add[ns][r] = p.bias ? p.bias[cof(ns) + r] : 0.f;
  };
  if (MODE == PW_CONVT) load_bias();
  // one voxel block; slot_c = the operand ring slot it consumes (and refills for block mb + PD mgroups)
  auto one_block = [&](auto slot_c) {
  constexpr int SL = decltype(slot_c)::value;
  // voxel of lane (j) for each m-subtile
  // voxels per sample < 2^31 (checked on the host): 32-bit index math - the 64-bit div/mod sequences of the transposed-conv
  // coordinates were ~300 of the kernel's ~490 VALU instructions per wave, which bound it
  uint32_t v[MS];
  bool valid[MS];
  size_t abase[MS];  // element offset of the voxel's channel 0 in x (CONV1/CONVT) or of sub-position 0 in dy (CONVTD)
  uint32_t cx[MS], cy[MS], cz[MS];   // (x, y, z) of the voxel on the p.W x p.H x p.D grid (CONVT / CONVTD)
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    v[ms] = ((uint32_t)mb * 4u + (uint32_t)wave) * (uint32_t)(MS * 16) + (uint32_t)(ms * 16 + j);
    valid[ms] = v[ms] < (uint32_t)p.vps;
    const uint32_t vv = valid[ms] ? v[ms] : 0u;
    if (MODE != PW_CONV1) {
      const uint32_t row = vv / (uint32_t)p.W;
      cx[ms] = vv - row * (uint32_t)p.W;
      cz[ms] = row / (uint32_t)p.H;
      cy[ms] = row - cz[ms] * (uint32_t)p.H;
    }
    if (MODE == PW_CONVTD) {
      abase[ms] = ((((size_t)n * p.sz * p.D + p.sz * cz[ms]) * 2 * p.H + 2 * cy[ms]) * 2 * p.W + 2 * cx[ms]) * (size_t)p.x_ld;
    } else {
      abase[ms] = ((size_t)n * p.vps + vv) * (size_t)p.x_ld;
    }
  }

  f32x4_t acc[MS][NS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int ksteps = (p.K / KPL + 3) / 4;
  for (int s = 0; s < ksteps; ++s) {
    const int k = (4 * s + g) * KPL;  // first reduction channel of this lane's operand
    const bool kin = k < p.K;
    u32x4_t wf[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
    {
      if (PRE && pre) { wf[ns] = wf_keep[PRE ? ns : 0]; continue; }
      const int gq = j >> 2;
      const int colw = PERM ? (2 * sp + (gq >> 1)) * p.Csub + pp * 32 + (ns >> 1) * 16 + (gq & 1) * 8 + (ns & 1) * 4 + (j & 3)
                            : col_base + gq * (4 * NS) + ns * 4 + (j & 3);
      wf[ns] = *reinterpret_cast<const u32x4_t*>(wp + ((size_t)(4 * s + g) * p.Ncols + colw) * KPL);
    }
    size_t koff;
    if (MODE == PW_CONVTD) {
      int sub = kin ? k / p.Csub : 0, c = kin ? k % p.Csub : 0;
      int a = (sub >> 2) & 1, b = (sub >> 1) & 1, cc = sub & 1;
      koff = (((size_t)a * 2 * p.H + b) * 2 * p.W + cc) * (size_t)p.x_ld + c;
    } else {
      koff = kin ? k : 0;
    }
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
      u32x4_t af = u32x4_t{0u, 0u, 0u, 0u};
      if (PRE && pre) af = af_pre[SL][PRE ? ms : 0];
      else if (kin && valid[ms]) af = *reinterpret_cast<const u32x4_t*>(xin + abase[ms] + koff);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = mfma_step<T>(wf[ns], af, acc[ms][ns]);
    }
  }
  if (PRE && pre && mb + PD * p.mgroups < p.mblocks) prefetch(mb + PD * p.mgroups, slot_c);   // before the stores below

  // ---- epilogue -------------------------------------------------------------------------------------------------------------------
  if (MODE != PW_CONVT) load_bias();
  bpx_nbwd_coef cf[NS][4];
  if (MODE == PW_CONV1 && p.coef) {
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 4; ++r) cf[ns][r] = p.coef[(size_t)n * p.Ncols + co0 + ns * 4 + r];
  }
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    if (!valid[ms]) continue;
    size_t ovox;
    if (MODE == PW_CONVT) {
      const int xw = (int)cx[ms], yh = (int)cy[ms], zd = (int)cz[ms];
      int a = (sub >> 2) & 1, b = (sub >> 1) & 1, cc = sub & 1;
      ovox = (((size_t)n * p.sz * p.D + p.sz * zd + a) * 2 * p.H + 2 * yh + b) * 2 * p.W + 2 * xw + cc;
    } else {
      ovox = (size_t)n * p.vps + v[ms];
    }
    float val[NS][4];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 4; ++r) val[ns][r] = acc[ms][ns][r] + add[ns][r];
    if (MODE == PW_CONV1) {
      if (p.coef) {
        float gq[NS][4], tq[NS][4];
        loadq<T, NS>(reinterpret_cast<const T*>(p.g) + ovox * (size_t)p.g_ld + co0, gq);
        if (!PL) loadq<TT, NS>(reinterpret_cast<const TT*>(p.t) + ovox * (size_t)p.t_ld + co0, tq);
        else loadq_planar<TT, NS>(reinterpret_cast<const TT*>(p.t) + ovox * (size_t)p.t_ld, co0, p.t_cs, tq);   // the decoder's concat buffer
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
          for (int r = 0; r < 4; ++r) val[ns][r] += cf[ns][r].a * gq[ns][r] + cf[ns][r].b * tq[ns][r] + cf[ns][r].c0;
      }
      if (p.addend) {
        float aq[NS][4];
        loadq<T, NS>(reinterpret_cast<const T*>(p.addend) + ovox * (size_t)p.addend_ld + co0, aq);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
          for (int r = 0; r < 4; ++r) val[ns][r] += aq[ns][r];
      }
    }
    if (MODE == PW_CONVT) {
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[ns][r] += val[ns][r]; s2[ns][r] += val[ns][r] * val[ns][r]; }
    }
    if (MODE == PW_CONV1 && p.y2 != nullptr) {   // two dense destinations instead of channel slices of one buffer
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const int col = co0 + ns * 4;
        T* dst = col < p.ysplit ? yout + ovox * (size_t)p.y_ld + col : reinterpret_cast<T*>(p.y2) + ovox * (size_t)p.y2_ld + (col - p.ysplit);
        storeq<T, 1>(dst, val + ns);
      }
    } else if (!PL) {
      storeq<T, NS>(yout + ovox * (size_t)p.y_ld + co0, val);
    } else if (PERM) {   // two 16-byte pieces, one per plane
      T* vb = yout + ovox * (size_t)p.y_ld;
      storeq<T, 2>(vb + (size_t)(co0 >> 4) * p.y_cs + (co0 & 15), val);
      storeq<T, 2>(vb + (size_t)((co0 >> 4) + 1) * p.y_cs + (co0 & 15), val + (NS > 2 ? 2 : 0));
    } else {     // chunk-planar y (the transposed conv writes its planes of the concat buffer)
      storeq_planar<T, NS>(yout + ovox * (size_t)p.y_ld, co0, p.y_cs, val);
    }
  }

  };   // one_block
  // one pass for every mode but the transposed-conv forward (a compile-time fact: as a run-time loop it cost the 48-column GEMM 40 VGPRs and a wave per SIMD);
  // the ring's slots are compile-time indices: the loop is unrolled PD times
  if constexpr (PD == 1) {
    do { one_block(std::integral_constant<int, 0>{}); } while (MODE == PW_CONVT && (mb += p.mgroups) < p.mblocks);
  } else {
    for (;;) {
      one_block(std::integral_constant<int, 0>{});
      if ((mb += p.mgroups) >= p.mblocks) break;
      one_block(std::integral_constant<int, 1>{});
      if ((mb += p.mgroups) >= p.mblocks) break;
      if constexpr (PD > 2) {
        one_block(std::integral_constant<int, (PD > 2 ? 2 : 0)>{});
        if ((mb += p.mgroups) >= p.mblocks) break;
      }
      if constexpr (PD > 3) {
        one_block(std::integral_constant<int, (PD > 3 ? 3 : 0)>{});
        if ((mb += p.mgroups) >= p.mblocks) break;
      }
    }
  }

  mb = grp;                                      // row of this workgroup's partial sums
  if (MODE == PW_CONVT && p.part != nullptr) {
    __shared__ float red[4 * NS * 16 * 2];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[ns][r]), b = row16_sum(s2[ns][r]);   // over the 16 voxels of the lane row: DPP, no LDS
        if (j == 0) {
          red[((wave * NS * 16) + g * 4 * NS + ns * 4 + r) * 2 + 0] = a;   // slot = column offset inside the block
          red[((wave * NS * 16) + g * 4 * NS + ns * 4 + r) * 2 + 1] = b;
        }
      }
    __syncthreads();
    if (tid < NS * 16 * 2) {
      int c = tid >> 1, k = tid & 1;
      float a = red[(0 * NS * 16 + c) * 2 + k] + red[(1 * NS * 16 + c) * 2 + k] + red[(2 * NS * 16 + c) * 2 + k] +
                red[(3 * NS * 16 + c) * 2 + k];
      int col = col_base + c, sub = col / p.Csub, co = col % p.Csub;
      if (PERM) {   // slot c = g * 16 + ns * 4 + r of the permuted binding
        const int gg = c >> 4, ns = (c >> 2) & 3;
        sub = 2 * sp + (gg >> 1);
        co = pp * 32 + (ns >> 1) * 16 + (gg & 1) * 8 + (ns & 1) * 4 + (c & 3);
      }
      p.part[((((size_t)n * p.mgroups + mb) * (4 * p.sz) + sub) * 2 + k) * p.Csub + co] = a;
    }
  }
}


// ---- the decoder blocks' input gradient as a stream (round 4) ---------------------------------------------------------------------------------
//   out[v][0 .. 3K) = Wsc^T dOut[v] + a (.) g[v] + b (.) t[v] + c0,   columns [0, ysplit) -> y_lo, the rest -> y_hi   (K = 16 KC = channels of dOut)
// is pw_kernel<.., PW_CONV1> with the IN-backward affine and two destinations - 10 tensor units of traffic at level 0 and, as thousands of
// 128-voxel workgroups that each set up their weights / coefficients and then wait for their loads, 4.0 TB/s (0.66 ms of the cfg-2 step; round 3
// moved it neither with more or fewer voxels per workgroup nor with all loads up front).  Same arithmetic here, fed like wgrad_k1_dma_kernel:
// persistent workgroups, the 7 KC 16-channel operand chunks of a TV-voxel block (dOut, g, t) brought in by `buffer_load ... lds` through a ring
// of stages, weights / coefficients loaded once per workgroup (per sample), one barrier per block.
struct PwsParams {
  const void* x; int x_ld;                       // dOut (interleaved)
  const void* g; int g_ld; const void* t; int t_ld; int t_cs;   // g interleaved; t interleaved (t_cs = 16) or chunk-planar
  const void* wp; const bpx_nbwd_coef* coef;
  void* y; int y_ld; void* y2; int y2_ld; int ysplit;            // columns [0, ysplit) -> y, the rest -> y2
  int64_t vps; int N; int nblocks; int groups; int bps;          // bps = blocks per sample
  float* part;                                                   // WG instances: this launch's partial slabs [groups][3 K][K] of dWsc
};
__host__ __device__ constexpr int pws_vmcnt(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

// WG (round 6): the block's SHORTCUT WEIGHT GRADIENT dWsc[ci][co] = sum_v t[v][ci] dOut[v][co] rides along.  Both operands of that k = 1 weight gradient
// are in this kernel's LDS stages already (t = the raw block input, dOut); as a kernel of its own (wgrad_k1_dma_kernel) it read them from HBM a second
// time - 4 tensor units, 209 us at level 0 of cfg 2, 60 us at level 1.  Here wave w multiplies the 32-voxel K chunks it owns (KC = 1: chunk w of the four;
// KC = 2: chunk w & 1 of the two, t chunks 3 (w >> 1) ..) with transposing LDS reads exactly as that kernel does (t converted fp16 -> bf16 behind the
// read in the mixed mode), the waves' sums meet once after the last block, one slab per workgroup, reduced with the step's other weight gradients.
template <int KC, int TV, typename TT, bool WG = false>
__global__ void __launch_bounds__(256) pw_nbs_kernel(const PwsParams p) {
  constexpr int VB = 32, SUBS = TV / 32, NCH = 7 * KC, NS = 3 * KC, NCOL = 48 * KC, K = 16 * KC;
  constexpr int STAGE = NCH * TV * VB, RING = 4 * STAGE <= 131072 ? 4 : 3;
  constexpr int NI = NCH * SUBS;
  static_assert(NI % 4 == 0 && TV % 64 == 0, "DMA instructions divide over the four waves; every wave owns TV / 4 voxels");
  constexpr int IPW = NI / 4, MS = TV / 64;                       // 16-voxel m-subtiles per wave
  constexpr int ST = MS * NS;                                     // store instructions per wave and block (they share the vmcnt queue with the DMAs)
  static_assert(2 * IPW + 3 * ST < 64 && RING == 4, "vmcnt range; the wait counts below assume three blocks in flight");
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING * STAGE];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int grp = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.g), 0, (int)0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.t), 0, (int)0x80000000u, 0x00020000);
  const uint32_t xvb = (uint32_t)p.x_ld * 2u, gvb = (uint32_t)p.g_ld * 2u, tvb = (uint32_t)p.t_ld * 2u, tcb = (uint32_t)p.t_cs * 2u;
  const uint32_t lane_v = (uint32_t)(lane >> 1), lane_h = (uint32_t)(lane & 1) * 16u;
  const int64_t voxels = (int64_t)p.N * p.vps;
  const int groups = p.groups, nblocks = p.nblocks, bps = p.bps;

  // LDS image of a stage: chunk list [dOut: KC][g: 3 KC][t: 3 KC], each [TV voxels][32 B]
  auto issue = [=](int blk, int slot) {
    const int64_t v0 = (int64_t)blk * TV;
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
      const int q = wave + 4 * k;
      const int ch = q / SUBS, sub = q % SUBS;
      const int64_t v = v0 + sub * 32 + lane_v;
      const bool in = v < voxels;
      const uint32_t vv = (uint32_t)v;
      unsigned char* dst = const_cast<unsigned char*>(smem) + slot * STAGE + ch * TV * VB + sub * 1024;
      if (ch < KC) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)dst, 16, in ? vv * xvb + (uint32_t)ch * 32u + lane_h : 0x80000000u, 0, 0, 0);
      else if (ch < KC + NS) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)dst, 16, in ? vv * gvb + (uint32_t)(ch - KC) * 32u + lane_h : 0x80000000u, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t, (lds_ptr_t)dst, 16, in ? vv * tvb + (uint32_t)(ch - KC - NS) * tcb + lane_h : 0x80000000u, 0, 0, 0);
    }
  };

  // weights of the 1x1x1 operator, once: MFMA row i of column block ns <-> column (i / 4) * 4 NS + ns * 4 + i % 4, so that lane (g, j) ends up with
  // the 4 NS consecutive columns g * 4 NS .. of voxel j (pw_kernel's binding)
  const uint16_t* __restrict__ wp = reinterpret_cast<const uint16_t*>(p.wp);
  u32x4_t wf[NS];
  const bool kin = g * 8 < K;
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int colw = (j >> 2) * (4 * NS) + ns * 4 + (j & 3);
    wf[ns] = kin ? *reinterpret_cast<const u32x4_t*>(wp + ((size_t)g * NCOL + colw) * 8) : u32x4_t{0u, 0u, 0u, 0u};
  }
  const int col0 = g * 4 * NS;                                   // first of this lane's columns
  f32x4_t cf[NS][4];                                             // {a, b, c0, -} of this lane's columns in the current sample
  int cur_n = -1;

  // WG: this wave's K chunk and t chunks; accumulators [t chunk][dOut chunk]
  constexpr int WCH = 3, WKC = SUBS == 4 ? 1 : 2;                 // t chunks per wave; waves per ... (SUBS == 4: every wave owns one K chunk and all three t chunks)
  static_assert(!WG || (KC == 1 && SUBS == 4) || (KC == 2 && SUBS == 2), "weight-gradient split of the two instances");
  const int w_kc = SUBS == 4 ? wave : (wave & 1), w_c0 = SUBS == 4 ? 0 : (wave >> 1) * WCH;
  const int a_base = (g * 8 + (j >> 2)) * VB + (j & 3) * 8;       // lane (j, g): voxels 8g .. 8g + 7 of a K chunk through ds_read_b64_tr_b16
  f32x4_t wacc[WG ? WCH : 1][WG ? KC : 1];
#pragma unroll
  for (int c = 0; c < (WG ? WCH : 1); ++c)
#pragma unroll
    for (int d = 0; d < (WG ? KC : 1); ++d) wacc[c][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  (void)WKC;

  const int nst = grp < nblocks ? (nblocks - grp + groups - 1) / groups : 0;
#pragma unroll
  for (int s = 0; s < RING - 1; ++s)
    if (s < nst) issue(grp + s * groups, s);
  for (int s = 0; s < nst; ++s) {
    const int blk = grp + s * groups;
    const int n = blk / bps;
    if (n != cur_n) {                                             // (before the wait: these loads are older than nothing the wait needs)
      cur_n = n;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 4; ++r) cf[ns][r] = *reinterpret_cast<const f32x4_t*>(&p.coef[(size_t)n * NCOL + col0 + ns * 4 + r]);
    }
    // block s has landed when only what was issued after it is outstanding: the DMAs of the later blocks in flight and the stores of the blocks
    // computed since (VMEM operations of a wave retire in order)
    // (steady state, block s >= 3: after DMA(s) came stores(s-3), DMA(s+1), stores(s-2), DMA(s+2), stores(s-1); the first three blocks had fewer
    //  stores behind them; the last blocks, with fewer DMAs behind them, simply wait for more than they need)
    const int later = nst - 1 - s < RING - 2 ? nst - 1 - s : RING - 2;
    if (later >= 2 && RING >= 4) {
      if (s >= 3) __builtin_amdgcn_s_waitcnt(pws_vmcnt(2 * IPW + 3 * ST));
      else if (s == 2) __builtin_amdgcn_s_waitcnt(pws_vmcnt(2 * IPW + 2 * ST));
      else if (s == 1) __builtin_amdgcn_s_waitcnt(pws_vmcnt(2 * IPW + ST));
      else __builtin_amdgcn_s_waitcnt(pws_vmcnt(2 * IPW));
    } else if (later == 1) __builtin_amdgcn_s_waitcnt(pws_vmcnt(IPW));
    else __builtin_amdgcn_s_waitcnt(pws_vmcnt(0));
    __syncthreads();
    const unsigned char* st = smem + (s % RING) * STAGE;
    // (WG: the transposing reads come BEFORE the next block's DMAs are issued.  To the compiler's wait-count pass an LDS-DMA is a store to "some LDS" and a
    //  ds_read_b64_tr_b16 intrinsic may alias it: behind the issue it put `s_waitcnt vmcnt(0)` in front of the first such read - a wait for the block just
    //  requested, every iteration: no prefetch left (level 0: 0.46 -> 0.63 ms).  Here the only wait is the one the barrier above already has.)
    if constexpr (WG) {
      typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;
      u32x4_t gfw[KC];
#pragma unroll
      for (int d = 0; d < KC; ++d) {
        const unsigned char* q = st + d * TV * VB + a_base + w_kc * 32 * VB;
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q)));
        const u32x2_t h2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 4 * VB)));
        gfw[d] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
      }
#pragma unroll
      for (int c = 0; c < WCH; ++c) {
        const unsigned char* q = st + (KC + NS + w_c0 + c) * TV * VB + a_base + w_kc * 32 * VB;
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q)));
        const u32x2_t h2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 4 * VB)));
        u32x4_t afw = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
        if constexpr (sizeof(TT) == 2 && !std::is_same<TT, uint16_t>::value) {   // fp16 activations: bf16 MFMA operands, as wgrad_k1_dma_kernel
#pragma unroll
          for (int e = 0; e < 4; ++e) afw[e] = cvt_pk_bf16(lo16<f16_t>(afw[e]), hi16<f16_t>(afw[e]));
        }
#pragma unroll
        for (int d = 0; d < KC; ++d)
          wacc[c][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, afw), __builtin_bit_cast(bf16x8_t, gfw[d]), wacc[c][d], 0, 0, 0);
      }
    }
    if (s + RING - 1 < nst) issue(grp + (s + RING - 1) * groups, (s + RING - 1) % RING);
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
      const int vl = (wave * MS + ms) * 16 + j;                   // voxel inside the block
      const int64_t v = (int64_t)blk * TV + vl;
      f32x4_t acc[NS];
      const u32x4_t af = kin ? *reinterpret_cast<const u32x4_t*>(st + ((g >> 1) * TV + vl) * VB + (g & 1) * 16) : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) acc[ns] = mfma_step<uint16_t>(wf[ns], af, f32x4_t{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const int col = col0 + ns * 4;
        const u32x2_t gq = *reinterpret_cast<const u32x2_t*>(st + ((KC + (col >> 4)) * TV + vl) * VB + (col & 15) * 2);
        const u32x2_t tq = *reinterpret_cast<const u32x2_t*>(st + ((KC + NS + (col >> 4)) * TV + vl) * VB + (col & 15) * 2);
        const float gf[4] = {lo16<uint16_t>(gq[0]), hi16<uint16_t>(gq[0]), lo16<uint16_t>(gq[1]), hi16<uint16_t>(gq[1])};
        const float tf[4] = {lo16<TT>(tq[0]), hi16<TT>(tq[0]), lo16<TT>(tq[1]), hi16<TT>(tq[1])};
        float val[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          val[r] = acc[ns][r];
          val[r] += cf[ns][r][0] * gf[r] + cf[ns][r][1] * tf[r] + cf[ns][r][2];
        }
        if (v < voxels) {
          uint16_t* dst = col < p.ysplit ? reinterpret_cast<uint16_t*>(p.y) + (size_t)v * p.y_ld + col
                                         : reinterpret_cast<uint16_t*>(p.y2) + (size_t)v * p.y2_ld + (col - p.ysplit);
          *reinterpret_cast<u32x2_t*>(dst) = u32x2_t{pk16<uint16_t>(val[0], val[1]), pk16<uint16_t>(val[2], val[3])};
        }
      }
    }
  }
  if constexpr (WG) {
    // the waves' sums (fixed order) -> this workgroup's slab [3 K][K]: dWsc[ci][co], ci = t channel, co = dOut channel
    __builtin_amdgcn_s_waitcnt(pws_vmcnt(0));                       // (no DMA of this wave is still writing LDS)
    __syncthreads();
    f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);                // [wave][WCH][KC][64 lanes]
    static_assert(4 * WCH * KC * 64 * 16 <= RING * STAGE, "reduction scratch");
#pragma unroll
    for (int c = 0; c < WCH; ++c)
#pragma unroll
      for (int d = 0; d < KC; ++d) red[((wave * WCH + c) * KC + d) * 64 + lane] = wacc[c][d];
    __syncthreads();
    constexpr int CIN = 48 * KC, COUT = 16 * KC;
    float* pp = p.part + (size_t)grp * CIN * COUT;
    for (int q = tid; q < 3 * KC * KC * 64; q += 256) {             // (t chunk tc, dOut chunk d, lane)
      const int ln = q & 63, cn = q >> 6, tc = cn / KC, d = cn % KC;
      f32x4_t a;
      if (SUBS == 4) {                                              // four K chunks: waves 0 .. 3 hold (tc, d)
        a = (red[((0 * WCH + tc) * KC + d) * 64 + ln] + red[((1 * WCH + tc) * KC + d) * 64 + ln]) +
            (red[((2 * WCH + tc) * KC + d) * 64 + ln] + red[((3 * WCH + tc) * KC + d) * 64 + ln]);
      } else {                                                      // two K chunks: waves 2 (tc / 3) and 2 (tc / 3) + 1 hold t chunk tc as their c = tc % 3
        const int w0 = 2 * (tc / WCH), c = tc % WCH;
        a = red[((w0 * WCH + c) * KC + d) * 64 + ln] + red[(((w0 + 1) * WCH + c) * KC + d) * 64 + ln];
      }
      const int gi = ln >> 4, ii = ln & 15;
#pragma unroll
      for (int r = 0; r < 4; ++r) pp[(size_t)(tc * 16 + 4 * gi + r) * COUT + d * 16 + ii] = a[r];
    }
  }
}

// ---- transposed-conv forward with ONE K step into a chunk-planar buffer (round 5; level 0 of cfg 2: 32 -> 32 channels, 64^3 -> 128^3) -------------
// The same arithmetic, column binding (pw_kernel's PERM form) and statistics rows as pw_kernel<T, MS, 4, PW_CONVT, true>, as a loop WITHOUT BRANCHES.
// In pw_kernel the operand loads and the stores sit in predicated blocks (voxels beyond the volume, K < 32 lanes) and behind a run-time K loop; at
// every such join the compiler's wait-count pass gives up and emits `s_waitcnt vmcnt(0)` - a wave's VMEM operations retire in order, so that is a
// wait for every store issued so far: each block was [operand latency + write latency], 2.6 TB/s written where a plain store loop reaches 4.8-6.5
// (scripts/probes/store_pattern_probe.hip), and a deeper operand ring changed nothing.  Here operands and results move through BUFFER instructions:
// a lane that has nothing to load / store uses an out-of-range offset (loads return zeros, stores are dropped), there is no branch between the
// loop head and its end, the waits are counted ones, and the operands of the next PD blocks are in flight across the stores.
template <typename T>
__global__ void __launch_bounds__(256, BPX_CONVT_OCC) convt_k1_kernel(const PwParams p) {
  constexpr int MS = BPX_PW_MS_CT, NS = 4, KPL = 8, PD = BPX_CONVT_PD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int nbk = p.Ncols / 64;
  const int nb = blockIdx.x % nbk;
  const int grp = (blockIdx.x / nbk) % p.mgroups;
  const int n = blockIdx.x / (nbk * p.mgroups);
  const int ppb = p.Csub / 32;
  const int sp = nb / ppb, pp = nb - sp * ppb;
  const int sub = 2 * sp + (g >> 1), co0 = pp * 32 + (g & 1) * 8;          // this lane's sub-position and first channel (8 of each of two planes)
  const T* __restrict__ wp = reinterpret_cast<const T*>(p.wp);
  u32x4_t wf[NS];
  f32x2_t add[NS][2];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int gq = j >> 2;
    const int colw = (2 * sp + (gq >> 1)) * p.Csub + pp * 32 + (ns >> 1) * 16 + (gq & 1) * 8 + (ns & 1) * 4 + (j & 3);
    wf[ns] = *reinterpret_cast<const u32x4_t*>(wp + ((size_t)g * p.Ncols + colw) * KPL);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = co0 + (ns >> 1) * 16 + (ns & 1) * 4 + 2 * h;
      add[ns][h] = p.bias ? f32x2_t{p.bias[c], p.bias[c + 1]} : f32x2_t{0.f, 0.f};
    }
  }
  // Addresses: W % 16 == 0 (host-checked), so the 16 voxels of an m-subtile are one run of an x row, all inside or all outside the sample: the
  // (x, y, z) of the run's first voxel is WAVE-UNIFORM - kept in scalar registers and advanced by a fixed (dx, dy, dz) per block, no division in
  // the loop - and goes into the buffer instructions' scalar offset; the lane's share of the address is a constant.  (pw_kernel derives the
  // coordinates per lane and block: two 32-bit divisions per m-subtile, a third of its VALU work.)
  const uint32_t xrow = (uint32_t)p.x_ld * 2u;                              // bytes per input voxel
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.x) + (size_t)n * p.vps * p.x_ld), 0, (int)((uint32_t)p.vps * xrow), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
  const uint32_t OOB = 0xFFFFFFF0u;
  const uint32_t xlane = (g * KPL < p.K) ? (uint32_t)j * xrow + (uint32_t)(g * 16) : OOB;   // lanes beyond K read zeros
  const uint32_t W2 = 2u * (uint32_t)p.W, H2 = 2u * (uint32_t)p.H;
  const uint32_t yvox = (uint32_t)p.y_ld * 2u;                              // bytes per output voxel of one plane
  const uint32_t yplane = (uint32_t)p.y_cs * 2u;
  const uint32_t a_ = (sub >> 2) & 1, b_ = (sub >> 1) & 1, c_ = sub & 1;
  const uint32_t ylane = ((a_ * H2 + b_) * W2 + 2u * (uint32_t)j + c_) * yvox + ((uint32_t)(co0 >> 4) * (uint32_t)p.y_cs + (uint32_t)(co0 & 15)) * 2u;
  const uint32_t zbase = (uint32_t)n * (uint32_t)(p.sz * p.D);
  const uint32_t uW = (uint32_t)p.W, uH = (uint32_t)p.H;
  const uint32_t bstep = (uint32_t)p.mgroups * 4u * (uint32_t)(MS * 16);    // voxels between consecutive blocks of this workgroup
  uint32_t vb = ((uint32_t)grp * 4u + (uint32_t)wave) * (uint32_t)(MS * 16);   // first voxel of this wave in the current block (uniform)
  uint32_t x0, y0, z0;
  { const uint32_t row = vb / uW; x0 = vb - row * uW; z0 = row / uH; y0 = row - z0 * uH; }
  uint32_t dx, dy, dz;
  { const uint32_t row = bstep / uW; dx = bstep - row * uW; dz = row / uH; dy = row - dz * uH; }
  x0 = __builtin_amdgcn_readfirstlane(x0); y0 = __builtin_amdgcn_readfirstlane(y0); z0 = __builtin_amdgcn_readfirstlane(z0);
  dx = __builtin_amdgcn_readfirstlane(dx); dy = __builtin_amdgcn_readfirstlane(dy); dz = __builtin_amdgcn_readfirstlane(dz);
  u32x4_t af[PD][MS];
  auto request = [&](uint32_t vbn, auto slot_c) {       // operands of the block whose first voxel (of this wave) is vbn
    constexpr int SL = decltype(slot_c)::value;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
      const uint32_t v = vbn + (uint32_t)(ms * 16);
      const bool in = v < (uint32_t)p.vps;                                  // (uniform)
      af[SL][ms] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(in ? xlane : OOB), (int)(in ? v * xrow : 0u), 0));
    }
  };
  f32x2_t s1[NS][2], s2[NS][2];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns)
#pragma unroll
    for (int h = 0; h < 2; ++h) s1[ns][h] = s2[ns][h] = f32x2_t{0.f, 0.f};
  auto one_block = [&](auto slot_c) {
    constexpr int SL = decltype(slot_c)::value;
    f32x4_t acc[MS][NS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = mfma_step<T>(wf[ns], af[SL][ms], f32x4_t{0.f, 0.f, 0.f, 0.f});
    request(vb + (uint32_t)PD * bstep, slot_c);                             // (beyond the sample: out of range, zeros nobody reads)
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
      // the run's first voxel: (x0 + 16 ms, y0, z0) with one carry (16 ms <= W)
      uint32_t x = x0 + (uint32_t)(ms * 16), y = y0, z = z0;
      if (x >= uW) { x -= uW; if (++y >= uH) { y -= uH; ++z; } }
      const bool valid = vb + (uint32_t)(ms * 16) < (uint32_t)p.vps;        // (uniform)
      const uint32_t srun = (((zbase + (uint32_t)p.sz * z) * H2 + 2u * y) * W2 + 2u * x) * yvox;
      f32x2_t val[NS][2];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int h = 0; h < 2; ++h) val[ns][h] = f32x2_t{acc[ms][ns][2 * h], acc[ms][ns][2 * h + 1]} + add[ns][h];
      if (valid) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            s1[ns][h] = s1[ns][h] + val[ns][h];
            s2[ns][h] = s2[ns][h] + val[ns][h] * val[ns][h];
          }
      }
      uint32_t pk[NS][2];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2_t w2 = val[ns][h];
          const float lo = w2[0], hi = w2[1];
          pk[ns][h] = pk16s<T>(lo, hi);
        }
      const u32x4_t q0{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
      const u32x4_t q1{pk[2][0], pk[2][1], pk[3][0], pk[3][1]};
      // The run's offset goes into the VECTOR offset of the stores (one v_add each), not into their scalar offset: with an SGPR soffset hipcc 7.2
      // leaves out the wait states between a 128-bit buffer store and a VALU write of its data registers (its hazard recogniser assumes the
      // hazard away for a register soffset).  On gfx950 that holds for ONE store but not for the second of two back-to-back stores
      // (scripts/probes/soffset_store_hazard.hip: 16 of 256 words lost with no wait state, none with one) - which is this epilogue: the next
      // m-subtile's v_pk_add overwrote two of the second store's four data registers and the fp16 output held fp32 halves (NaNs; the bf16
      // instance happened to allocate differently).  The loads have no data registers to lose.
      __builtin_amdgcn_raw_buffer_store_b128(q0, rs_y, (int)(valid ? ylane + srun : OOB), 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(q1, rs_y, (int)(valid ? ylane + srun + yplane : OOB), 0, 0);
    }
    // the next block of this workgroup
    vb += bstep;
    x0 += dx; uint32_t cy = 0u, cz = 0u;
    if (x0 >= uW) { x0 -= uW; cy = 1u; }
    y0 += dy + cy;
    if (y0 >= uH) { y0 -= uH; cz = 1u; }
    z0 += dz + cz;
  };
  // The prologue issues what a steady-state block issues - a request, then 2 MS stores (out of range: dropped by the hardware, but counted) - so
  // that the wait count the compiler derives at the loop head for "slot 0 has landed" is the steady state's (2 (PD - 1) MS loads + 2 PD MS stores
  // may stay in flight) and not the entry path's: the count at a join is the smaller one, and with a bare prologue the first block of every
  // PD-block trip waited for all but 2 PD MS - 1 operations, i.e. for the stores of the whole previous trip.
  auto dummy_stores = [&](int k) {   // (distinct out-of-range offsets: identical stores would be merged; y_bytes < 0xFFFFFF00, host-checked)
#pragma unroll
    for (int q = 0; q < 2 * MS; ++q)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{0u, 0u, 0u, 0u}, rs_y, (int)(0xFFFFFF00u + (uint32_t)(((k * 2 * MS + q) & 15) * 16)), 0, 0);
  };
  request(vb, std::integral_constant<int, 0>{});
  dummy_stores(0);
  if (PD > 1) { request(vb + bstep, std::integral_constant<int, (PD > 1 ? 1 : 0)>{}); dummy_stores(1); }
  if (PD > 2) { request(vb + 2u * bstep, std::integral_constant<int, (PD > 2 ? 2 : 0)>{}); dummy_stores(2); }
  if (PD > 3) { request(vb + 3u * bstep, std::integral_constant<int, (PD > 3 ? 3 : 0)>{}); dummy_stores(3); }
  // full trips of PD blocks in a loop with ONE back edge, the last 0 .. PD - 1 blocks behind it.  (With an exit test after every block the exits
  // share a latch block that also leads back to the loop head, and along that path - never taken back, but the wait counts are static - only the
  // operations of ONE block follow slot 0's request: the head's wait became vmcnt(5) instead of vmcnt(17).)
  const int nblk = (p.mblocks - grp + p.mgroups - 1) / p.mgroups;
  int t = 0;
  for (; t + PD <= nblk; t += PD) {
    one_block(std::integral_constant<int, 0>{});
    if constexpr (PD > 1) one_block(std::integral_constant<int, (PD > 1 ? 1 : 0)>{});
    if constexpr (PD > 2) one_block(std::integral_constant<int, (PD > 2 ? 2 : 0)>{});
    if constexpr (PD > 3) one_block(std::integral_constant<int, (PD > 3 ? 3 : 0)>{});
  }
  if (t < nblk) { one_block(std::integral_constant<int, 0>{}); ++t; }
  if (PD > 2 && t < nblk) { one_block(std::integral_constant<int, (PD > 1 ? 1 : 0)>{}); ++t; }
  if (PD > 3 && t < nblk) { one_block(std::integral_constant<int, (PD > 2 ? 2 : 0)>{}); ++t; }
  if (p.part != nullptr) {   // one partial row per workgroup, pw_kernel's layout and summation order
    __shared__ float red[4 * NS * 16 * 2];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[ns][r >> 1][r & 1]), b = row16_sum(s2[ns][r >> 1][r & 1]);
        if (j == 0) {
          red[((wave * NS * 16) + g * 4 * NS + ns * 4 + r) * 2 + 0] = a;
          red[((wave * NS * 16) + g * 4 * NS + ns * 4 + r) * 2 + 1] = b;
        }
      }
    __syncthreads();
    if (tid < NS * 16 * 2) {
      const int c = tid >> 1, k = tid & 1;
      const float a = red[(0 * NS * 16 + c) * 2 + k] + red[(1 * NS * 16 + c) * 2 + k] + red[(2 * NS * 16 + c) * 2 + k] + red[(3 * NS * 16 + c) * 2 + k];
      const int gg = c >> 4, ns = (c >> 2) & 3;
      const int sub_c = 2 * sp + (gg >> 1);
      const int co = pp * 32 + (ns >> 1) * 16 + (gg & 1) * 8 + (ns & 1) * 4 + (c & 3);
      p.part[((((size_t)n * p.mgroups + grp) * (4 * p.sz) + sub_c) * 2 + k) * p.Csub + co] = a;
    }
  }
}

int g_pw_stream = 1;   // bpx_debug_set_pw_stream
int g_convt_k1 = -1;   // bpx_debug_set_convt_k1: -1 = environment (BPX_CONVT_K1, default on), 0 = pw_kernel, 1 = convt_k1_kernel where it applies
constexpr int PW_MS = 2;  // 4 waves x 2 x 16 = 128 voxels per workgroup
constexpr int PW_MS_CT = BPX_PW_MS_CT;   // the same for the transposed-conv forward.  Measured (32 -> 32 channels, 64^3 -> 128^3, planar output):
                                         // 1 (78 VGPRs, 5 waves / SIMD) 266 us, 2 266 us, 4 (162 VGPRs) 271 us - neither tile size nor residency binds

// transposed conv forward: at most this many persistent workgroups per (sample, 64-column block); with 4 samples x 4 column blocks (32 -> 32
// channels, level 0) that is 1024 workgroups of 4 waves, four per CU
inline int convt_groups(int64_t vps) { return (int)std::min<int64_t>(cdiv64(vps, 64 * PW_MS_CT), 64); }

inline int pw_ns(int ncols_per_group) { return (ncols_per_group % 64 == 0) ? 4 : (ncols_per_group % 48 == 0) ? 3 : (ncols_per_group % 32 == 0) ? 2 : 1; }

template <typename T, int MODE, typename TT = T>
int launch_pw(PwParams& p, int ns, hipStream_t s) {
  constexpr int MSK = (MODE == PW_CONVT) ? PW_MS_CT : PW_MS;
  if (p.vps >= (1ll << 31) - 64 * MSK) { bpx_set_error("pointwise kernels: more than 2^31 voxels per sample"); return 1; }
  p.mblocks = (int)cdiv64(p.vps, 64 * MSK);
  p.mgroups = MODE == PW_CONVT ? convt_groups(p.vps) : p.mblocks;
  int nbk = p.Ncols / (16 * ns);
  dim3 grid((unsigned)((int64_t)p.N * p.mgroups * nbk));
  const bool planar = (MODE == PW_CONV1 && p.coef != nullptr && p.t_cs != 16) || (MODE == PW_CONVT && p.y_cs != 16);
  if (MODE == PW_CONVT && planar && ns == 4 && p.Csub % 32 != 0) { bpx_set_error("transposed conv: the 64-column planar form needs Cout % 32 == 0"); return 1; }
  if (MODE != PW_CONVTD && planar) {
    constexpr bool PL = MODE != PW_CONVTD;    // no planar instances of the transposed-conv dgrad
    if constexpr (MODE == PW_CONVT && sizeof(T) == 2) {
      // one K step: the branch-free buffer-addressed kernel when x (per sample) and y lie within 32-bit byte offsets of their bases
      const int64_t ovox = (int64_t)p.N * p.vps * 4 * p.sz;
      const int64_t ybytes = ((int64_t)(p.Csub / 16 - 1) * p.y_cs + (ovox - 1) * p.y_ld + 16) * 2;
      static const bool k1_env = getenv("BPX_CONVT_K1") == nullptr || atoi(getenv("BPX_CONVT_K1")) != 0;   // A/B: BPX_CONVT_K1=0 = pw_kernel
      const bool k1 = g_convt_k1 < 0 ? k1_env : g_convt_k1 != 0;
      if (k1 && ns == 4 && p.K * 2 <= 64 && p.W % 16 == 0 && p.vps * p.x_ld * 2 < 0xFFFFFF00ll && ybytes < 0xFFFFFF00ll) {
        p.y_bytes = (uint32_t)ybytes;
        convt_k1_kernel<T><<<grid, 256, 0, s>>>(p);
        return 0;
      }
    }
    // one K step without the buffer-addressed kernel (W % 16 != 0, operands beyond 32-bit offsets, BPX_CONVT_K1 = 0): the round-4 instance that
    // keeps its weights in registers and requests the next blocks' operands ahead of the stores (it had lost its launch)
    if (ns == 4 && MODE == PW_CONVT && sizeof(T) == 2 && p.K * (int)sizeof(T) <= 64) pw_kernel<T, MSK, 4, MODE, PL, TT, MODE == PW_CONVT && sizeof(T) == 2><<<grid, 256, 0, s>>>(p);
    else if (ns == 4) pw_kernel<T, MSK, 4, MODE, PL, TT><<<grid, 256, 0, s>>>(p);
    else if (ns == 3) pw_kernel<T, MSK, 3, MODE, PL, TT><<<grid, 256, 0, s>>>(p);
    else if (ns == 2) pw_kernel<T, MSK, 2, MODE, PL, TT><<<grid, 256, 0, s>>>(p);
    else pw_kernel<T, MSK, 1, MODE, PL, TT><<<grid, 256, 0, s>>>(p);
    return 0;
  }
  if (ns == 4) pw_kernel<T, MSK, 4, MODE, false, TT><<<grid, 256, 0, s>>>(p);
  else if (ns == 3) pw_kernel<T, MSK, 3, MODE, false, TT><<<grid, 256, 0, s>>>(p);
  else if (ns == 2) pw_kernel<T, MSK, 2, MODE, false, TT><<<grid, 256, 0, s>>>(p);
  else pw_kernel<T, MSK, 1, MODE, false, TT><<<grid, 256, 0, s>>>(p);
  return 0;
}

int chk(const char* fn, const char* name, const bpx_tensor& t, int es) {
  BPX_CHECK(t.ptr != nullptr, "%s: %s.ptr is null", fn, name);
  BPX_CHECK(t.C % 16 == 0 && t.ld >= (t.cs ? 16 : t.C), "%s: %s needs C %% 16 == 0 and ld >= C (C=%d ld=%d)", fn, name, t.C, t.ld);
  BPX_CHECK(t.cs == 0 || t.cs % 8 == 0, "%s: %s has chunk stride %lld (must be a multiple of 8 elements)", fn, name, (long long)t.cs);
  BPX_CHECK(((uintptr_t)t.ptr % 16) == 0 && ((size_t)t.ld * es) % 16 == 0, "%s: %s must be 16-byte aligned", fn, name);
  return 0;
}

}  // namespace

extern "C" int bpx_debug_set_pw_stream(int on) { g_pw_stream = on; return 0; }
extern "C" int bpx_debug_set_convt_k1(int on) { g_convt_k1 = on; return 0; }
extern "C" int bpx_convT3d_stats_tiles(int D, int H, int W, int sz) { return convt_groups((int64_t)D * H * W) * 4 * (sz == 1 ? 1 : 2); }

static int conv1x1_impl(const char* fn, int dtype, int N, int64_t vps, bpx_tensor x, const void* w_packed_d, const float* bias_d,
                        bpx_tensor g, bpx_tensor t, const bpx_nbwd_coef* coef_d, bpx_tensor addend, bpx_tensor y, bpx_tensor y2,
                        bpx_stream_t stream, float* wg_part = nullptr, int* wg_groups = nullptr) {
  // wg_part (bpx_conv1x1_fwd_split_wgrad): the streaming kernel also forms the shortcut weight gradient's partial slabs there and reports how many
  BPX_CHECK(dtype == BPX_BF16 || dtype == BPX_F32 || dtype == BPX_MIX16 || dtype == BPX_F16,
            "%s: dtype must be BF16, F32, MIX16 (t fp16, everything else bf16) or F16 (plain forward GEMM, no IN-backward operands)", fn);
  const bool mix = dtype == BPX_MIX16 && coef_d != nullptr;
  if (dtype == BPX_MIX16) dtype = BPX_BF16;
  const bool f16 = dtype == BPX_F16;
  if (f16) BPX_CHECK(coef_d == nullptr, "%s: F16 is the forward GEMM only", fn);
  int es = (int)dtype_size(dtype);
  if (chk(fn, "x", x, es) || chk(fn, "y", y, es)) return 1;
  BPX_CHECK(w_packed_d, "%s: weights null", fn);
  if (coef_d) { if (chk(fn, "g", g, es) || chk(fn, "t", t, es)) return 1; }
  const int ncols = y.C + (y2.ptr ? y2.C : 0);
  if (y2.ptr) {
    if (chk(fn, "y_hi", y2, es)) return 1;
    BPX_CHECK(y.C % 4 == 0 && y2.C % 4 == 0, "%s: split outputs need channel counts that are multiples of 4 (got %d, %d)", fn, y.C, y2.C);
  }
  PwParams p{};
  p.N = N; p.D = 1; p.H = 1; p.W = 1; p.vps = vps;
  p.x = x.ptr; p.x_ld = x.ld; p.K = x.C; p.wp = w_packed_d; p.bias = bias_d;
  p.y = y.ptr; p.y_ld = y.ld; p.Ncols = ncols; p.Csub = ncols;
  p.y2 = y2.ptr; p.y2_ld = y2.ld; p.ysplit = y.C;
  p.g = g.ptr; p.g_ld = g.ld; p.t = t.ptr; p.t_ld = t.ld; p.coef = coef_d;
  p.t_cs = t.cs ? (int)t.cs : 16; p.y_cs = 16;
  BPX_CHECK(t.cs == 0 || (t.cs >= ((int64_t)N * vps - 1) * t.ld + 16 && t.cs < (1ll << 31)), "%s: t has chunk stride %lld for %lld voxels", fn, (long long)t.cs,
            (long long)N * vps);
  p.addend = addend.ptr; p.addend_ld = addend.ld;
  int ns = pw_ns(ncols);
  {   // the decoder blocks' input gradient at the large levels: the streaming kernel (pw_nbs_kernel)
    const int KC = x.C / 16;
    const int TV = KC == 1 ? 128 : 64;
    const int64_t vox = (int64_t)N * vps;
    auto span = [&](const bpx_tensor& q, int chunks) { return (q.cs ? (int64_t)q.cs * (chunks - 1) : 0) * 2 + vox * (int64_t)q.ld * 2; };
    const bool ok = g_pw_stream && dtype == BPX_BF16 && (KC == 1 || KC == 2) && x.C % 16 == 0 && coef_d != nullptr && y2.ptr != nullptr && y.C + y2.C == 3 * x.C && y.C % 4 == 0 &&
                    g.C == 3 * x.C && t.C == 3 * x.C && bias_d == nullptr && addend.ptr == nullptr && vps % TV == 0 && vox >= 262144 &&
                    span(x, 1) < (1ll << 31) && span(g, 1) < (1ll << 31) && span(t, 3 * KC) < (1ll << 31) && (x.ld & 7) == 0 && (g.ld & 7) == 0 && (t.ld & 7) == 0 &&
                    (y.ld & 3) == 0 && (y2.ld & 3) == 0 && (((uintptr_t)x.ptr | (uintptr_t)g.ptr | (uintptr_t)t.ptr | (uintptr_t)coef_d) & 15) == 0 &&
                    (((uintptr_t)y.ptr | (uintptr_t)y2.ptr) & 7) == 0;
    if (ok) {
      PwsParams q{};
      q.x = x.ptr; q.x_ld = x.ld; q.g = g.ptr; q.g_ld = g.ld; q.t = t.ptr; q.t_ld = t.ld; q.t_cs = t.cs ? (int)t.cs : 16;
      q.wp = w_packed_d; q.coef = coef_d; q.y = y.ptr; q.y_ld = y.ld; q.y2 = y2.ptr; q.y2_ld = y2.ld; q.ysplit = y.C;
      q.vps = vps; q.N = N; q.bps = (int)(vps / TV); q.nblocks = (int)(vox / TV);
      q.groups = (int)std::min<int64_t>(256, q.nblocks);        // one persistent workgroup per CU (the ring holds ~115 KB)
      q.part = wg_part;
      hipStream_t s = (hipStream_t)stream;
      if (wg_part != nullptr) {
        if (KC == 1) { if (mix) pw_nbs_kernel<1, 128, f16_t, true><<<q.groups, 256, 0, s>>>(q); else pw_nbs_kernel<1, 128, uint16_t, true><<<q.groups, 256, 0, s>>>(q); }
        else { if (mix) pw_nbs_kernel<2, 64, f16_t, true><<<q.groups, 256, 0, s>>>(q); else pw_nbs_kernel<2, 64, uint16_t, true><<<q.groups, 256, 0, s>>>(q); }
        *wg_groups = q.groups;
      } else if (KC == 1) { if (mix) pw_nbs_kernel<1, 128, f16_t><<<q.groups, 256, 0, s>>>(q); else pw_nbs_kernel<1, 128, uint16_t><<<q.groups, 256, 0, s>>>(q); }
      else { if (mix) pw_nbs_kernel<2, 64, f16_t><<<q.groups, 256, 0, s>>>(q); else pw_nbs_kernel<2, 64, uint16_t><<<q.groups, 256, 0, s>>>(q); }
      BPX_LAUNCH_CHECK(fn);
      return 0;
    }
  }
  BPX_CHECK(wg_part == nullptr, "%s: these operands do not take the streaming kernel (bpx_conv1x1_fwd_split_wgrad_workspace answers 0 for the shape classes it "
            "refuses; alignment, pitches and 32-bit spans must hold too): use bpx_conv1x1_fwd_split and bpx_conv3d_wgrad", fn);
  if ((mix ? launch_pw<uint16_t, PW_CONV1, f16_t>(p, ns, (hipStream_t)stream)
       : f16 ? launch_pw<f16_t, PW_CONV1>(p, ns, (hipStream_t)stream)
       : dtype == BPX_BF16 ? launch_pw<uint16_t, PW_CONV1>(p, ns, (hipStream_t)stream) : launch_pw<float, PW_CONV1>(p, ns, (hipStream_t)stream)) != 0) return 1;
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

// bpx_conv1x1_fwd_split with the block's shortcut weight gradient riding along (pw_nbs_kernel<.., WG = true>)
extern "C" int64_t bpx_conv1x1_fwd_split_wgrad_workspace(int dtype, int N, int64_t vps, int K) {
  static const bool on = getenv("BPX_PWS_WG") == nullptr || atoi(getenv("BPX_PWS_WG")) != 0;   // A/B: BPX_PWS_WG=0 = the two separate kernels
  if (!on || !g_pw_stream || (dtype != BPX_BF16 && dtype != BPX_MIX16) || (K != 16 && K != 32)) return 0;
  const int TV = K == 16 ? 128 : 64;
  if (vps % TV != 0 || (int64_t)N * vps < 262144) return 0;
  return (int64_t)256 * (3 * K) * K * 4;
}
extern "C" int bpx_conv1x1_fwd_split_wgrad(int dtype, int N, int64_t vps, bpx_tensor x, const void* w_packed_d, bpx_tensor g, bpx_tensor t,
                                           const bpx_nbwd_coef* coef_d, bpx_tensor y_lo, bpx_tensor y_hi, float* dw_d, void* ws_d, int64_t ws_bytes,
                                           bpx_stream_t stream) {
  const char* fn = "bpx_conv1x1_fwd_split_wgrad";
  BPX_CHECK(x.cs == 0 && g.cs == 0 && y_lo.cs == 0 && y_hi.cs == 0, "%s: only t may be chunk-planar", fn);
  BPX_CHECK(y_hi.ptr != nullptr && dw_d != nullptr && ws_d != nullptr && coef_d != nullptr, "%s: null pointer", fn);
  const int64_t need = bpx_conv1x1_fwd_split_wgrad_workspace(dtype, N, vps, x.C);
  BPX_CHECK(need > 0, "%s: unsupported here (dtype %d, %d channels, %lld voxels per sample): use bpx_conv1x1_fwd_split and bpx_conv3d_wgrad", fn, dtype, x.C, (long long)vps);
  BPX_CHECK(ws_bytes >= need && (((uintptr_t)ws_d) & 15) == 0, "%s: workspace too small or misaligned (%lld < %lld bytes)", fn, (long long)ws_bytes, (long long)need);
  int groups = 0;
  if (conv1x1_impl(fn, dtype, N, vps, x, w_packed_d, nullptr, g, t, coef_d, bpx_tensor{nullptr, 0, 0}, y_lo, y_hi, stream, reinterpret_cast<float*>(ws_d), &groups) != 0) return 1;
  // dWsc (Cout = x.C, Cin = t.C, 1, 1, 1): index co * Cin + ci; slabs [groups][Cin][Cout]
  return bpxred::reduce_partials(fn, reinterpret_cast<const float*>(ws_d), dw_d, groups, 1, t.C, x.C, 1, t.C, 1, nullptr, nullptr, 0, true, (hipStream_t)stream);
}

extern "C" int bpx_conv1x1_fwd(int dtype, int N, int64_t vps, bpx_tensor x, const void* w_packed_d, const float* bias_d,
                               bpx_tensor g, bpx_tensor t, const bpx_nbwd_coef* coef_d, bpx_tensor addend, bpx_tensor y,
                               bpx_stream_t stream) {
  BPX_CHECK(x.cs == 0 && g.cs == 0 && addend.cs == 0 && y.cs == 0, "bpx_conv1x1_fwd: only t may be chunk-planar");
  return conv1x1_impl("bpx_conv1x1_fwd", dtype, N, vps, x, w_packed_d, bias_d, g, t, coef_d, addend, y, bpx_tensor{nullptr, 0, 0}, stream);
}

extern "C" int bpx_conv1x1_fwd_split(int dtype, int N, int64_t vps, bpx_tensor x, const void* w_packed_d, const float* bias_d,
                                     bpx_tensor g, bpx_tensor t, const bpx_nbwd_coef* coef_d, bpx_tensor addend, bpx_tensor y_lo,
                                     bpx_tensor y_hi, bpx_stream_t stream) {
  BPX_CHECK(x.cs == 0 && g.cs == 0 && addend.cs == 0 && y_lo.cs == 0 && y_hi.cs == 0, "bpx_conv1x1_fwd_split: only t may be chunk-planar");
  const char* fn = "bpx_conv1x1_fwd_split";
  BPX_CHECK(y_hi.ptr != nullptr, "%s: y_hi null", fn);
  return conv1x1_impl(fn, dtype, N, vps, x, w_packed_d, bias_d, g, t, coef_d, addend, y_lo, y_hi, stream);
}

extern "C" int bpx_convT3d_k2s2_fwd(int dtype, int N, int D, int H, int W, int sz, bpx_tensor x, const void* w_packed_d, const float* bias_d,
                                    bpx_tensor y, float* stats_part_d, bpx_stream_t stream) {
  BPX_CHECK(x.cs == 0, "bpx_convT3d_k2s2_fwd: only y may be chunk-planar");
  BPX_CHECK(y.cs == 0 || (y.cs >= ((int64_t)N * D * H * W * 4 * sz - 1) * y.ld + 16 && y.cs < (1ll << 31)), "bpx_convT3d_k2s2_fwd: y has chunk stride %lld",
            (long long)y.cs);
  const char* fn = "bpx_convT3d_k2s2_fwd";
  BPX_CHECK(dtype == BPX_BF16 || dtype == BPX_F32 || dtype == BPX_F16, "%s: dtype must be BF16, F16 or F32", fn);
  BPX_CHECK(sz == 1 || sz == 2, "%s: z stride must be 1 or 2 (got %d)", fn, sz);
  int es = (int)dtype_size(dtype);
  if (chk(fn, "x", x, es) || chk(fn, "y", y, es)) return 1;
  BPX_CHECK(w_packed_d, "%s: weights null", fn);
  PwParams p{};
  p.N = N; p.D = D; p.H = H; p.W = W; p.sz = sz; p.vps = (int64_t)D * H * W;
  p.x = x.ptr; p.x_ld = x.ld; p.K = x.C; p.wp = w_packed_d; p.bias = bias_d;
  p.y = y.ptr; p.y_ld = y.ld; p.Ncols = 4 * sz * y.C; p.Csub = y.C; p.part = stats_part_d;
  p.t_cs = 16; p.y_cs = y.cs ? (int)y.cs : 16;
  // 64 columns per workgroup column block for dense and chunk-planar outputs: the 16 columns of lanes g and g + 2 are then the two x
  // sub-positions (2x, 2x + 1) of the same channels, i.e. ONE contiguous 64 / 128-byte piece of the output per input voxel and store
  // instruction instead of two half-written lines from different workgroups (32 -> 32 channels, 64^3 -> 128^3, B = 4: 308 -> 256 us;
  // into a 32-of-48-channel slice of an interleaved buffer the wide form is slower, 362 -> 415 us, and is not used)
  static const bool wide = getenv("BPX_CONVT_NS") == nullptr;
  int ns = (wide && (y.cs != 0 ? y.C % 32 == 0 : y.ld == y.C) && (4 * sz * y.C) % 64 == 0) ? 4 : pw_ns(y.C);
  if ((dtype == BPX_BF16 ? launch_pw<uint16_t, PW_CONVT>(p, ns, (hipStream_t)stream)
       : dtype == BPX_F16 ? launch_pw<f16_t, PW_CONVT>(p, ns, (hipStream_t)stream)
                          : launch_pw<float, PW_CONVT>(p, ns, (hipStream_t)stream)) != 0) return 1;
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

extern "C" int bpx_convT3d_k2s2_dgrad(int dtype, int N, int D, int H, int W, int sz, bpx_tensor dy, const void* w_packed_T_d, bpx_tensor dx,
                                      bpx_stream_t stream) {
  BPX_CHECK(dy.cs == 0 && dx.cs == 0, "bpx_convT3d_k2s2_dgrad: chunk-planar tensors (cs != 0) are not accepted here");
  const char* fn = "bpx_convT3d_k2s2_dgrad";
  BPX_CHECK(dtype == BPX_BF16 || dtype == BPX_F32, "%s: dtype must be BF16 or F32", fn);
  BPX_CHECK(sz == 1 || sz == 2, "%s: z stride must be 1 or 2 (got %d)", fn, sz);
  int es = (int)dtype_size(dtype);
  if (chk(fn, "dy", dy, es) || chk(fn, "dx", dx, es)) return 1;
  BPX_CHECK(w_packed_T_d, "%s: weights null", fn);
  PwParams p{};
  p.N = N; p.D = D; p.H = H; p.W = W; p.sz = sz; p.vps = (int64_t)D * H * W;
  p.x = dy.ptr; p.x_ld = dy.ld; p.K = 4 * sz * dy.C; p.Csub = dy.C; p.wp = w_packed_T_d;
  p.y = dx.ptr; p.y_ld = dx.ld; p.Ncols = dx.C; p.t_cs = 16; p.y_cs = 16;
  int ns = pw_ns(dx.C);
  if ((dtype == BPX_BF16 ? launch_pw<uint16_t, PW_CONVTD>(p, ns, (hipStream_t)stream) : launch_pw<float, PW_CONVTD>(p, ns, (hipStream_t)stream)) != 0) return 1;
  BPX_LAUNCH_CHECK(fn);
  return 0;
}
