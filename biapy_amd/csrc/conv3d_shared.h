// Declarations shared by the implicit-GEMM conv kernels (conv3d_igemm.hip: double-buffered 4-wave kernel, fp32 exact mode
// and small bf16 layers; conv3d_lean.hip: lean persistent bf16 kernel for the large layers).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "bpx_common.h"

namespace bpxconv {

enum { EPI_FWD = 0, EPI_DGRAD = 1 };

struct Conv3Params {
  int N, D, H, W;
  const void* x; int x_ld; int Cin;
  const bpx_norm_rec* in_norm; int act;
  const void* wp; const float* bias;
  const void* sc; int sc_ld; int sc_C; const void* wsc; const float* bias_sc;
  void* y; int y_ld; int Cout;
  float* part;  // [N][tiles][2][Cout]
  const void* t; int t_ld; const bpx_norm_rec* t_norm; int t_act;
  // fused MaxPool3d (pool_sz,2,2) of the output (forward, lean kernel only): pooled tensor + its statistics partials
  void* pool; int pool_ld; int pool_sz; float* pool_part;
  int tilesY, tilesX, tilesPerSample, totalTiles, tilesPerXcd;
  int tilesZ;  // conv3d_zmarch.hip only
  long long* stamps;  // profiling: per-workgroup s_memtime stamps [block][16] (BPX_CONV_STAMPS), else null
  int dbg;  // ablation switches for profiling (BPX_CONV_DBG): 1 = skip MFMA steps, 2 = skip staging transform+loads
  // distance in ELEMENTS between consecutive 16-channel chunks of a voxel: 16 for the ordinary interleaved layout, the plane size for
  // chunk-planar tensors (bpx_tensor.cs)
  int x_cs, sc_cs, y_cs, t_cs;
  int f16;   // 16-bit storage is fp16 instead of bf16 (forward only)
  int t_f16; // dgrad, BPX_MIX16: the activation operand `t` is fp16 while dy / weights / g are bf16
  int t_dma; // dgrad: t lies within 2 GB of its base, i.e. the lean kernel's >= 32-channel instances can fetch their t tile with `buffer_load ... lds`
             // (they have no other path: the launcher keeps tensors beyond that on the plain kernel)
  int ps;    // forward, lean kernel only: 3-D pixel shuffle by `ps` fused into the store (bpx_conv3d_fwd_shuffle).  Cout = 16 * ps^3 ordered
             // [sub-position (a, b, e)][16 channels]; y is the (N, ps*D, ps*H, ps*W, 16) tensor, block (a, b, e) of voxel (z, y, x) goes to
             // voxel (ps*z + a, ps*y + b, ps*x + e): the ps^3-fold channel tensor never exists in memory
};
inline int chunk_stride(const bpx_tensor& t) { return t.cs ? (int)t.cs : 16; }

// ACTK = 1: ELU known at compile time (the reference default) - no per-element control flow; ACTK = 0: runtime switch.
template <typename T, int ACTK = 0> __device__ __forceinline__ float apply_act_rt(float u, int act) {
  constexpr bool PRECISE = std::is_same<T, float>::value;
  if (ACTK == 1) return u > 0.f ? u : (PRECISE ? expm1f(u) : (__expf(u) - 1.f));
  return bpx_act_rt<PRECISE, ACTK == 2>(u, act);     // ACTK = 2: the plain kernels' instances that also take the codes 4-8
}
template <typename T, int ACTK = 0> __device__ __forceinline__ float apply_act_bwd_rt(float u, int act) {
  constexpr bool PRECISE = std::is_same<T, float>::value;
  if (ACTK == 1) return u > 0.f ? 1.f : (PRECISE ? expf(u) : __expf(u));
  return bpx_act_bwd_rt<PRECISE, ACTK == 2>(u, act);
}


template <int HY, int HX, int VB> __device__ __forceinline__ constexpr int tap_off(int tap) {
  return (((tap / 9) * HY + ((tap / 3) % 3)) * HX + (tap % 3)) * VB;
}

// Tile walk of the persistent kernels.  XCD x owns the contiguous id range [x T/8, (x + 1) T/8) and its co-resident workgroups sit on
// consecutive ids; ids enumerate a sample's tiles in Y-STRIPS: (strip of `stripY` tile rows, z, row inside the strip, x).  The ~100 tiles an
// XCD works on at a time then form a block of a few z-planes x stripY rows x all x instead of half a z-plane: the y and z neighbours whose
// halos a tile re-reads were fetched by the same XCD a moment ago and sit in its 4 MB L2 (z-major order: the z neighbour is tilesY * tilesX
// ids away and long evicted).  strip_rows: ~32 tiles per z step of a strip.
__host__ __device__ inline int strip_rows(int tilesX) { const int r = 32 / (tilesX > 0 ? tilesX : 1); return r < 1 ? 1 : r; }
__device__ __forceinline__ void decode_tile(int id, int tilesZ, int tilesY, int tilesX, int tilesPerSample, int stripY, int& n, int& tzi, int& tyi, int& txi) {
  n = id / tilesPerSample;
  const int r = id - n * tilesPerSample;
  const int per_strip = stripY * tilesZ * tilesX;          // tiles of a full strip (every strip before the last one is full)
  const int s = r / per_strip, rs = r - s * per_strip;
  const int left = tilesY - s * stripY, sy = left < stripY ? left : stripY;
  const int row = sy * tilesX;
  tzi = rs / row;
  const int r2 = rs - tzi * row, yin = r2 / tilesX;
  txi = r2 - yin * tilesX;
  tyi = s * stripY + yin;
}

struct TileCfg { int tz, ty, tx, ns; };

inline TileCfg pick_cfg(int dtype, int D, int H, int W, int Cout) {
  TileCfg c;
  // 16*NS output channels per workgroup; 48/96-channel outputs (dgrad into the concat buffers) take NS = 3 so that the
  // halo is staged once instead of three times
  c.ns = (Cout % 64 == 0) ? 4 : (Cout % 48 == 0) ? 3 : (Cout % 32 == 0) ? 2 : 1;
  if (c.ns == 4 && (int64_t)D * H * W <= 512) c.ns = 2;   // 8^3 bottleneck: twice the workgroups (29 -> 23, 62 -> 48 us)
  // NS = 2 at 16^3 too was measured (round 2): fwd 384->128 92 -> 84 us, 128->128+sc 49 -> 44, but dgrad 128->384 61 -> 76: a wash; at
  // 32^3 it loses 25-35 % everywhere
  // <= 16^3 volumes: 4x4x8 tiles double the workgroup count of these latency-bound launches (measured 114 -> 92, 61 -> 48 us)
  if (W > 8 && (int64_t)D * H * W > 4096) {
    c.tx = 16; c.tz = 4;
    bool big = (dtype == BPX_BF16 || dtype == BPX_F16) && c.ns == 1 && (int64_t)D * H * W >= 32768 && H >= 8 && getenv("BPX_SMALL_TILE") == nullptr;
    c.ty = big ? 8 : 4;
  } else {
    c.tx = 8; c.tz = 4; c.ty = 4;
  }
  return c;
}

extern long long* g_conv_stamps;  // profiling hook (bpx_debug_set_conv_stamps)

// lean persistent bf16 kernel (conv3d_lean.hip) - the production kernel of the >= 64^3 layers
int launch_conv3_lean(int epi, const Conv3Params& p, const TileCfg& c, hipStream_t s);
// z-marching forward kernel (conv3d_zmarch.hip) for the 16-output-channel layers of the big tile; 1 = not applicable, take the lean kernel (same bits)
int launch_conv3_zm(const Conv3Params& p, const TileCfg& c, hipStream_t s);
// (The DMA-pipelined forward / dgrad schedule of round 3 - conv3d_dma.hip: halo by LDS-DMA into a second buffer, in-place transform - measured
//  EQUAL to the lean kernel on every cfg-2 layer (profiles/r03_conv_dma_vs_lean.txt) and was deleted in round 4; the LDS-DMA staging lives on in
//  the fused backward kernel, bwd_fused.hip, where the operands need no transform or are transformed once per voxel.)

}  // namespace bpxconv
