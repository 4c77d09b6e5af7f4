// Weight gradients as a reduction-over-voxels GEMM on the matrix cores.
//
//   dW[i = ci][j = co] (per tap) = sum_{v} A[v (+tap)][ci] * G[v][co]
//
// Both operands are stored channel-contiguous ([voxel][channel]) but the MFMA wants the REDUCTION
// index (voxels) contiguous per lane, i.e. both operands transposed.  bf16: the gfx950 LDS
// transpose read (ds_read_b64_tr_b16: a 16-lane group reads a [4 voxel][16 channel] block and lane
// i receives channel i of the 4 voxels) delivers exactly that with no shuffles and, because its
// alignment requirement is along the CHANNEL axis, works unchanged for the +-1 voxel tap shifts.
// f32: v_mfma_f32_16x16x4_f32 takes one scalar per lane, a plain ds_read_b32 gather.
//
//   block   : 256 threads, one (tile-group, 16-channel ci chunk, 16*NS co block).  The activation
//             halo (normalisation + activation applied while staging, exactly as in the forward
//             kernel) and the matching dy tile are staged in LDS once per tile and reused by all 27 taps.
//   waves   : TAPS=27 -> wave w owns taps {w, w+4, ...} (<= 7 accumulators x NS);
//             TAPS=1  -> wave w owns K-chunks {w, w+4, ...} of the tile.
//   output  : every block walks all tiles of its group, then stores its partial dW (plain coalesced
//             stores into a [groups][taps][Cin][Cout] workspace - fp32 atomics measured 14 G/s, i.e. 20x
//             slower than the same bytes streamed); a second tiny kernel sums the groups in a fixed
//             order and writes the PyTorch-layout gradient, so the result is deterministic.
//             Bias gradient: column sums from the ci-chunk-0 blocks (few atomics).
#include <algorithm>
#include <type_traits>
#include <vector>

#include "conv3d_shared.h"
#include "wgrad_shared.h"

namespace {
using bpxwg::sd_mfma_phase;

struct WgradParams {
  int N, D, H, W;          // logical voxel grid of the reduction
  const void* x; int x_ld; int Cin; const bpx_norm_rec* in_norm; int act;
  int x_cs;                // elements between the 16-channel chunks of an x voxel: 16, or the plane size of a chunk-planar tensor
  int x_f16;               // BPX_MIX16: x (the forward pass's activation tensor) is fp16; it is converted to bf16 while it is staged, dy is bf16
  const void* dy; int dy_ld; int Cout; int dy_vs; int dy_oz, dy_oy, dy_ox;  // dy voxel = vs*v + off (ConvTranspose)
  int dy_vz;               // z stride of that mapping (0 = same as dy_vs): kernel (1,2,2) has vz = 1
  float* part;                                   // [groups][taps][Cin][Cout] per-block partial sums (workspace)
  float* dw; int64_t si, sj, st; int64_t off;   // final dW index = ci*si + co*sj + tap*st + off (reduce kernel)
  float* db;                                     // bias gradient (accumulated into by the reduce kernel), or null
  float* dbpart;                                 // [groups][Cout] per-group column sums of dy (workspace, behind `part`): no atomics
  float* db2;                                    // host side only: a second destination of the bias gradient (reduce kernel), or null
  int tilesY, tilesX, tilesPerSample, totalTiles, groups;
  int tilesZ, stripY, order;   // order 1 (windowed shift-dy kernel): XCD-contiguous tile ranges walked in y-strips (conv3d_shared.h decode_tile); 0: tile = group + k * groups
};

template <typename T, int ACTK = 0> __device__ __forceinline__ float act_rt(float u, int act) {
  constexpr bool PRECISE = std::is_same<T, float>::value;
  if (ACTK == 1) return u > 0.f ? u : (PRECISE ? expm1f(u) : (__expf(u) - 1.f));
  return bpx_act_rt<PRECISE, ACTK == 2>(u, act);
}

// Stage EZ*EY*EX voxels x NCH channels (NCH multiple of 16/GPT pieces) into LDS [voxel][NCH].
// Source voxel = vs*(o+h)+a; voxels whose LOGICAL coordinate (o+h) is outside [0,D)x[0,H)x[0,W) are zero.
template <typename T, int EZ, int EY, int EX, int NCH, int ACTK = 0>
__device__ __forceinline__ void stage_any(unsigned char* smem, const T* __restrict__ src, int ld, int c0, int n, int D, int H, int W,
                                          int oz, int oy, int ox, int vs, int az, int ay, int ax,
                                          const bpx_norm_rec* __restrict__ norm, int C_norm, int act, int tid) {
  constexpr int KPL = ElemTraits<T>::KPL;
  constexpr int PPV = NCH / KPL;  // 16-byte pieces per voxel
  constexpr int PIECES = EZ * EY * EX * PPV;
  constexpr int VB = NCH * (int)sizeof(T);
  static_assert(256 % PPV == 0, "piece index must be thread-invariant");
  const int sub = tid % PPV;
  float sc[KPL], sh[KPL];
  if (norm) {
#pragma unroll
    for (int e = 0; e < KPL; ++e) {
      bpx_norm_rec r = norm[(size_t)n * C_norm + c0 + sub * KPL + e];
      sc[e] = r.scale; sh[e] = r.shift;
    }
  }
  const T* sbase = src + c0 + sub * KPL;
  const int Dp = D * vs, Hp = H * vs, Wp = W * vs;
  constexpr int UNR = 4;
  for (int base = 0; base < PIECES; base += 256 * UNR) {
    u32x4_t buf[UNR];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      int idx = base + u * 256 + tid;
      ok[u] = false;
      buf[u] = u32x4_t{0u, 0u, 0u, 0u};
      if (idx < PIECES) {
        int hv = idx / PPV;
        int hx = hv % EX, hy = (hv / EX) % EY, hz = hv / (EX * EY);
        int gz = oz + hz, gy = oy + hy, gx = ox + hx;
        if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) {
          ok[u] = true;
          size_t vox = (((size_t)n * Dp + (gz * vs + az)) * Hp + (gy * vs + ay)) * Wp + (gx * vs + ax);
          buf[u] = *reinterpret_cast<const u32x4_t*>(sbase + vox * (size_t)ld);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      int idx = base + u * 256 + tid;
      if (idx < PIECES) {
        u32x4_t v = buf[u];
        if (norm && ok[u]) {
          float f[KPL];
          unpack16<T>(v, f);
#pragma unroll
          for (int e = 0; e < KPL; ++e) f[e] = fmaf(sc[e], f[e], sh[e]);
          if (ACTK == 1) {
#pragma unroll
            for (int e = 0; e < KPL; ++e) f[e] = act_rt<T, 1>(f[e], act);
          } else {
            bpx_act_vec<std::is_same<T, float>::value, KPL, ACTK == 2>(f, act);
          }
          v = pack16<T>(f);
        }
        *reinterpret_cast<u32x4_t*>(smem + (size_t)(idx / PPV) * VB + sub * 16) = v;
      }
    }
  }
}

// bf16 transposed fragment: 8 voxels (k) of channel (lane&15), voxels at `vox_byte` + {0..7}*VB.
// Lane supplies the address of ITS 8-byte piece of the [4 voxel][16 ch] block: voxel (lane&15)/4,
// channels ((lane&15)%4)*4..+3; the hardware hands lane i column i.
template <int VB, bool USE_TR>
__device__ __forceinline__ u32x4_t frag_T_bf16(const unsigned char* smem, int vox_byte, int ch_byte, int i) {
  u32x4_t out;
  if (USE_TR) {
    const unsigned char* p = smem + vox_byte + (i >> 2) * VB + ch_byte + (i & 3) * 8;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 4 * VB));
    u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
    out[0] = l2[0]; out[1] = l2[1]; out[2] = h2[0]; out[3] = h2[1];
  } else {
    const unsigned char* p = smem + vox_byte + ch_byte + i * 2;
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const uint16_t*>(p + k * VB);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = v[2 * k] | (v[2 * k + 1] << 16);
  }
  return out;
}

template <typename T, int TZ, int TY, int TX, int NS, int TAPS, bool USE_TR, int ACTK>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradParams p) {
  using Tr = ElemTraits<T>;
  constexpr bool BF = std::is_same<T, uint16_t>::value;
  constexpr int HALO = (TAPS == 27) ? 1 : 0;
  constexpr int HZ = TZ + 2 * HALO, HY = TY + 2 * HALO, HX = TX + 2 * HALO, HV = HZ * HY * HX;
  constexpr int VBA = 16 * (int)sizeof(T);        // activation tile: 16 channels per voxel
  constexpr int CB = 16 * NS;                     // dy tile: CB channels per voxel
  constexpr int VBG = CB * (int)sizeof(T);
  constexpr int TV = TZ * TY * TX;
  constexpr int KC = BF ? 32 : 4;                 // voxels per MFMA step
  constexpr int NKC = TV / KC;
  constexpr int NT = (TAPS == 27) ? 7 : 1;        // accumulator sets per wave
  static_assert(TX % 8 == 0, "a lane's 8-voxel run must stay inside one x row");
  __shared__ __attribute__((aligned(16))) unsigned char smem[HV * VBA + TV * VBG + 4 * CB * 4];
  unsigned char* sA = smem;
  unsigned char* sG = smem + HV * VBA;
  float* sB = reinterpret_cast<float*>(smem + HV * VBA + TV * VBG);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8 (observed dispatch).  The ci-chunks of one tile group read the
  // SAME dy tiles and the same 128-byte lines of x (16 of Cin channels each), so they are given consecutive ids on
  // one XCD - the private L2 then serves the re-reads (PMC before: FETCH_SIZE 2.2x the algorithmic bytes, L2 hit 30 %).
  const int nchunks_ = p.Cin / 16, groups8 = (p.groups + 7) & ~7;
  const int per_cb = groups8 * nchunks_;
  const int cbi = (int)blockIdx.x / per_cb, rem = (int)blockIdx.x % per_cb;
  const int chunk = (rem % (8 * nchunks_)) / 8;
  const int grp = (rem / (8 * nchunks_)) * 8 + rem % 8;
  const int co_base = cbi * CB;
  if (grp >= p.groups) return;   // padding workgroups (whole block, before any barrier)

  f32x4_t acc[NT][NS];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) acc[a][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  const T* __restrict__ xin = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ gin = reinterpret_cast<const T*>(p.dy);

  // ---- staging plan.  16-byte pieces; the LDS images are piece-linear (piece idx at byte idx*16).  The global loads of
  //      the NEXT tile are issued before the MFMA phase of the current one and land in registers (async-STAGE split);
  //      they are normalised / activated / written to LDS after the barrier that ends the MFMA phase.
  constexpr int KPL = Tr::KPL, GPT = 16 / KPL, PPVG = CB / KPL;
  constexpr int NPA = (HV * GPT + 255) / 256, NPG = (TV * PPVG + 255) / 256;
  static_assert(256 % GPT == 0 && 256 % PPVG == 0, "piece sub-index must be thread-invariant");
  const int subA = tid % GPT, subG = tid % PPVG;
  u32x4_t pa[NPA], pg[NPG];
  uint32_t va = 0;  // bit u: piece u of the activation halo is inside the volume (gets the prologue)
  float psc[KPL], psh[KPL];
  const int vz = p.dy_vz ? p.dy_vz : p.dy_vs;
  const int Dp = p.D * vz, Hp = p.H * p.dy_vs, Wp = p.W * p.dy_vs;
  int n_cur = -1;

  auto issue_loads = [&](int tt) {
    const int n = tt / p.tilesPerSample, tile = tt % p.tilesPerSample;
    const int z0 = (tile / (p.tilesX * p.tilesY)) * TZ, y0 = ((tile / p.tilesX) % p.tilesY) * TY, x0 = (tile % p.tilesX) * TX;
    va = 0;
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
      const int idx = u * 256 + tid, hv = idx / GPT;
      const int gz = z0 - HALO + hv / (HX * HY), gy = y0 - HALO + (hv / HX) % HY, gx = x0 - HALO + hv % HX;
      pa[u] = u32x4_t{0u, 0u, 0u, 0u};
      if (idx < HV * GPT && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
        pa[u] = *reinterpret_cast<const u32x4_t*>(xin + ((((size_t)n * p.D + gz) * p.H + gy) * p.W + gx) * (size_t)p.x_ld + (size_t)chunk * p.x_cs + subA * KPL);
        va |= 1u << u;
      }
    }
#pragma unroll
    for (int u = 0; u < NPG; ++u) {
      const int idx = u * 256 + tid, t = idx / PPVG;
      const int gz = z0 + t / (TY * TX), gy = y0 + (t / TX) % TY, gx = x0 + t % TX;
      pg[u] = u32x4_t{0u, 0u, 0u, 0u};
      if (idx < TV * PPVG && gz < p.D && gy < p.H && gx < p.W) {
        const size_t vox = (((size_t)n * Dp + (gz * vz + p.dy_oz)) * Hp + (gy * p.dy_vs + p.dy_oy)) * Wp + (gx * p.dy_vs + p.dy_ox);
        pg[u] = *reinterpret_cast<const u32x4_t*>(gin + vox * (size_t)p.dy_ld + co_base + subG * KPL);
      }
    }
    if (p.in_norm && n != n_cur) {
#pragma unroll
      for (int e = 0; e < KPL; ++e) {
        bpx_norm_rec r = p.in_norm[(size_t)n * p.Cin + chunk * 16 + subA * KPL + e];
        psc[e] = r.scale; psh[e] = r.shift;
      }
    }
    n_cur = n;
  };
  auto write_staged = [&]() {
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
      const int idx = u * 256 + tid;
      if (idx < HV * GPT) {
        u32x4_t v = pa[u];
        if ((p.in_norm || (BF && p.x_f16)) && ((va >> u) & 1u)) {
          float f[KPL];
          if constexpr (BF) {   // fp16 activations beside bf16 gradients (uniform switch): convert on the way to the bf16 MFMA operand
            if (p.x_f16) unpack16<f16_t>(v, f); else unpack16<T>(v, f);
          } else {
            unpack16<T>(v, f);
          }
          if (p.in_norm) {
#pragma unroll
            for (int e = 0; e < KPL; ++e) f[e] = fmaf(psc[e], f[e], psh[e]);
            if (ACTK == 1) {
#pragma unroll
              for (int e = 0; e < KPL; ++e) f[e] = act_rt<T, 1>(f[e], p.act);
            } else {
              bpx_act_vec<std::is_same<T, float>::value, KPL, ACTK == 2>(f, p.act);
            }
          }
          v = pack16<T>(f);
        }
        *reinterpret_cast<u32x4_t*>(sA + (size_t)idx * 16) = v;
      }
    }
#pragma unroll
    for (int u = 0; u < NPG; ++u) {
      const int idx = u * 256 + tid;
      if (idx < TV * PPVG) *reinterpret_cast<u32x4_t*>(sG + (size_t)idx * 16) = pg[u];
    }
  };

  // NB: the norm record of a tile must be the one its loads were issued with: psc/psh are refreshed in issue_loads, and
  // write_staged of tile k runs BEFORE issue_loads(k+1), so they still belong to tile k.
  if (grp < p.totalTiles) issue_loads(grp);
  for (int tt = grp; tt < p.totalTiles; tt += p.groups) {
    __syncthreads();          // previous MFMA phase has finished reading the LDS tiles
    write_staged();
    __syncthreads();
    if (tt + p.groups < p.totalTiles) issue_loads(tt + p.groups);

    if (p.db != nullptr && chunk == 0) {  // bias gradient: column sums of the dy tile
      const int c = tid % CB;
      for (int v = tid / CB; v < TV; v += 256 / CB) bsum += Tr::ld(reinterpret_cast<const T*>(sG + (size_t)v * VBG) + c);
    }

    // K loop over the tile's voxels: fetch every operand of a K-chunk (dy fragments + the activation fragments of all of
    // the wave's taps) first, then issue the MFMAs back to back - one LDS wait per K-chunk instead of one per MFMA.
    constexpr int KSTEP = (TAPS == 27) ? 1 : 4;
    auto fetch = [&](int kc, u32x4_t* gfv, u32x4_t* afv) {
      const int t0 = kc * KC + (BF ? g * 8 : g);
      const int tz = t0 / (TY * TX), ty = (t0 / TX) % TY, tx = t0 % TX;
      const int hbase = ((tz * HY + ty) * HX + tx) * VBA;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        if (BF) gfv[ns] = frag_T_bf16<VBG, USE_TR>(sG, t0 * VBG, ns * 32, i);
        else gfv[ns][0] = *reinterpret_cast<const uint32_t*>(sG + (size_t)t0 * VBG + (ns * 16 + i) * 4);
      }
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        int tap = (TAPS == 27) ? (wave + 4 * a) : 0;
        if (TAPS == 27 && tap > 26) tap = 26;   // wave 3 has one tap less: its 7th accumulator is computed but never flushed
        const int toff = (TAPS == 27) ? ((((tap / 9) * HY + ((tap / 3) % 3)) * HX + (tap % 3)) * VBA) : 0;
        if (BF) afv[a] = frag_T_bf16<VBA, USE_TR>(sA, hbase + toff, 0, i);
        else afv[a][0] = *reinterpret_cast<const uint32_t*>(sA + hbase + toff + i * 4);
      }
    };
    auto mma = [&](const u32x4_t* gfv, const u32x4_t* afv) {
#pragma unroll
      for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          if (BF) acc[a][ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, afv[a]), __builtin_bit_cast(bf16x8_t, gfv[ns]), acc[a][ns], 0, 0, 0);
          else acc[a][ns] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(afv[a][0]), __uint_as_float(gfv[ns][0]), acc[a][ns], 0, 0, 0);
        }
      }
    };
    // (a second register set to prefetch K-chunk k+1 was measured: 230 VGPRs -> one wave per SIMD -> 1.4-1.6x SLOWER;
    //  co-resident workgroups hide the LDS latency better than a deeper pipeline inside one wave.)
    u32x4_t gfA[NS], afA[NT];
    for (int kc = (TAPS == 27) ? 0 : wave; kc < NKC; kc += KSTEP) {
      fetch(kc, gfA, afA);
      __builtin_amdgcn_sched_barrier(0);
      mma(gfA, afA);
    }
  }

  // flush: lane holds D[ci = 4g+r][co = i]
  if (TAPS == 27) {
    float* pp = p.part + (size_t)grp * 27 * p.Cin * p.Cout;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      int tap = wave + 4 * a;
      if (tap >= 27) continue;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int ci = chunk * 16 + 4 * g + r, co = co_base + ns * 16 + i;
          pp[((size_t)tap * p.Cin + ci) * p.Cout + co] = acc[0 + a][ns][r];
        }
    }
  } else {
    // TAPS == 1: the four waves hold partial sums over disjoint voxel subsets -> combine through LDS
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][NS][64 lanes][4]
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * NS + ns) * 64 + lane) * 4 + r] = acc[0][ns][r];
    __syncthreads();
    float* pp = p.part + (size_t)grp * p.Cin * p.Cout;
    for (int q = tid; q < NS * 64 * 4; q += 256) {
      int r = q & 3, ln = (q >> 2) & 63, ns = q >> 8;
      float s = red[((0 * NS + ns) * 64 + ln) * 4 + r] + red[((1 * NS + ns) * 64 + ln) * 4 + r] + red[((2 * NS + ns) * 64 + ln) * 4 + r] +
                red[((3 * NS + ns) * 64 + ln) * 4 + r];
      int ci = chunk * 16 + 4 * (ln >> 4) + r, co = co_base + ns * 16 + (ln & 15);
      pp[(size_t)ci * p.Cout + co] = s;
    }
  }
  if (p.db != nullptr && chunk == 0) {
    __syncthreads();
    // reduce the 256/CB partial sums per channel through LDS
    float* red = reinterpret_cast<float*>(smem);
    red[tid] = bsum;
    __syncthreads();
    if (tid < CB) {
      float s = 0.f;
      for (int k = tid; k < 256; k += CB) s += red[k];
      p.dbpart[(size_t)grp * p.Cout + co_base + tid] = s;   // one writer per (group, channel)
    }
  }
  (void)sB;
}


// ---- 3x3x3 wgrad, bf16, large volumes: "shift dy" orientation ----------------------------------------------------------
// dW[tap][ci][co] = sum_v A[v + tap - 1][ci] * G[v][co] = sum_u A[u][ci] * G[u - (tap - 1)][co]:
// the tile of ACTIVATIONS is staged un-haloed (each voxel is normalised + activated exactly once - the prologue is the
// VALU bottleneck of this kernel, and a haloed activation tile repeats it 2.5x), and the halo goes to dy, which is a raw
// copy.  The activation fragment of a K-chunk is then shared by all taps; each tap reads its own shifted dy fragment.
// Lean schedule as in conv3d_lean.hip: no register prefetch across the MFMA phase, <= 128 (NS=1) / 168 VGPRs, 29 / 50 KB
// LDS -> 4 / 3 workgroups per CU; 32-bit byte offsets; per-tile index math reduced to base + per-lane constants.
template <int NS, int ACTK, bool XF16 = false>
__global__ void __launch_bounds__(256, NS == 1 ? 4 : 3) wgrad_sd_kernel(const WgradParams p) {
  using T = uint16_t;
  using TXE = typename std::conditional<XF16, f16_t, uint16_t>::type;   // element type of x (BPX_MIX16: fp16)
  constexpr int TZ = 4, TY = 4, TX = 16, TV = TZ * TY * TX;
  constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;
  constexpr int KPL = 8, VBA = 32, CB = 16 * NS, VBG = CB * 2, PPVG = 2 * NS;
  constexpr int NKC = TV / 32, NT = 7;
  constexpr int NPA = TV * 2 / 256, NPGT = HV * PPVG, NPG = (NPGT + 255) / 256;
  static_assert(256 % PPVG == 0, "piece sub-index must be thread-invariant");
  __shared__ __attribute__((aligned(16))) unsigned char smem[TV * VBA + HV * VBG];
  unsigned char* sA = smem;
  unsigned char* sG = smem + TV * VBA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int D = p.D, H = p.H, W = p.W;
  // workgroup -> (tile group, ci chunk, co block): same XCD-aware order as wgrad_kernel above
  const int nchunks_ = p.Cin / 16, groups8 = (p.groups + 7) & ~7;
  const int per_cb = groups8 * nchunks_;
  const int cbi = (int)blockIdx.x / per_cb, rem = (int)blockIdx.x % per_cb;
  const int chunk = (rem % (8 * nchunks_)) / 8;
  const int grp = (rem / (8 * nchunks_)) * 8 + rem % 8;
  const int co_base = cbi * CB;
  if (grp >= p.groups) return;

  f32x4_t acc[NT][NS];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) acc[a][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  const char* __restrict__ xin = reinterpret_cast<const char*>(p.x);
  const char* __restrict__ gin = reinterpret_cast<const char*>(p.dy);

  // ---- per-workgroup constants: byte offsets of this thread's pieces relative to the tile / halo origin -----------------
  uint32_t rel_a[NPA], rel_g[NPG];
#pragma unroll
  for (int u = 0; u < NPA; ++u) {
    const int t = (u * 256 + tid) >> 1;
    rel_a[u] = (uint32_t)((((t >> 6) * H + ((t >> 4) & 3)) * W + (t & 15)) * p.x_ld + chunk * p.x_cs + (tid & 1) * KPL) * 2u;
    asm volatile("" : "+v"(rel_a[u]));
  }
  const int subG = tid % PPVG;
#pragma unroll
  for (int u = 0; u < NPG; ++u) {
    const int hv = (u * 256 + tid) / PPVG;
    const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
    rel_g[u] = (uint32_t)(((hz * H + hy) * W + hx) * p.dy_ld + co_base + subG * KPL) * 2u;
    asm volatile("" : "+v"(rel_g[u]));
  }
  const bool last_ok = (NPG - 1) * 256 + tid < NPGT;
  // fragment bases: lane (i, g) reads voxels 8g..8g+7 of a 32-voxel K-chunk (one x run), channel block via the tr-read
  const int trl = (i >> 2), trc = (i & 3) * 8;
  const int a_base = g * 8 * VBA + trl * VBA + trc;                                          // + kc*32*VBA
  const int g_lane = (((g >> 1) * HX + (g & 1) * 8) + trl) * VBG + trc;                      // + kc part + tap shift + ns*32
  int g_base[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    int tap = wave + 4 * a;
    if (tap > 26) tap = 26;  // wave 3 has one tap less: its 7th accumulator is computed but never flushed
    const int dz = tap / 9, dy_ = (tap / 3) % 3, dx = tap % 3;
    g_base[a] = g_lane + (((2 - dz) * HY + (2 - dy_)) * HX + (2 - dx)) * VBG;
  }
  float psc[KPL], psh[KPL];
  int n_cur = -1;

  for (int tt = grp; tt < p.totalTiles; tt += p.groups) {
    const int n = tt / p.tilesPerSample, tile = tt - n * p.tilesPerSample;
    const int z0 = (tile / (p.tilesX * p.tilesY)) * TZ, y0 = ((tile / p.tilesX) % p.tilesY) * TY, x0 = (tile % p.tilesX) * TX;
    const bool full = z0 + TZ <= D && y0 + TY <= H && x0 + TX <= W;
    const bool interior = full && z0 >= 1 && z0 + TZ + 1 <= D && y0 >= 1 && y0 + TY + 1 <= H && x0 >= 1 && x0 + TX + 1 <= W;
    const uint32_t base_a = (uint32_t)(((n * D + z0) * H + y0) * W + x0) * (uint32_t)p.x_ld * 2u;
    const uint32_t base_g = (uint32_t)(((n * D + z0 - 1) * H + (y0 - 1)) * W + (x0 - 1)) * (uint32_t)p.dy_ld * 2u;

    // dy pieces move in batches of GB registers: everything in flight at once for NS = 1; two batches for NS = 2 so that
    // the kernel stays below 168 VGPRs (3 workgroups per CU) without scratch
    constexpr int GB = (NS == 1) ? NPG : 6;
    u32x4_t pa[NPA], pg[GB];
    bool oka[NPA];
    auto load_g = [&](int b0) {
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const int u = b0 + q;
        pg[q] = u32x4_t{0u, 0u, 0u, 0u};
        if (u < NPG) {
          bool ok = (u < NPG - 1) || last_ok;
          if (!interior) {
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));
            const int hv = (u * 256 + tid_o) / PPVG;
            const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
            ok = ok && (unsigned)(z0 - 1 + hz) < (unsigned)D && (unsigned)(y0 - 1 + hy) < (unsigned)H && (unsigned)(x0 - 1 + hx) < (unsigned)W;
          }
          if (ok) pg[q] = *reinterpret_cast<const u32x4_t*>(gin + (base_g + rel_g[u]));
        }
      }
    };
    auto store_g = [&](int b0) {
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const int u = b0 + q;
        if (u < NPG && (u < NPG - 1 || last_ok)) *reinterpret_cast<u32x4_t*>(sG + (size_t)(u * 256 + tid) * 16) = pg[q];
      }
    };
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
      const int t = (u * 256 + tid) >> 1;
      oka[u] = full || (z0 + (t >> 6) < D && y0 + ((t >> 4) & 3) < H && x0 + (t & 15) < W);
      pa[u] = u32x4_t{0u, 0u, 0u, 0u};
      if (oka[u]) pa[u] = *reinterpret_cast<const u32x4_t*>(xin + (base_a + rel_a[u]));
    }
    load_g(0);
    if (p.in_norm && n != n_cur) {
#pragma unroll
      for (int e = 0; e < KPL; ++e) {
        const f32x2_t ss = *reinterpret_cast<const f32x2_t*>(&p.in_norm[(size_t)n * p.Cin + chunk * 16 + (tid & 1) * KPL + e].scale);
        psc[e] = ss[0]; psh[e] = ss[1];
      }
      n_cur = n;
    }
    __syncthreads();  // the previous tile's MFMA phase has finished reading LDS
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
      u32x4_t v = pa[u];
      if ((p.in_norm || XF16) && oka[u]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a = lo16<TXE>(v[q]), b = hi16<TXE>(v[q]);
          if (p.in_norm) {
            a = fmaf(psc[2 * q], a, psh[2 * q]); b = fmaf(psc[2 * q + 1], b, psh[2 * q + 1]);
            act_pair<ACTK>(a, b, p.act);
          }
          v[q] = cvt_pk_bf16(a, b);
        }
      }
      *reinterpret_cast<u32x4_t*>(sA + (size_t)(u * 256 + tid) * 16) = v;
    }
    store_g(0);
#pragma unroll
    for (int b0 = GB; b0 < NPG; b0 += GB) { load_g(b0); store_g(b0); }
    __syncthreads();

    if (p.db != nullptr && chunk == 0) {  // bias gradient: column sums of dy over the tile's own voxels
      const int c = tid % CB;
      for (int v = tid / CB; v < TV; v += 256 / CB) {
        const int hidx = (((v >> 6) + 1) * HY + ((v >> 4) & 3) + 1) * HX + (v & 15) + 1;
        bsum += bf16_to_f32(*reinterpret_cast<const uint16_t*>(sG + (size_t)hidx * VBG + c * 2));
      }
    }

#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      const int ka = kc * 32 * VBA;
      const int kg = (((kc >> 1) * HY + (kc & 1) * 2) * HX) * VBG;
      u32x4_t af, gf[NT][NS];
      {
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(sA + a_base + ka));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(sA + a_base + ka + 4 * VBA));
        u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
        af = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
      }
      // taps in groups of TG: all dy fragments of a group are fetched before its MFMAs (one LDS wait per group); NS = 2
      // uses two groups to halve the fragment registers
      constexpr int TG = (NS == 1) ? NT : 4;
#pragma unroll
      for (int a0 = 0; a0 < NT; a0 += TG) {
#pragma unroll
        for (int a = a0; a < a0 + TG && a < NT; ++a)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) {
            const unsigned char* q = sG + g_base[a] + kg + ns * 32;
            s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(q));
            s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(q + 4 * VBG));
            u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
            gf[a][ns] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = a0; a < a0 + TG && a < NT; ++a)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
            acc[a][ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, gf[a][ns]), acc[a][ns], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // flush: lane holds D[ci = 4g+r][co = i] of its taps
  float* pp = p.part + (size_t)grp * 27 * p.Cin * p.Cout;
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    const int tap = wave + 4 * a;
    if (tap >= 27) continue;
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = chunk * 16 + 4 * g + r, co = co_base + ns * 16 + i;
        pp[((size_t)tap * p.Cin + ci) * p.Cout + co] = acc[a][ns][r];
      }
  }
  if (p.db != nullptr && chunk == 0) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    red[tid] = bsum;
    __syncthreads();
    if (tid < CB) {
      float s = 0.f;
      for (int k = tid; k < 256; k += CB) s += red[k];
      p.dbpart[(size_t)grp * p.Cout + co_base + tid] = s;   // one writer per (group, channel)
    }
  }
}


// ---- shift-dy wgrad, MC input-channel chunks per workgroup (Cout block = 16) -----------------------------------------------
// wgrad_sd_kernel issues 2 + 14 transposing LDS reads per 7 MFMAs and K-chunk; measured, those reads bound it (the kernel
// runs at the same speed with or without software pipelining of the fragments).  The shifted dy fragments do not depend on
// the input-channel chunk, so a workgroup that owns MC chunks of the same tiles re-uses every dy fragment MC times:
// 2*MC + 14 reads per 7*MC MFMAs (MC = 2: 1.29 reads per MFMA instead of 2.29).  The InstanceNorm scale/shift of the MC*16
// channels live in LDS (they would cost 32*MC registers), everything else follows wgrad_sd_kernel.
//
// Second saving: the three taps of one (dz, dy) row use dy fragments that are the same 8-voxel x-window slid by one voxel.
// A lane reads the row's 10..12-voxel window once (3 transposing reads) and derives the three fragments in registers
// (v_alignbit_b32 for the odd shift) instead of 6 reads.  Wave w owns the taps [7w, 7w+7) in (dz, dy, dx) order, i.e. 2-3 row
// segments; the four waves run four compile-time specialisations of the MFMA phase.  LDS reads per K-chunk and workgroup:
// 64 -> 39 (MC = 1), 72 -> 47 (MC = 2), 80 -> 55 (MC = 3).
// the two-voxel shift of the window is an odd-aligned register quad (not a legal MFMA operand: the compiler copies it with four v_mov);
// reading it again from LDS instead measured 1-2.5 % faster on the 128^3 / 64^3 layers (16->16 256 -> 251 us, 96->32 256 -> 250), flat elsewhere.
// BPX_WGRAD_REREAD = 2 reads the one-voxel shift from LDS too (two reads instead of a third window read + four v_alignbit per tap; the
// kernel is VALU-bound, the LDS has slack): 48->16 @128^3 508 -> 477 us, 96->32 @64^3 231 -> 224, 32->32 88 -> 84.5, 16->16 flat
template <int MC, int ACTK, bool XF16 = false>
__global__ void __launch_bounds__(256, MC == 1 ? 4 : MC == 2 ? 3 : 2) wgrad_sdm_kernel(const WgradParams p) {
  using TXE = typename std::conditional<XF16, f16_t, uint16_t>::type;   // element type of x (BPX_MIX16: fp16)
  constexpr int TZ = 4, TY = 4, TX = 16, TV = TZ * TY * TX;
  constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;
  constexpr int KPL = 8, VBA = 32, CB = 16, VBG = CB * 2, PPVG = 2;
  constexpr int NKC = TV / 32, NT = 7;
  constexpr int NPA = TV * 2 / 256, NPGT = HV * PPVG, NPG = (NPGT + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char smem[MC * TV * VBA + HV * VBG + 64 + MC * 16 * 8];   // 64: window over-read of the last halo row
#if BPX_WGRAD_HC == 2
  __shared__ u32x4_t sHC[256];   // halo coordinates of every thread's dy pieces (16 bits each: hz | hy << 4 | hx << 8), read by the edge tiles only
#endif
  unsigned char* sA = smem;                                   // [MC][TV][32 B]
  unsigned char* sG = smem + MC * TV * VBA;
  float* sN = reinterpret_cast<float*>(smem + MC * TV * VBA + HV * VBG + 64);   // [MC*16][scale, shift]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int D = p.D, H = p.H, W = p.W;
  const int ncg = p.Cin / (16 * MC), groups8 = (p.groups + 7) & ~7;   // chunk groups
  const int per_cb = groups8 * ncg;
  const int cbi = (int)blockIdx.x / per_cb, rem = (int)blockIdx.x % per_cb;
  const int cgi = (rem % (8 * ncg)) / 8;
  const int grp = (rem / (8 * ncg)) * 8 + rem % 8;
  const int co_base = cbi * CB, ci_base = cgi * 16 * MC;
  if (grp >= p.groups) return;

  f32x4_t acc[NT][MC];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int c = 0; c < MC; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool want_b = p.db != nullptr && cgi == 0;            // wave 3 keeps the column sums of dy (the bias gradient) in acc[6][0], see sd_mfma_phase

  const char* __restrict__ xin = reinterpret_cast<const char*>(p.x);
  const char* __restrict__ gin = reinterpret_cast<const char*>(p.dy);
  static_assert(NPG <= 8, "eight 16-bit coordinate fields per thread");
  uint32_t rel_a[NPA], rel_g[NPG], hc_g[BPX_WGRAD_HC == 2 ? 4 : (NPG + 1) / 2];   // hc_g: halo coordinates of the thread's dy pieces, two per register, 16 bits each:
  //                                                          hz | hy << 4 | hx << 8 (edge tiles test them)
#pragma unroll
  for (int u = 0; u < (BPX_WGRAD_HC == 2 ? 4 : (NPG + 1) / 2); ++u) hc_g[u] = 0u;
#pragma unroll
  for (int u = 0; u < NPA; ++u) {
    const int t = (u * 256 + tid) >> 1;
    rel_a[u] = (uint32_t)((((t >> 6) * H + ((t >> 4) & 3)) * W + (t & 15)) * p.x_ld + cgi * MC * p.x_cs + (tid & 1) * KPL) * 2u;
    asm volatile("" : "+v"(rel_a[u]));
  }
  const int subG = tid % PPVG;
#pragma unroll
  for (int u = 0; u < NPG; ++u) {
    const int hv = (u * 256 + tid) / PPVG;
    const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
    rel_g[u] = (uint32_t)(((hz * H + hy) * W + hx) * p.dy_ld + co_base + subG * KPL) * 2u;
#if BPX_WGRAD_HC
    hc_g[u >> 1] |= ((uint32_t)hz | ((uint32_t)hy << 4) | ((uint32_t)hx << 8)) << (16 * (u & 1));
#endif
    asm volatile("" : "+v"(rel_g[u]));
  }
#if BPX_WGRAD_HC == 1
#pragma unroll
  for (int u = 0; u < (NPG + 1) / 2; ++u) asm volatile("" : "+v"(hc_g[u]));
#elif BPX_WGRAD_HC == 2
  sHC[tid] = u32x4_t{hc_g[0], hc_g[1], hc_g[2], hc_g[3]};   // read back by the same thread: no barrier needed
#endif
  const bool last_ok = (NPG - 1) * 256 + tid < NPGT;
  const int trl = (i >> 2), trc = (i & 3) * 8;
  const int a_base = g * 8 * VBA + trl * VBA + trc;
  const int g_lane = (((g >> 1) * HX + (g & 1) * 8) + trl) * VBG + trc;
  int n_cur = -1;

  // Tile walk.  order 0 (until round 3): tile = group + k * groups - consecutive groups sit on different XCDs (block b runs on XCD b % 8), so the
  // dy halo a tile shares with its x / y / z neighbours was re-fetched from HBM by up to three XCDs (PMC: 1.65x the algorithmic bytes).
  // order 1: the groups of XCD x (grp % 8 == x) own the contiguous id range [x T/nx, (x + 1) T/nx) and walk it side by side in y-strips.
  const int nx = p.groups < 8 ? p.groups : 8;
  const int wxcd = grp & 7, wslot = grp >> 3, wspx = (p.groups - wxcd + 7) >> 3, tilesPerXcd = (p.totalTiles + nx - 1) / nx;
  const int k_end = p.order ? tilesPerXcd : p.totalTiles, k_step = p.order ? wspx : p.groups;
  for (int kk = p.order ? wslot : grp; kk < k_end; kk += k_step) {
    int n, tzi, tyi, txi;
    if (p.order) {
      const int id = wxcd * tilesPerXcd + kk;
      if (id >= p.totalTiles) break;
      bpxconv::decode_tile(id, p.tilesZ, p.tilesY, p.tilesX, p.tilesPerSample, p.stripY, n, tzi, tyi, txi);
    } else {
      n = kk / p.tilesPerSample;
      const int tile = kk - n * p.tilesPerSample;
      tzi = tile / (p.tilesX * p.tilesY); tyi = (tile / p.tilesX) % p.tilesY; txi = tile % p.tilesX;
    }
    const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;
    const bool full = z0 + TZ <= D && y0 + TY <= H && x0 + TX <= W;
    const bool interior = full && z0 >= 1 && z0 + TZ + 1 <= D && y0 >= 1 && y0 + TY + 1 <= H && x0 >= 1 && x0 + TX + 1 <= W;
    const uint32_t base_a = (uint32_t)(((n * D + z0) * H + y0) * W + x0) * (uint32_t)p.x_ld * 2u;
    const uint32_t base_g = (uint32_t)(((n * D + z0 - 1) * H + (y0 - 1)) * W + (x0 - 1)) * (uint32_t)p.dy_ld * 2u;

    u32x4_t pa[MC][NPA], pg[NPG];
    bool oka[NPA];
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
      const int t = (u * 256 + tid) >> 1;
      oka[u] = full || (z0 + (t >> 6) < D && y0 + ((t >> 4) & 3) < H && x0 + (t & 15) < W);
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        pa[c][u] = u32x4_t{0u, 0u, 0u, 0u};
        if (oka[u]) pa[c][u] = *reinterpret_cast<const u32x4_t*>(xin + (base_a + rel_a[u]) + (uint32_t)c * ((uint32_t)p.x_cs * 2u));
      }
    }
#if BPX_WGRAD_HC == 2
    u32x4_t hcv = u32x4_t{0u, 0u, 0u, 0u};
    if (!interior) hcv = sHC[tid];
#endif
#pragma unroll
    for (int u = 0; u < NPG; ++u) {
      pg[u] = u32x4_t{0u, 0u, 0u, 0u};
      bool ok = (u < NPG - 1) || last_ok;
      if (!interior) {   // a third of the 128^3 tiles: three field extracts per piece (the divisions by the halo extents were ~40 VALU instructions each)
#if BPX_WGRAD_HC
        const uint32_t c = (BPX_WGRAD_HC == 2 ? hcv[u >> 1] : hc_g[u >> 1]) >> (16 * (u & 1));
        ok = ok && (unsigned)(z0 - 1 + (int)(c & 15u)) < (unsigned)D && (unsigned)(y0 - 1 + (int)((c >> 4) & 15u)) < (unsigned)H &&
             (unsigned)(x0 - 1 + (int)((c >> 8) & 255u)) < (unsigned)W;
#else
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        const int hv = (u * 256 + tid_o) / PPVG;
        const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
        ok = ok && (unsigned)(z0 - 1 + hz) < (unsigned)D && (unsigned)(y0 - 1 + hy) < (unsigned)H && (unsigned)(x0 - 1 + hx) < (unsigned)W;
#endif
      }
      if (ok) pg[u] = *reinterpret_cast<const u32x4_t*>(gin + (base_g + rel_g[u]));
    }
    const bool new_n = p.in_norm && n != n_cur;   // uniform
    if (new_n && tid < MC * 16) {
      const f32x2_t ss = *reinterpret_cast<const f32x2_t*>(&p.in_norm[(size_t)n * p.Cin + ci_base + tid].scale);
      // every wave is past the previous tile's staging phase (it passed that tile's second barrier), nobody reads sN now
      sN[2 * tid] = ss[0]; sN[2 * tid + 1] = ss[1];
    }
    n_cur = n;
    __syncthreads();  // the previous tile's MFMA phase has finished reading LDS; sN is visible
#pragma unroll
    for (int c = 0; c < MC; ++c) {
      float psc[KPL], psh[KPL];
      if (p.in_norm) {
        const f32x4_t* q = reinterpret_cast<const f32x4_t*>(sN + 2 * (c * 16 + (tid & 1) * KPL));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f32x4_t v = q[e];
          psc[2 * e] = v[0]; psh[2 * e] = v[1]; psc[2 * e + 1] = v[2]; psh[2 * e + 1] = v[3];
        }
      }
#pragma unroll
      for (int u = 0; u < NPA; ++u) {
        u32x4_t v = pa[c][u];
        if ((p.in_norm || XF16) && oka[u]) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float a = lo16<TXE>(v[q]), b = hi16<TXE>(v[q]);
            if (p.in_norm) {
              a = fmaf(psc[2 * q], a, psh[2 * q]); b = fmaf(psc[2 * q + 1], b, psh[2 * q + 1]);
              act_pair<ACTK>(a, b, p.act);
            }
            v[q] = cvt_pk_bf16(a, b);
          }
        }
        *reinterpret_cast<u32x4_t*>(sA + (size_t)c * TV * VBA + (size_t)(u * 256 + tid) * 16) = v;
      }
    }
#pragma unroll
    for (int u = 0; u < NPG; ++u)
      if (u < NPG - 1 || last_ok) *reinterpret_cast<u32x4_t*>(sG + (size_t)(u * 256 + tid) * 16) = pg[u];
    __syncthreads();

    switch (wave) {   // wave-uniform
      case 0: sd_mfma_phase<0, MC, HY, HX, VBA, VBG, TV, NKC>(sA, sG, a_base, g_lane, acc, want_b); break;
      case 1: sd_mfma_phase<1, MC, HY, HX, VBA, VBG, TV, NKC>(sA, sG, a_base, g_lane, acc, want_b); break;
      case 2: sd_mfma_phase<2, MC, HY, HX, VBA, VBG, TV, NKC>(sA, sG, a_base, g_lane, acc, want_b); break;
      default: sd_mfma_phase<3, MC, HY, HX, VBA, VBG, TV, NKC>(sA, sG, a_base, g_lane, acc, want_b); break;
    }
  }

  float* pp = p.part + (size_t)grp * 27 * p.Cin * p.Cout;
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    const int tap = 7 * wave + a;
    if (tap >= 27) continue;
#pragma unroll
    for (int c = 0; c < MC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ci_base + c * 16 + 4 * g + r, co = co_base + i;
        pp[((size_t)tap * p.Cin + ci) * p.Cout + co] = acc[a][c][r];
      }
  }
  if (want_b && wave == 3 && g == 0) p.dbpart[(size_t)grp * p.Cout + co_base + i] = acc[6][0][0];   // row 0 of D: one writer per (group, channel)
}


// ---- ConvTranspose3d k=2 s=2 wgrad, bf16: all 8 sub-positions in ONE pass over x and dy ---------------------------------
// dW[ci][co][sub] = sum_v x[v][ci] * dy[2v + sub][co].  The generic kernel above needs one launch per sub-position and
// re-reads x eight times with a stride-2 gather of dy; here a workgroup stages a 2x4x16 tile of x and the matching
// CONTIGUOUS 4x8x32 block of dy (de-interleaved into 8 per-sub planes while writing LDS), wave w owns subs 2w and 2w+1,
// and the x fragment of a K-chunk is shared by all subs.  Partials [group][sub][Cin][Cout] -> wgrad_reduce_kernel.
// MC (round 6): x chunks per workgroup.  With one chunk a 32 -> 32 layer stages the 64 KB dy block of a tile TWICE (once per x chunk) for 16 MFMAs per
// wave each time - the staging, not HBM, is what the level-0 launch of cfg 2 takes (150 us for 600 MB); MC = 2 stages it once for 32.
template <int NS, int SZ, int MC = 1>   // SZ = z extent of the kernel = z stride (1 or 2): 4*SZ sub-positions, sub = (a*2 + b)*2 + c
__global__ void __launch_bounds__(256, NS == 1 ? 4 : 2) wgrad_ct_kernel(const WgradParams p) {
  using T = uint16_t;
  constexpr int TZ = 2, TY = 4, TX = 16, TV = TZ * TY * TX;       // x tile
  constexpr int NSUB = 4 * SZ, SPW = NSUB / 4;                    // sub-positions; per wave
  constexpr int GV = NSUB * TV;                                   // dy voxels of the tile
  constexpr int KPL = 8, VBA = 32, CB = 16 * NS, VBG = CB * 2, PPVG = 2 * NS;
  constexpr int NKC = TV / 32;
  constexpr int NPG = GV * PPVG / 256, BATCH = NPG < 8 ? NPG : 8;
  static_assert(NPG % BATCH == 0 && 256 % PPVG == 0, "staging plan");
  __shared__ __attribute__((aligned(16))) unsigned char smem[MC * TV * VBA + GV * VBG];
  unsigned char* sA = smem;                                       // MC tiles of x, one after the other
  unsigned char* sG = smem + MC * TV * VBA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int D = p.D, H = p.H, W = p.W;
  const int nchunks_ = p.Cin / (16 * MC), groups8 = (p.groups + 7) & ~7;   // (chunk GROUPS of MC chunks)
  const int per_cb = groups8 * nchunks_;
  const int cbi = (int)blockIdx.x / per_cb, rem = (int)blockIdx.x % per_cb;
  const int chunk = (rem % (8 * nchunks_)) / 8;
  const int grp = (rem / (8 * nchunks_)) * 8 + rem % 8;
  const int co_base = cbi * CB;
  if (grp >= p.groups) return;

  f32x4_t acc[MC][SPW][NS];
#pragma unroll
  for (int c = 0; c < MC; ++c)
#pragma unroll
    for (int a = 0; a < SPW; ++a)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) acc[c][a][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[KPL];
#pragma unroll
  for (int e = 0; e < KPL; ++e) bsum[e] = 0.f;
  const bool want_bias = p.db != nullptr && chunk == 0;

  const char* __restrict__ xin = reinterpret_cast<const char*>(p.x);
  const char* __restrict__ gin = reinterpret_cast<const char*>(p.dy);
  // this thread's x piece: voxel t = tid>>1 of the tile, 8 channels
  const int ta = tid >> 1, taz = ta >> 6, tay = (ta >> 4) & 3, tax = ta & 15;
  const uint32_t rel_a = (uint32_t)(((taz * H + tay) * W + tax) * p.x_ld + chunk * MC * p.x_cs + (tid & 1) * KPL) * 2u;   // chunk c of the group: + c * x_cs elements
  const int subG = tid % PPVG, qlane = tid / PPVG;                 // dy piece u: block voxel q = u*(256/PPVG) + qlane
  const int trl = (i >> 2), trc = (i & 3) * 8;
  const int a_base = (g * 8 + trl) * VBA + trc;
  const int g_base = ((SPW * wave) * TV + g * 8 + trl) * VBG + trc;  // first sub of this wave; the next one is TV*VBG further

  for (int tt = grp; tt < p.totalTiles; tt += p.groups) {
    const int n = tt / p.tilesPerSample, tile = tt - n * p.tilesPerSample;
    const int z0 = (tile / (p.tilesX * p.tilesY)) * TZ, y0 = ((tile / p.tilesX) % p.tilesY) * TY, x0 = (tile % p.tilesX) * TX;
    const bool full = z0 + TZ <= D && y0 + TY <= H && x0 + TX <= W;
    const uint32_t base_a = (uint32_t)(((n * D + z0) * H + y0) * W + x0) * (uint32_t)p.x_ld * 2u;
    const uint32_t base_g = (uint32_t)(((n * SZ * D + SZ * z0) * 2 * H + 2 * y0) * 2 * W + 2 * x0) * (uint32_t)p.dy_ld * 2u;

    u32x4_t pa[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
      pa[c] = u32x4_t{0u, 0u, 0u, 0u};
      if (full || (z0 + taz < D && y0 + tay < H && x0 + tax < W)) pa[c] = *reinterpret_cast<const u32x4_t*>(xin + (base_a + rel_a + (uint32_t)(c * p.x_cs) * 2u));
      if (p.x_f16) {   // BPX_MIX16 (uniform): x is the forward pass's fp16 tensor, the MFMA operands are bf16
#pragma unroll
        for (int q = 0; q < 4; ++q) pa[c][q] = cvt_pk_bf16(lo16<f16_t>(pa[c][q]), hi16<f16_t>(pa[c][q]));
      }
    }
    __syncthreads();  // previous tile's MFMA phase is done with LDS
#pragma unroll
    for (int c = 0; c < MC; ++c) *reinterpret_cast<u32x4_t*>(sA + (size_t)c * TV * VBA + (size_t)tid * 16) = pa[c];
#pragma unroll
    for (int b0 = 0; b0 < NPG; b0 += BATCH) {
      u32x4_t pg[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int q = (b0 + u) * (256 / PPVG) + qlane;             // voxel of the (SZ*TZ)x8x32 dy block, x fastest
        const int X = q & 31, Y = (q >> 5) & 7, Z = q >> 8;
        pg[u] = u32x4_t{0u, 0u, 0u, 0u};
        if (full || (SZ * z0 + Z < SZ * D && 2 * y0 + Y < 2 * H && 2 * x0 + X < 2 * W))
          pg[u] = *reinterpret_cast<const u32x4_t*>(gin + (base_g + (uint32_t)(((Z * 2 * H + Y) * 2 * W + X) * p.dy_ld + co_base + subG * KPL) * 2u));
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int q = (b0 + u) * (256 / PPVG) + qlane;
        const int X = q & 31, Y = (q >> 5) & 7, Z = q >> 8;
        const int sub = ((SZ == 2 ? (Z & 1) : 0) << 2) | ((Y & 1) << 1) | (X & 1), v = (((SZ == 2 ? (Z >> 1) : Z) * TY + (Y >> 1)) * TX) + (X >> 1);
        *reinterpret_cast<u32x4_t*>(sG + (size_t)((sub * TV + v) * PPVG + subG) * 16) = pg[u];
        if (want_bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { bsum[2 * e] += bf16lo(pg[u][e]); bsum[2 * e + 1] += bf16hi(pg[u][e]); }
        }
      }
    }
    __syncthreads();

#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      u32x4_t af[MC], gf[SPW][NS];
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        const unsigned char* q = sA + c * TV * VBA + a_base + kc * 32 * VBA;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(q));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(q + 4 * VBA));
        u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
        af[c] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
      }
#pragma unroll
      for (int a = 0; a < SPW; ++a)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          const unsigned char* q = sG + g_base + (a * TV + kc * 32) * VBG + ns * 32;
          s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(q));
          s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(q + 4 * VBG));
          u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
          gf[a][ns] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < MC; ++c)
#pragma unroll
        for (int a = 0; a < SPW; ++a)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
            acc[c][a][ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[c]), __builtin_bit_cast(bf16x8_t, gf[a][ns]), acc[c][a][ns], 0, 0, 0);
    }
  }

  float* pp = p.part + (size_t)grp * NSUB * p.Cin * p.Cout;
#pragma unroll
  for (int c = 0; c < MC; ++c)
#pragma unroll
    for (int a = 0; a < SPW; ++a)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ci = (chunk * MC + c) * 16 + 4 * g + r, co = co_base + ns * 16 + i;
          pp[((size_t)(SPW * wave + a) * p.Cin + ci) * p.Cout + co] = acc[c][a][ns][r];
        }
  if (want_bias) {  // every thread summed the 8 channels of its pieces: combine the 256/PPVG threads of a channel group
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [256][8]
#pragma unroll
    for (int e = 0; e < KPL; ++e) red[tid * KPL + e] = bsum[e];
    __syncthreads();
    if (tid < CB) {
      const int sg = tid / KPL, e = tid % KPL;
      float sum = 0.f;
      for (int k = sg; k < 256; k += PPVG) sum += red[k * KPL + e];
      p.dbpart[(size_t)grp * p.Cout + co_base + tid] = sum;
    }
  }
}

int g_ct_resident = 1;   // test hook bit 6 clears it: the former 2048-workgroup target
struct CtCfg { int ns, groups, totalTiles, tilesY, tilesX, tilesPerSample; };
// x chunks per workgroup of the tile kernel: two for the big even-chunk layers (cfg 2: 32 -> 32 @64^3 -> 128^3), where the dy staging binds; the small
// levels keep one (they need every workgroup they can get).  BPX_CT_MC=1: the one-chunk kernel everywhere (A/B).
inline int ct_mc(int N, int D, int H, int W, int sz, int Cin, int Cout) {
  static const int env = getenv("BPX_CT_MC") ? atoi(getenv("BPX_CT_MC")) : 2;
  return (env >= 2 && sz == 2 && (Cin / 16) % 2 == 0 && Cout % 32 == 0 && (int64_t)N * D * H * W >= 262144) ? 2 : 1;
}
inline CtCfg pick_ct(int N, int D, int H, int W, int sz, int Cin, int Cout) {
  CtCfg c;
  c.ns = (Cout % 32 == 0) ? 2 : 1;
  c.tilesY = cdiv(H, 4); c.tilesX = cdiv(W, 16);
  c.tilesPerSample = cdiv(D, 2) * c.tilesY * c.tilesX;
  c.totalTiles = N * c.tilesPerSample;
  const int nchunks = Cin / (16 * ct_mc(N, D, H, W, sz, Cin, Cout)), nb = Cout / (16 * c.ns);   // workgroups per group
  const int64_t cap = std::max<int64_t>(1, (int64_t)6400000 / ((int64_t)4 * sz * Cin * Cout));
  // one resident wave of workgroups (256 CUs x occupancy of wgrad_ct_kernel), each looping over its share of the tiles:
  // measured best for the shift-dy kernels, same structure here
  const int occ = c.ns == 1 ? 4 : 2;
  const int64_t want = g_ct_resident ? (cdiv64((int64_t)256 * occ, nchunks * nb) + 7) & ~7ll : std::max(1, cdiv(2048, nchunks * nb));
  c.groups = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(c.totalTiles, cap), want));
  return c;
}

// sums the per-group partials in a fixed order and writes dW in its final layout.
// 1024 threads = 32 consecutive elements x 32 group lanes, four loads in flight per thread: the smallest dW has only 6912
// elements (216 blocks) against up to 1024 partial slabs, so the kernel is a latency chain per block - with 8 group lanes
// and two loads in flight it took ~70 us for the 16->16 layers, more than a quarter of their MFMA kernel.
struct ReduceJob { const float* part; float* dw; int groups, taps, Cin, Cout; int64_t si, sj, st, off; const float* dbpart; float* db; int ndb; float* db2 = nullptr; int64_t db_stride = 0; };   // ndb: length of a bias row (0 = Cout); db2: a second tensor that takes the same sums; db_stride: floats between the bias rows of consecutive groups (0 = ndb)

// 1024 threads = EL consecutive elements x GL group lanes (GL = reduce_glanes(groups), EL = 1024 / GL), up to four loads in
// flight per thread; lane sums are combined in a fixed order, so the result does not depend on scheduling.
__host__ __device__ inline int reduce_glanes(int groups) {
  int gl = 1;
  while (gl < 32 && gl * 4 < groups) gl <<= 1;   // ~4 slabs per thread and more for the many-group layers
  return gl;
}
// elements of a job: the dW entries, then (with a bias gradient) the Cout column sums.  A thread owns FOUR consecutive elements
// (16-byte loads of the partial slabs: the reduction reads ~0.7 GB per cfg-2 step and was at 2.3 TB/s with 4-byte loads).
inline int reduce_blocks(const ReduceJob& j) {
  return (int)cdiv64((int64_t)j.taps * j.Cin * j.Cout + (j.db ? (j.ndb ? j.ndb : j.Cout) : 0), 4 * (1024 / reduce_glanes(j.groups)));
}

__device__ __forceinline__ void wgrad_reduce_block(const ReduceJob& j, int block) {
  __shared__ f32x4_t red[1024];
  const float* __restrict__ part = j.part;
  const int groups = j.groups;
  const int GL = reduce_glanes(groups), EL = 1024 / GL;
  const int ndb = j.ndb ? j.ndb : j.Cout;
  const int64_t total = (int64_t)j.taps * j.Cin * j.Cout, all = total + (j.db ? ndb : 0);   // total is a multiple of 16
  const int e = threadIdx.x % EL, gl = threadIdx.x / EL;
  const int64_t idx = ((int64_t)block * EL + e) * 4;
  // elements idx .. idx+3 of slab g: dW entries, or (idx >= total) column sums of the bias-gradient partials [groups][ndb]
  const float* __restrict__ src = idx < total ? part + idx : j.dbpart + (idx - total);
  const int64_t stride = idx < total ? total : (j.db_stride ? j.db_stride : ndb);
  const int nval = idx >= all ? 0 : (int)(all - idx < 4 ? all - idx : 4);
  const bool vec = nval == 4 && (stride & 3) == 0 && (((uintptr_t)src) & 15) == 0;
  f32x4_t s0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (vec) {
    int gq = gl;
    for (; gq + 3 * GL < groups; gq += 4 * GL) {
      s0 += *reinterpret_cast<const f32x4_t*>(src + (size_t)gq * stride);
      s1 += *reinterpret_cast<const f32x4_t*>(src + (size_t)(gq + GL) * stride);
      s2 += *reinterpret_cast<const f32x4_t*>(src + (size_t)(gq + 2 * GL) * stride);
      s3 += *reinterpret_cast<const f32x4_t*>(src + (size_t)(gq + 3 * GL) * stride);
    }
    for (; gq < groups; gq += GL) s0 += *reinterpret_cast<const f32x4_t*>(src + (size_t)gq * stride);
  } else if (nval > 0) {   // ragged tail / unaligned bias rows (head gradients): element by element, same order of additions
    // (four rows requested before the first addition: as one load per addition the head's bias column was a chain of 32 dependent L2 round
    //  trips - 21 us for a 17-element gradient; the additions keep their order)
    int gq = gl;
    for (; gq + 3 * GL < groups; gq += 4 * GL) {
      float a[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[u][k] = k < nval ? src[(size_t)(gq + u * GL) * stride + k] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < nval) s0[k] += a[u][k];
    }
    for (; gq < groups; gq += GL)
      for (int k = 0; k < nval; ++k) s0[k] += src[(size_t)gq * stride + k];
  }
  red[gl * EL + e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (gl == 0 && nval > 0) {
    f32x4_t s = red[e];
    for (int q = 1; q < GL; ++q) s += red[q * EL + e];
    for (int k = 0; k < nval; ++k) {
      const int64_t i = idx + k;
      if (i < total) {
        int co = (int)(i % j.Cout), ci = (int)((i / j.Cout) % j.Cin), tap = (int)(i / ((int64_t)j.Cout * j.Cin));
        j.dw[ci * j.si + co * j.sj + tap * j.st + j.off] = s[k];
      } else {
        j.db[i - total] += s[k];   // accumulated like the former atomics (the caller zeroes db), single writer, fixed order
        if (j.db2) j.db2[i - total] += s[k];
      }
    }
  }
}

__global__ void __launch_bounds__(1024) wgrad_reduce_kernel(const ReduceJob j) { wgrad_reduce_block(j, (int)blockIdx.x); }

// deferred form: up to 32 pending reductions in ONE launch (bpx_wgrad_defer_begin / _flush).  A layer's reduction is a
// latency chain of a few hundred blocks; 29 of them back to back cost 0.6 ms per step, together they overlap.
constexpr int RB_MAX = 32;
struct ReduceBatch { ReduceJob job[RB_MAX]; int first_block[RB_MAX + 1]; int count; };
__global__ void __launch_bounds__(1024) wgrad_reduce_batch_kernel(const ReduceBatch b) {
  int lo = 0, hi = b.count;                          // job k owns blocks [first_block[k], first_block[k+1]); block-uniform search
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= b.first_block[mid]) lo = mid; else hi = mid;
  }
  wgrad_reduce_block(b.job[lo], (int)blockIdx.x - b.first_block[lo]);
}

struct DeferState { bool active = false; std::vector<ReduceJob> jobs; };
thread_local DeferState t_defer;

// reduce now, or queue the reduction while the deferred mode is on
int finish_wgrad(const char* fn, const ReduceJob& j, hipStream_t s) {
  if (t_defer.active) { t_defer.jobs.push_back(j); return 0; }
  wgrad_reduce_kernel<<<reduce_blocks(j), 1024, 0, s>>>(j);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

// ---- k = 1 weight gradient of a RAW input (the residual blocks' 1x1x1 shortcut: blocks.py ResConvBlock `shortcut(x)`) as a stream -----------------
//   dW[ci][co] = sum_v x[v][ci] * dy[v][co]
// has no halo, no prologue and 0.4 FLOP per byte: it is a reduction over two streams.  The generic kernel above walks it in 256-voxel tiles with a
// barrier pair per tile and register-staged loads (3.7 TB/s for x 48 . dy 16 at 128^3, 0.29 ms of the cfg-2 step).  Here: persistent workgroups, the
// two operands of a TV-voxel block brought in by `buffer_load ... lds` through a ring of RING stages (RING - 1 blocks in flight per workgroup,
// ~96 KB per CU), one barrier per block; wave w multiplies the block's 32-voxel K chunks w, w + 4, ..; the four waves' accumulators meet once, at
// the end.  x may be fp16 (BPX_MIX16: converted to bf16 after the transposing LDS read) and chunk-planar.
struct K1Params {
  const void* x; int x_ld; int x_cs; const void* dy; int dy_ld;
  int Cin, Cout; int64_t voxels; int nblocks; int groups; float* part;
};
__host__ __device__ constexpr int k1_ring(int stage_bytes) { return 4 * stage_bytes <= 131072 ? 4 : 3 * stage_bytes <= 131072 ? 3 : 2; }
__host__ __device__ constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }   // s_waitcnt vmcnt(n), the other counters open

template <int MC, int NS, int TV, bool XF16>
__global__ void __launch_bounds__(256) wgrad_k1_dma_kernel(const K1Params p) {
  constexpr int VB = 32, SUBS = TV / 32;                          // a DMA instruction moves 32 voxels x 32 bytes (lane: voxel l >> 1, half l & 1)
  constexpr int A_BYTES = MC * TV * VB, STAGE = (MC + NS) * TV * VB, RING = k1_ring(STAGE);
  constexpr int NI = (MC + NS) * SUBS;                            // instructions per stage, instruction q belongs to wave q & 3
  static_assert(NI % 4 == 0, "every wave issues the same number of DMA instructions per stage");
  constexpr int IPW = NI / 4;
  static_assert((RING - 2) * IPW < 64, "vmcnt range");
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING * STAGE];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int grp = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, (int)0x80000000u, 0x00020000);
  const uint32_t xvb = (uint32_t)p.x_ld * 2u, gvb = (uint32_t)p.dy_ld * 2u, xcb = (uint32_t)p.x_cs * 2u;
  const uint32_t lane_v = (uint32_t)(lane >> 1), lane_h = (uint32_t)(lane & 1) * 16u;
  const int64_t voxels = p.voxels;
  const int groups = p.groups, nblocks = p.nblocks;

  auto issue = [=](int blk, int slot) {
    const int64_t v0 = (int64_t)blk * TV;
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
      const int q = wave + 4 * k;                                 // (operand chunk, 32-voxel run) of this instruction
      const int ch = q / SUBS, sub = q % SUBS;
      const int64_t v = v0 + sub * 32 + lane_v;
      const bool in = v < voxels;
      const uint32_t vv = (uint32_t)v;
      unsigned char* dst = const_cast<unsigned char*>(smem) + slot * STAGE + ch * TV * VB + sub * 1024;
      if (ch < MC) {
        const uint32_t off = in ? vv * xvb + (uint32_t)ch * xcb + lane_h : 0x80000000u;      // out of range: the DMA writes zeros
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)dst, 16, off, 0, 0, 0);
      } else {
        const uint32_t off = in ? vv * gvb + (uint32_t)(ch - MC) * 32u + lane_h : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)dst, 16, off, 0, 0, 0);
      }
    }
  };

  f32x4_t acc[MC][NS];
#pragma unroll
  for (int c = 0; c < MC; ++c)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) acc[c][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int a_base = (g * 8 + (i >> 2)) * VB + (i & 3) * 8;      // lane (i, g): voxels 8g .. 8g + 7 of a K chunk through ds_read_b64_tr_b16

  const int nst = grp < nblocks ? (nblocks - grp + groups - 1) / groups : 0;    // blocks grp, grp + groups, ..
#pragma unroll
  for (int s = 0; s < RING - 1; ++s)
    if (s < nst) issue(grp + s * groups, s);
  for (int s = 0; s < nst; ++s) {
    // this wave's share of block s has landed when at most the shares of the later blocks in flight are outstanding (VMEM returns in order)
    const int later = nst - 1 - s < RING - 2 ? nst - 1 - s : RING - 2;
    if (later >= 2 && RING >= 4) __builtin_amdgcn_s_waitcnt(vmcnt_imm(2 * IPW));
    else if (later == 1 && RING >= 3) __builtin_amdgcn_s_waitcnt(vmcnt_imm(IPW));
    else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    __syncthreads();                                              // every wave's share is in LDS; slot (s - 1) % RING is no longer read
    if (s + RING - 1 < nst) issue(grp + (s + RING - 1) * groups, (s + RING - 1) % RING);
    const unsigned char* st = smem + (s % RING) * STAGE;
#pragma unroll
    for (int kc0 = 0; kc0 < SUBS; kc0 += 4) {
      const int kc = kc0 + wave;
      if (SUBS % 4 != 0 && kc >= SUBS) break;
      u32x4_t gf[NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const unsigned char* q = st + A_BYTES + ns * TV * VB + a_base + kc * 32 * VB;
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q)));
        const u32x2_t h2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 4 * VB)));
        gf[ns] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
      }
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        const unsigned char* q = st + c * TV * VB + a_base + kc * 32 * VB;
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q)));
        const u32x2_t h2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 4 * VB)));
        u32x4_t af = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
        if (XF16) {
#pragma unroll
          for (int e = 0; e < 4; ++e) af[e] = cvt_pk_bf16(lo16<f16_t>(af[e]), hi16<f16_t>(af[e]));
        }
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          acc[c][ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, gf[ns]), acc[c][ns], 0, 0, 0);
      }
    }
  }
  // the four waves' sums (wave order: fixed) -> this workgroup's slab [Cin][Cout]
  __syncthreads();
  f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);                // [wave][MC][NS][64 lanes]
  static_assert(4 * MC * NS * 64 * 16 <= RING * STAGE, "reduction scratch");
#pragma unroll
  for (int c = 0; c < MC; ++c)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) red[((wave * MC + c) * NS + ns) * 64 + lane] = acc[c][ns];
  __syncthreads();
  float* pp = p.part + (size_t)grp * p.Cin * p.Cout;
  for (int q = tid; q < MC * NS * 64; q += 256) {
    const int ln = q & 63, cn = q >> 6, c = cn / NS, ns = cn % NS;
    const f32x4_t a = (red[((0 * MC + c) * NS + ns) * 64 + ln] + red[((1 * MC + c) * NS + ns) * 64 + ln]) +
                      (red[((2 * MC + c) * NS + ns) * 64 + ln] + red[((3 * MC + c) * NS + ns) * 64 + ln]);
    const int gi = ln >> 4, ii = ln & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) pp[(size_t)(c * 16 + 4 * gi + r) * p.Cout + ns * 16 + ii] = a[r];
  }
}

int g_k1_dma = 1;     // bpx_debug_set_wgrad_k1: 0 = the generic kernel everywhere
// instance for (Cin / 16, Cout / 16), or 0: the cfg-2 shortcuts at the 128^3 / 64^3 levels (48 . 16, 96 . 32, 16 . 32)
inline int k1_instance(int mc, int ns) { return (mc == 3 && ns == 1) ? 1 : (mc == 6 && ns == 2) ? 2 : (mc == 1 && ns == 2) ? 3 : 0; }
inline int k1_tv(int inst) { return inst == 2 ? 128 : 256; }
inline bool k1_dma_ok(int dtype, const WgradParams& p, int taps) {
  if (!g_k1_dma || dtype != BPX_BF16 || taps != 1 || p.in_norm != nullptr || p.db != nullptr || p.dy_vs != 1 || p.dy_oz || p.dy_oy || p.dy_ox) return false;
  if (p.Cin % 16 || p.Cout % 16 || !k1_instance(p.Cin / 16, p.Cout / 16)) return false;
  const int64_t vox = (int64_t)p.N * p.D * p.H * p.W;
  if (vox < 65536) return false;                                 // small levels: nothing to stream
  const int64_t xspan = (int64_t)(p.Cin / 16 - 1) * p.x_cs * 2 + vox * p.x_ld * 2, gspan = vox * (int64_t)p.dy_ld * 2;
  return xspan < (1ll << 31) && gspan < (1ll << 31) && (((uintptr_t)p.x | (uintptr_t)p.dy) & 15) == 0 && (p.x_ld & 7) == 0 && (p.dy_ld & 7) == 0 && (p.x_cs & 7) == 0;
}
inline int k1_groups(int64_t vox, int inst) { return (int)std::min<int64_t>(256, cdiv64(vox, k1_tv(inst))); }   // one persistent workgroup per CU (the ring holds ~128 KB)

int launch_wgrad_k1_dma(const WgradParams& w, int groups, hipStream_t s) {
  const int inst = k1_instance(w.Cin / 16, w.Cout / 16);
  K1Params p{};
  p.x = w.x; p.x_ld = w.x_ld; p.x_cs = w.x_cs; p.dy = w.dy; p.dy_ld = w.dy_ld; p.Cin = w.Cin; p.Cout = w.Cout;
  p.voxels = (int64_t)w.N * w.D * w.H * w.W;
  p.nblocks = (int)cdiv64(p.voxels, k1_tv(inst));
  p.groups = groups; p.part = w.part;
#define K1(I, MC_, NS_, TV_)                                                                                      \
  if (inst == I) {                                                                                                \
    if (w.x_f16) wgrad_k1_dma_kernel<MC_, NS_, TV_, true><<<groups, 256, 0, s>>>(p);                              \
    else wgrad_k1_dma_kernel<MC_, NS_, TV_, false><<<groups, 256, 0, s>>>(p);                                     \
    return 0;                                                                                                     \
  }
  K1(1, 3, 1, 256) K1(2, 6, 2, 128) K1(3, 1, 2, 256)
#undef K1
  return 1;
}

// ---- transposed-conv (k = s = 2) weight gradient as a stream (round 4) ----------------------------------------------------------------------------
//   dW[sub][ci][co] = sum_v x[v][ci] * dy[2v + sub][co],  db[co] = sum over every dy voxel
// The tile kernel above (wgrad_ct_kernel) gives a workgroup one (16-channel x chunk, co block): dy is staged once per x chunk, through registers, with
// a barrier pair per 256-voxel tile - 0.17 ms for 32 -> 32 @64^3 -> 128^3 (2.25 tensor units: 3.6 TB/s).  Same recipe as wgrad_k1_dma_kernel: persistent
// workgroups, the x chunks and the 8 sub-position gathers of dy of a TV-voxel block through an LDS-DMA ring; a workgroup owns ALL channels (x and dy
// are read once), wave w owns sub-positions 2w, 2w + 1 - no cross-wave sum for dW.
struct CtsParams {
  const void* x; int x_ld; const void* dy; int dy_ld;
  int Cin, Cout, D, H, W; int64_t voxels; int nblocks; int groups; float* part; float* dbpart;
};

template <int MC, int NS, int TV, bool XF16>
__global__ void __launch_bounds__(256) wgrad_ct_dma_kernel(const CtsParams p) {
  constexpr int VB = 32, SUBS = TV / 32, NCH = MC + 8 * NS, A_BYTES = MC * TV * VB, STAGE = NCH * TV * VB, RING = 3;
  constexpr int NI = NCH * SUBS;
  static_assert(NI % 4 == 0, "every wave issues the same number of DMA instructions per stage");
  constexpr int IPW = NI / 4;
  static_assert(RING * STAGE <= 131072 && IPW < 64, "ring size / vmcnt range");
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING * STAGE];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int grp = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, (int)0x80000000u, 0x00020000);
  const uint32_t xvb = (uint32_t)p.x_ld * 2u, gvb = (uint32_t)p.dy_ld * 2u;
  const uint32_t lane_v = (uint32_t)(lane >> 1), lane_h = (uint32_t)(lane & 1) * 16u;
  const int64_t voxels = p.voxels;
  const int groups = p.groups, nblocks = p.nblocks;
  const uint32_t W = (uint32_t)p.W, H = (uint32_t)p.H;

  auto issue = [=](int blk, int slot) {
    const int64_t v0 = (int64_t)blk * TV;
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
      const int q = wave + 4 * k;
      const int ch = q / SUBS, run = q % SUBS;
      const int64_t v = v0 + run * 32 + lane_v;
      const bool in = v < voxels;
      const uint32_t vv = (uint32_t)v;
      unsigned char* dst = const_cast<unsigned char*>(smem) + slot * STAGE + ch * TV * VB + run * 1024;
      if (ch < MC) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)dst, 16, in ? vv * xvb + (uint32_t)ch * 32u + lane_h : 0x80000000u, 0, 0, 0);
      } else {
        const int sub = (ch - MC) / NS, ns = (ch - MC) % NS;
        const uint32_t row = vv / W, xx = vv - row * W, zz = row / H, yy = row - zz * H;      // zz = n * D + z
        const uint32_t hi = ((zz * 2u + (uint32_t)((sub >> 2) & 1)) * (2u * H) + 2u * yy + (uint32_t)((sub >> 1) & 1)) * (2u * W) + 2u * xx + (uint32_t)(sub & 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)dst, 16, in ? hi * gvb + (uint32_t)ns * 32u + lane_h : 0x80000000u, 0, 0, 0);
      }
    }
  };

  f32x4_t acc[2][MC][NS], accb[NS];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < MC; ++c)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) acc[a][c][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) accb[ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int a_base = (g * 8 + (i >> 2)) * VB + (i & 3) * 8;
  const u32x4_t ones = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const bool want_b = p.dbpart != nullptr;

  const int nst = grp < nblocks ? (nblocks - grp + groups - 1) / groups : 0;
#pragma unroll
  for (int s = 0; s < RING - 1; ++s)
    if (s < nst) issue(grp + s * groups, s);
  for (int s = 0; s < nst; ++s) {
    if (nst - 1 - s >= 1) __builtin_amdgcn_s_waitcnt(vmcnt_imm(IPW));     // block s + 1 may stay in flight
    else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    __syncthreads();
    if (s + RING - 1 < nst) issue(grp + (s + RING - 1) * groups, (s + RING - 1) % RING);
    const unsigned char* st = smem + (s % RING) * STAGE;
#pragma unroll
    for (int kc = 0; kc < SUBS; ++kc) {
      u32x4_t af[MC];
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        const unsigned char* q = st + c * TV * VB + a_base + kc * 32 * VB;
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q)));
        const u32x2_t h2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 4 * VB)));
        af[c] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
        if (XF16) {
#pragma unroll
          for (int e = 0; e < 4; ++e) af[c][e] = cvt_pk_bf16(lo16<f16_t>(af[c][e]), hi16<f16_t>(af[c][e]));
        }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int sub = 2 * wave + a;
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          const unsigned char* q = st + A_BYTES + (sub * NS + ns) * TV * VB + a_base + kc * 32 * VB;
          const u32x2_t l2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q)));
          const u32x2_t h2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 4 * VB)));
          const u32x4_t gf = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
#pragma unroll
          for (int c = 0; c < MC; ++c)
            acc[a][c][ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[c]), __builtin_bit_cast(bf16x8_t, gf), acc[a][c][ns], 0, 0, 0);
          if (want_b) accb[ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ones), __builtin_bit_cast(bf16x8_t, gf), accb[ns], 0, 0, 0);
        }
      }
    }
  }
  // partial slab [grp][sub][Cin][Cout]: this wave's two sub-positions
  float* pp = p.part + (size_t)grp * 8 * p.Cin * p.Cout;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < MC; ++c)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[((size_t)(2 * wave + a) * p.Cin + c * 16 + 4 * g + r) * p.Cout + ns * 16 + i] = acc[a][c][ns][r];
  if (want_b) {   // every row of accb holds the column sums of this wave's sub-positions: row 0, waves in order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);      // [wave][NS * 16]
    if (g == 0) {
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) red[wave * NS * 16 + ns * 16 + i] = accb[ns][0];
    }
    __syncthreads();
    if (tid < NS * 16) p.dbpart[(size_t)grp * p.Cout + tid] = (red[tid] + red[NS * 16 + tid]) + (red[2 * NS * 16 + tid] + red[3 * NS * 16 + tid]);
  }
}

int g_ct_dma = 1;     // bpx_debug_set_wgrad_k1: 1 = both streaming kernels, 0 = neither, 3 = k = 1 only, 5 = transposed conv only
// (2, 2) = 32 -> 32 @64^3 measured SLOWER than the tile kernel (223 vs 171 us: 18 chunk gathers of 64 B-strided half lines per 36 KB stage, two stages
// in flight) and is only selectable through the hook value 7; (4, 4) = 64 -> 64 @32^3: 93 -> 59 us
int g_ct_dma_all = 0;
inline int cts_instance(int mc, int ns) { return (mc == 2 && ns == 2) ? (g_ct_dma_all ? 1 : 0) : (mc == 4 && ns == 4) ? 2 : 0; }
inline int cts_tv(int inst) { return inst == 1 ? 64 : 32; }

int g_cap_pct = 100;   // partial-slab caps of the two plans below in percent (bpx_debug_set_wgrad_cap)
int g_k1_wgs = 768;    // workgroups targeted by the k = 1 launches: 768 (three per CU) measured best of 256..2048 (test hook: bit 7 + percent of 1024 in bits 8..)
struct WCfg { int tz, ty, tx, ns, groups; };
// Deterministic in its arguments: the workspace query and the launch must agree.
inline WCfg pick_wcfg(int N, int D, int H, int W, int Cin, int Cout, int taps, bool big_ok) {
  WCfg c;
  c.tz = 4; c.ty = 4;
  c.tx = (W > 8) ? 16 : 8;
  int ns = (Cout % 64 == 0) ? 4 : (Cout % 32 == 0) ? 2 : 1;
  // k = 1 (shortcut, transposed conv): a 256-voxel tile is only 8 K-chunks - two per wave between barriers.  Large
  // volumes take 8x8x16 tiles (8 K-chunks per wave); the dy tile then limits the block to 32 output channels.
  if (big_ok && taps == 1 && W > 8 && D >= 8 && H >= 8) { c.tz = 8; c.ty = 8; if (ns > 2) ns = 2; }
  const int totalTiles = N * cdiv(D, c.tz) * cdiv(H, c.ty) * cdiv(W, c.tx);
  const int64_t dwElems = (int64_t)taps * Cin * Cout;
  const int nchunks = Cin / 16;
  for (;;) {
    int nb = Cout / (16 * ns);
    int64_t cap = std::max<int64_t>(1, (int64_t)64000 * g_cap_pct / dwElems);  // keep the partial slab <= ~50 MB round trip
    int groups = (int)std::min<int64_t>(std::min<int64_t>(totalTiles, cap), std::max(1, cdiv(taps == 1 ? g_k1_wgs : 2048, nchunks * nb)));
    c.ns = ns; c.groups = groups;
    const int minblk = ((int64_t)D * H * W <= 512) ? 128 : 512;   // 8^3 bottleneck: wider co blocks win (82 -> 61 us); elsewhere NS = 1
    if (ns == 1 || (int64_t)groups * nchunks * nb >= minblk) break;
    ns >>= 1;                                                                  // not enough workgroups: split the co blocks finer
  }
  return c;
}

template <typename T, int TAPS>
int launch_wgrad(const WgradParams& p0, const WCfg& c, bool use_tr, hipStream_t s) {
  WgradParams p = p0;
  int tilesZ = cdiv(p.D, c.tz);
  p.tilesY = cdiv(p.H, c.ty);
  p.tilesX = cdiv(p.W, c.tx);
  p.tilesPerSample = tilesZ * p.tilesY * p.tilesX;
  p.totalTiles = p.N * p.tilesPerSample;
  int nchunks = p.Cin / 16, nb = p.Cout / (16 * c.ns);
  int groups = c.groups;
  p.groups = groups;
  p.dbpart = p.part + (size_t)groups * TAPS * p.Cin * p.Cout;
  const bool elu = p.in_norm != nullptr && p.act == BPX_ACT_ELU;
  const bool ext = p.in_norm != nullptr && p.act > BPX_ACT_SILU;   // leaky_relu ... softplus: the ACTK = 2 instances
  dim3 grid((unsigned)(((groups + 7) & ~7) * nchunks * nb));
#define L(TZY, TX, NS)                                                                              \
  if (c.tz == TZY && c.tx == TX && c.ns == NS) {                                                    \
    if (use_tr && elu) wgrad_kernel<T, TZY, TZY, TX, NS, TAPS, true, 1><<<grid, 256, 0, s>>>(p);     \
    else if (use_tr && ext) wgrad_kernel<T, TZY, TZY, TX, NS, TAPS, true, 2><<<grid, 256, 0, s>>>(p); \
    else if (use_tr) wgrad_kernel<T, TZY, TZY, TX, NS, TAPS, true, 0><<<grid, 256, 0, s>>>(p);       \
    else if (elu) wgrad_kernel<T, TZY, TZY, TX, NS, TAPS, false, 1><<<grid, 256, 0, s>>>(p);         \
    else if (ext) wgrad_kernel<T, TZY, TZY, TX, NS, TAPS, false, 2><<<grid, 256, 0, s>>>(p);         \
    else wgrad_kernel<T, TZY, TZY, TX, NS, TAPS, false, 0><<<grid, 256, 0, s>>>(p);                  \
    return 0;                                                                                       \
  }
  L(4, 16, 1) L(4, 16, 2) L(4, 16, 4) L(4, 8, 1) L(4, 8, 2) L(4, 8, 4)
  if constexpr (TAPS == 1 && sizeof(T) == 2) { L(8, 16, 1) L(8, 16, 2) }
#undef L
  return 1;
}

int g_use_tr = 1;
int g_sd_mc = -1;     // input-channel chunks per workgroup of the windowed shift-dy kernel: -1 automatic, 1 / 2 / 3 forced, 0 = the plain
                      // shift-dy kernel (hook bits 3..5)
int g_wgrad_sd = -1;  // -1 automatic (wherever it applies), 0 never, 1 always (bpx_debug_set_wgrad_tr bits 1/2)

int g_sd_order = 1;    // windowed kernel: XCD-contiguous y-strip tile walk (bpx_debug_set_tile_order)
int g_sd_fill = 100;  // windowed kernel: workgroups launched, in percent of the co-resident capacity (256 CUs x occupancy)

// Plan of the windowed shift-dy kernel: chunks per workgroup and tile groups.  Every workgroup is resident at once (one
// "wave" of workgroups, each looping over its share of the tiles), so the partial slab is groups x 27 x Cin x Cout floats.
struct SdmPlan { int mc, groups; };
inline SdmPlan sdm_plan(int N, int D, int H, int W, int Cin, int Cout) {
  const int nchunks = Cin / 16, nb = Cout / 16;
  SdmPlan q;
  q.mc = g_sd_mc > 0 ? g_sd_mc : (nchunks % 2 == 0 ? 2 : nchunks % 3 == 0 ? 3 : 1);   // 4 chunks (254 VGPRs) measured no better than 2
  if (g_sd_mc == 0 || nchunks % q.mc != 0) { q.mc = 0; q.groups = 0; return q; }
  const int occ = q.mc == 1 ? 4 : q.mc == 2 ? 3 : 2;
  const int units = (nchunks / q.mc) * nb;
  const int64_t totalTiles = (int64_t)N * cdiv(D, 4) * cdiv(H, 4) * cdiv(W, 16);
  int64_t g = cdiv64((int64_t)256 * occ * g_sd_fill / 100, units);
  g = (g + 7) & ~7ll;                                                      // the grid rounds groups up to a multiple of 8 anyway
  const int64_t cap = std::max<int64_t>(8, ((int64_t)160000 * g_cap_pct / ((int64_t)27 * Cin * Cout)) & ~7ll);   // partial slab <= 64 MB
  q.groups = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(g, cap), totalTiles));
  return q;
}

// shift-dy kernel: bf16, 3x3x3, W > 8; 32-bit byte offsets -> every tensor < 4 GB
int launch_wgrad_sd(const WgradParams& p0, WCfg& c, hipStream_t s) {
  WgradParams p = p0;
  p.tilesY = cdiv(p.H, 4);
  p.tilesX = cdiv(p.W, 16);
  p.tilesPerSample = cdiv(p.D, 4) * p.tilesY * p.tilesX;
  p.totalTiles = p.N * p.tilesPerSample;
  const int ns = std::min(c.ns, 2);
  const int nchunks = p.Cin / 16, nb = p.Cout / (16 * ns);
  const bool elu = p.in_norm != nullptr && p.act == BPX_ACT_ELU;
  const SdmPlan q = sdm_plan(p.N, p.D, p.H, p.W, p.Cin, p.Cout);
  const int mc = q.mc;
  if (ns == 1 && mc > 0) {
    c.groups = q.groups;                                   // the reduce that follows reads this many partial slabs
    p.groups = q.groups;
    p.tilesZ = cdiv(p.D, 4); p.stripY = bpxconv::strip_rows(p.tilesX); p.order = g_sd_order;
    p.dbpart = p.part + (size_t)p.groups * 27 * p.Cin * p.Cout;
    dim3 gridm((unsigned)(((c.groups + 7) & ~7) * (nchunks / mc) * nb));
#define SDM(MC_)                                                                                   \
    if (mc == MC_) {                                                                               \
      if (p.x_f16) { if (elu) wgrad_sdm_kernel<MC_, 1, true><<<gridm, 256, 0, s>>>(p); else wgrad_sdm_kernel<MC_, 0, true><<<gridm, 256, 0, s>>>(p); } \
      else { if (elu) wgrad_sdm_kernel<MC_, 1><<<gridm, 256, 0, s>>>(p); else wgrad_sdm_kernel<MC_, 0><<<gridm, 256, 0, s>>>(p); }                     \
      return 0;                                                                                    \
    }
    SDM(1) SDM(2) SDM(3)
#undef SDM
    return 1;
  }
  p.groups = c.groups;
  p.dbpart = p.part + (size_t)p.groups * 27 * p.Cin * p.Cout;
  dim3 grid((unsigned)(((c.groups + 7) & ~7) * nchunks * nb));
#define SD(NS_)                                                                                    \
  if (ns == NS_) {                                                                                 \
    if (p.x_f16) { if (elu) wgrad_sd_kernel<NS_, 1, true><<<grid, 256, 0, s>>>(p); else wgrad_sd_kernel<NS_, 0, true><<<grid, 256, 0, s>>>(p); } \
    else { if (elu) wgrad_sd_kernel<NS_, 1><<<grid, 256, 0, s>>>(p); else wgrad_sd_kernel<NS_, 0><<<grid, 256, 0, s>>>(p); }                     \
    return 0;                                                                                      \
  }
  SD(1) SD(2)
#undef SD
  return 1;
}

int run_wgrad(const char* fn, int dtype, WgradParams& p, int taps, void* ws, int64_t ws_bytes, hipStream_t s) {
  WCfg c = pick_wcfg(p.N, p.D, p.H, p.W, p.Cin, p.Cout, taps, false);  // 8x8x16 tiles for k=1 measured slower (convT wgrad 0.76 -> 1.02 ms): off
  const int64_t slab = ((int64_t)taps * p.Cin + 1) * p.Cout * 4;   // dW partials + the bias-gradient row of a group
  int64_t need = (int64_t)c.groups * slab;
  if (taps == 27) need = std::max(need, (int64_t)sdm_plan(p.N, p.D, p.H, p.W, p.Cin, p.Cout).groups * slab);
  BPX_CHECK(ws != nullptr && ws_bytes >= need, "%s: workspace too small (%lld < %lld bytes)", fn, (long long)ws_bytes, (long long)need);
  p.part = reinterpret_cast<float*>(ws);
  if (k1_dma_ok(dtype, p, taps)) {   // raw-input k = 1 (the blocks' shortcut) at the large levels: the streaming kernel
    const int groups = k1_groups((int64_t)p.N * p.D * p.H * p.W, k1_instance(p.Cin / 16, p.Cout / 16));
    BPX_CHECK(ws_bytes >= (int64_t)groups * p.Cin * p.Cout * 4, "%s: workspace too small (%lld bytes)", fn, (long long)ws_bytes);
    BPX_CHECK(launch_wgrad_k1_dma(p, groups, s) == 0, "%s: no streaming instance", fn);
    BPX_LAUNCH_CHECK(fn);
    return finish_wgrad(fn, ReduceJob{p.part, p.dw, groups, 1, p.Cin, p.Cout, p.si, p.sj, p.st, p.off, nullptr, nullptr, 0, nullptr}, s);
  }
  int rc;
  const int64_t vox = (int64_t)p.N * p.D * p.H * p.W;
  const bool sd_ok = dtype == BPX_BF16 && taps == 27 && p.dy_vs == 1 && c.tx == 16 && c.tz == 4 && g_use_tr != 0 &&
                     vox * std::max(p.x_ld, p.dy_ld) < (1ll << 31) && (((uintptr_t)p.in_norm) & 7) == 0 &&
                     !(p.in_norm != nullptr && p.act > BPX_ACT_SILU);   // the round-4 activation codes take the generic kernel
  if (sd_ok && g_wgrad_sd != 0) rc = launch_wgrad_sd(p, c, s);   // measured faster at every cfg-2 layer with W > 8
  else if (dtype == BPX_BF16) rc = (taps == 27) ? launch_wgrad<uint16_t, 27>(p, c, g_use_tr != 0, s) : launch_wgrad<uint16_t, 1>(p, c, g_use_tr != 0, s);
  else rc = (taps == 27) ? launch_wgrad<float, 27>(p, c, false, s) : launch_wgrad<float, 1>(p, c, false, s);
  BPX_CHECK(rc == 0, "%s: no kernel for tile config", fn);
  BPX_LAUNCH_CHECK(fn);
  const float* dbpart = p.db ? p.part + (size_t)c.groups * taps * p.Cin * p.Cout : nullptr;   // where the launcher pointed the kernel
  return finish_wgrad(fn, ReduceJob{p.part, p.dw, c.groups, taps, p.Cin, p.Cout, p.si, p.sj, p.st, p.off, dbpart, p.db, 0, p.db ? p.db2 : nullptr}, s);
}

}  // namespace

// for the small weight-gradient kernels of elementwise.hip (first layer, rank-1 shortcut, head): the same fixed-order reduction of
// per-workgroup partials, queued with the others while the deferred mode is on
namespace bpxred {
int reduce_partials(const char* fn, const float* part, float* dw, int groups, int taps, int Cin, int Cout, int64_t si, int64_t sj, int64_t st,
                    const float* dbpart, float* db, int ndb, bool may_defer, hipStream_t s) {
  return reduce_partials2(fn, part, dw, groups, taps, Cin, Cout, si, sj, st, dbpart, db, nullptr, ndb, may_defer, s);
}
bool defer_active() { return t_defer.active; }
// dst[i] += sum over g < groups of rows[g * stride + i], i < n, in group order; queued like the others while the deferred mode is on
int reduce_rows(const char* fn, const float* rows, int groups, int64_t stride, int n, float* dst, hipStream_t s) {
  ReduceJob j{nullptr, nullptr, groups, 0, 0, n, 0, 0, 0, 0, rows, dst, n, nullptr, stride};
  return finish_wgrad(fn, j, s);
}
int reduce_partials2(const char* fn, const float* part, float* dw, int groups, int taps, int Cin, int Cout, int64_t si, int64_t sj, int64_t st,
                     const float* dbpart, float* db, float* db2, int ndb, bool may_defer, hipStream_t s) {
  const ReduceJob j{part, dw, groups, taps, Cin, Cout, si, sj, st, 0, db ? dbpart : nullptr, db, ndb, db ? db2 : nullptr};
  if (may_defer) return finish_wgrad(fn, j, s);
  wgrad_reduce_kernel<<<reduce_blocks(j), 1024, 0, s>>>(j);   // the caller reads the result right away (head gradients)
  BPX_LAUNCH_CHECK(fn);
  return 0;
}
}  // namespace bpxred

extern "C" int64_t bpx_conv3d_wgrad_workspace(int N, int D, int H, int W, int Cin, int Cout, int k) {
  int taps = k * k * k;
  WCfg c = pick_wcfg(N, D, H, W, Cin, Cout, taps, false);  // small tiles give the larger group count: an upper bound
  int64_t groups = c.groups;
  if (taps == 27) groups = std::max<int64_t>(groups, sdm_plan(N, D, H, W, Cin, Cout).groups);
  if (taps == 1) groups = std::max<int64_t>(groups, 256);   // the streaming k = 1 kernel: one slab per CU
  return groups * ((int64_t)taps * Cin + 1) * Cout * 4;   // per group: the dW partials and one row of bias-gradient column sums
}
extern "C" int64_t bpx_convT3d_k2s2_wgrad_workspace(int N, int D, int H, int W, int sz, int Cin, int Cout) {
  WCfg c = pick_wcfg(N, D, H, W, Cin, Cout, 1, false);           // fp32: one launch per sub-position
  CtCfg t = pick_ct(N, D, H, W, sz, Cin, Cout);                  // bf16: single pass, [groups][4*sz][Cin][Cout]
  return std::max(std::max((int64_t)c.groups * ((int64_t)Cin + 1) * Cout * 4, (int64_t)t.groups * ((int64_t)4 * sz * Cin + 1) * Cout * 4),
                  (int64_t)256 * ((int64_t)4 * sz * Cin + 1) * Cout * 4);   // (the streaming kernel: one slab per CU)
}

// test hook: 0 = scalar LDS gathers instead of ds_read_b64_tr_b16 in the bf16 wgrad
// bits 1..2 select the 3x3x3 bf16 schedule: 0 automatic, 2 = never the shift-dy kernel, 4 = always
extern "C" int bpx_debug_set_wgrad_tr(int use_tr) {
  g_use_tr = use_tr & 1;
  g_wgrad_sd = (use_tr & 4) ? 1 : (use_tr & 2) ? 0 : -1;
  g_sd_mc = ((use_tr >> 3) & 7) == 4 ? 0 : ((use_tr >> 3) & 3) ? ((use_tr >> 3) & 3) : -1;
  g_ct_resident = ((use_tr >> 6) & 1) ? 0 : 1;
  if ((use_tr >> 7) & 1) { g_k1_wgs = 1024 * (use_tr >> 8) / 100; g_sd_fill = 100; }
  else g_sd_fill = (use_tr >> 8) ? (use_tr >> 8) : 100;   // bits 8..: workgroups in percent of the co-resident capacity
  return 0;
}

// test / A-B hook: bit 0 = the windowed shift-dy wgrad kernel walks XCD-contiguous tile ranges in y-strips (default 1)
extern "C" int bpx_debug_set_tile_order(int bits) { g_sd_order = bits & 1; return 0; }

extern "C" int bpx_conv3d_wgrad(int dtype, int N, int D, int H, int W, bpx_tensor x, const bpx_norm_rec* in_norm_d, int act,
                                bpx_tensor dy, int k, float* dw_d, float* db_d, void* ws_d, int64_t ws_bytes, bpx_stream_t stream) {
  return bpx_conv3d_wgrad_db2(dtype, N, D, H, W, x, in_norm_d, act, dy, k, dw_d, db_d, nullptr, ws_d, ws_bytes, stream);
}

extern "C" int bpx_conv3d_wgrad_db2(int dtype, int N, int D, int H, int W, bpx_tensor x, const bpx_norm_rec* in_norm_d, int act,
                                    bpx_tensor dy, int k, float* dw_d, float* db_d, float* db2_d, void* ws_d, int64_t ws_bytes, bpx_stream_t stream) {
  BPX_CHECK(db2_d == nullptr || db_d != nullptr, "bpx_conv3d_wgrad_db2: db2_d needs db_d");
  BPX_CHECK(dy.cs == 0, "bpx_conv3d_wgrad: only x may be chunk-planar");
  BPX_CHECK(x.cs == 0 || (x.ld >= 16 && x.cs % 8 == 0 && x.cs >= ((int64_t)N * D * H * W - 1) * x.ld + 16 && x.cs * (x.C / 16) < (1ll << 31)),
            "bpx_conv3d_wgrad: bad chunk stride %lld", (long long)x.cs);
  const char* fn = "bpx_conv3d_wgrad";
  BPX_CHECK(dtype == BPX_BF16 || dtype == BPX_F32 || dtype == BPX_MIX16, "%s: dtype must be BF16, F32 or MIX16 (x fp16, dy bf16)", fn);
  const bool mix = dtype == BPX_MIX16;
  if (mix) dtype = BPX_BF16;
  BPX_CHECK(k == 1 || k == 3, "%s: k must be 1 or 3", fn);
  BPX_CHECK(x.ptr && dy.ptr && dw_d, "%s: null pointer", fn);
  BPX_CHECK(x.C % 16 == 0 && dy.C % 16 == 0, "%s: channels must be multiples of 16 (got %d, %d)", fn, x.C, dy.C);
  WgradParams p{};
  p.N = N; p.D = D; p.H = H; p.W = W;
  p.x = x.ptr; p.x_ld = x.ld; p.Cin = x.C; p.in_norm = in_norm_d; p.act = act;
  p.x_cs = x.cs ? (int)x.cs : 16;
  p.x_f16 = mix ? 1 : 0;
  p.dy = dy.ptr; p.dy_ld = dy.ld; p.Cout = dy.C; p.dy_vs = 1;
  int taps = k * k * k;
  p.dw = dw_d; p.si = taps; p.sj = (int64_t)x.C * taps; p.st = 1; p.off = 0;  // (Cout,Cin,k,k,k)
  p.db = db_d; p.db2 = db2_d;
  return run_wgrad(fn, dtype, p, taps, ws_d, ws_bytes, (hipStream_t)stream);
}

extern "C" int bpx_convT3d_k2s2_wgrad(int dtype, int N, int D, int H, int W, int sz, bpx_tensor x, bpx_tensor dy, float* dw_d, float* db_d,
                                      void* ws_d, int64_t ws_bytes, bpx_stream_t stream) {
  BPX_CHECK(x.cs == 0 && dy.cs == 0, "bpx_convT3d_k2s2_wgrad: chunk-planar tensors (cs != 0) are not accepted here");
  const char* fn = "bpx_convT3d_k2s2_wgrad";
  BPX_CHECK(dtype == BPX_BF16 || dtype == BPX_F32 || dtype == BPX_MIX16, "%s: dtype must be BF16, F32 or MIX16 (x fp16, dy bf16)", fn);
  const bool mix = dtype == BPX_MIX16;
  if (mix) dtype = BPX_BF16;
  BPX_CHECK(sz == 1 || sz == 2, "%s: z stride must be 1 or 2 (got %d)", fn, sz);
  const int nsub = 4 * sz;
  BPX_CHECK(x.ptr && dy.ptr && dw_d, "%s: null pointer", fn);
  BPX_CHECK(x.C % 16 == 0 && dy.C % 16 == 0, "%s: channels must be multiples of 16 (got %d, %d)", fn, x.C, dy.C);
  const int cts = cts_instance(x.C / 16, dy.C / 16);
  if (dtype == BPX_BF16 && g_use_tr != 0 && g_ct_dma && sz == 2 && cts && x.C % 16 == 0 && dy.C % 16 == 0 && W % 32 == 0 && (int64_t)N * D * H * W >= 65536 &&
      (int64_t)N * D * H * W * 8 * dy.ld * 2 < (1ll << 31) && (int64_t)N * D * H * W * x.ld * 2 < (1ll << 31) &&
      (((uintptr_t)x.ptr | (uintptr_t)dy.ptr) & 15) == 0 && (x.ld & 7) == 0 && (dy.ld & 7) == 0) {
    // the large levels (cfg 2: 32 -> 32 @64^3, 64 -> 64 @32^3): the streaming kernel, one persistent workgroup per CU
    CtsParams q{};
    q.x = x.ptr; q.x_ld = x.ld; q.dy = dy.ptr; q.dy_ld = dy.ld; q.Cin = x.C; q.Cout = dy.C; q.D = D; q.H = H; q.W = W;
    q.voxels = (int64_t)N * D * H * W;
    q.nblocks = (int)cdiv64(q.voxels, cts_tv(cts));
    q.groups = (int)std::min<int64_t>(256, q.nblocks);
    const int64_t need = (int64_t)q.groups * ((int64_t)nsub * x.C + 1) * dy.C * 4;
    BPX_CHECK(ws_d != nullptr && ws_bytes >= need, "%s: workspace too small (%lld < %lld bytes)", fn, (long long)ws_bytes, (long long)need);
    q.part = reinterpret_cast<float*>(ws_d);
    q.dbpart = db_d ? q.part + (size_t)q.groups * nsub * x.C * dy.C : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (cts == 1) { if (mix) wgrad_ct_dma_kernel<2, 2, 64, true><<<q.groups, 256, 0, s>>>(q); else wgrad_ct_dma_kernel<2, 2, 64, false><<<q.groups, 256, 0, s>>>(q); }
    else { if (mix) wgrad_ct_dma_kernel<4, 4, 32, true><<<q.groups, 256, 0, s>>>(q); else wgrad_ct_dma_kernel<4, 4, 32, false><<<q.groups, 256, 0, s>>>(q); }
    BPX_LAUNCH_CHECK(fn);
    return finish_wgrad(fn, ReduceJob{q.part, dw_d, q.groups, nsub, x.C, dy.C, (int64_t)dy.C * nsub, nsub, 1, 0, q.dbpart, db_d, 0}, s);
  }
  if (dtype == BPX_BF16 && g_use_tr != 0 && (int64_t)N * D * H * W * nsub * std::max(x.ld, dy.ld) < (1ll << 31)) {
    CtCfg c = pick_ct(N, D, H, W, sz, x.C, dy.C);
    const int64_t need = (int64_t)c.groups * ((int64_t)nsub * x.C + 1) * dy.C * 4;
    BPX_CHECK(ws_d != nullptr && ws_bytes >= need, "%s: workspace too small (%lld < %lld bytes)", fn, (long long)ws_bytes, (long long)need);
    WgradParams p{};
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.x = x.ptr; p.x_ld = x.ld; p.Cin = x.C; p.x_cs = 16; p.x_f16 = mix ? 1 : 0;
    p.dy = dy.ptr; p.dy_ld = dy.ld; p.Cout = dy.C;
    p.part = reinterpret_cast<float*>(ws_d); p.db = db_d;
    p.dbpart = p.part + (size_t)c.groups * nsub * x.C * dy.C;
    p.tilesY = c.tilesY; p.tilesX = c.tilesX; p.tilesPerSample = c.tilesPerSample; p.totalTiles = c.totalTiles; p.groups = c.groups;
    const int mc = ct_mc(N, D, H, W, sz, x.C, dy.C);
    const int nchunks = x.C / (16 * mc), nb = dy.C / (16 * c.ns);
    dim3 grid((unsigned)(((c.groups + 7) & ~7) * nchunks * nb));
    hipStream_t s = (hipStream_t)stream;
    if (mc == 2) wgrad_ct_kernel<2, 2, 2><<<grid, 256, 0, s>>>(p);      // (ct_mc: sz == 2 and 32-channel output blocks only)
    else if (sz == 2) { if (c.ns == 2) wgrad_ct_kernel<2, 2><<<grid, 256, 0, s>>>(p); else wgrad_ct_kernel<1, 2><<<grid, 256, 0, s>>>(p); }
    else { if (c.ns == 2) wgrad_ct_kernel<2, 1><<<grid, 256, 0, s>>>(p); else wgrad_ct_kernel<1, 1><<<grid, 256, 0, s>>>(p); }
    BPX_LAUNCH_CHECK(fn);
    // (Cin, Cout, sz, 2, 2): index = ci*Cout*nsub + co*nsub + sub
    return finish_wgrad(fn, ReduceJob{p.part, dw_d, c.groups, nsub, x.C, dy.C, (int64_t)dy.C * nsub, nsub, 1, 0, db_d ? p.dbpart : nullptr, db_d, 0}, s);
  }
  // one launch per sub-position, all through the same workspace: these reductions cannot wait
  struct NoDefer { bool was; NoDefer() : was(t_defer.active) { t_defer.active = false; } ~NoDefer() { t_defer.active = was; } } no_defer;
  for (int sub = 0; sub < nsub; ++sub) {
    WgradParams p{};
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.x = x.ptr; p.x_ld = x.ld; p.Cin = x.C; p.in_norm = nullptr; p.act = 0; p.x_cs = 16; p.x_f16 = mix ? 1 : 0;
    p.dy = dy.ptr; p.dy_ld = dy.ld; p.Cout = dy.C; p.dy_vs = 2; p.dy_vz = sz;
    p.dy_oz = (sub >> 2) & 1; p.dy_oy = (sub >> 1) & 1; p.dy_ox = sub & 1;
    p.dw = dw_d; p.si = (int64_t)dy.C * nsub; p.sj = nsub; p.st = 0; p.off = sub;  // (Cin,Cout,sz,2,2)
    p.db = db_d;  // every sub contributes its voxels to the bias gradient
    if (run_wgrad(fn, dtype, p, 1, ws_d, ws_bytes, (hipStream_t)stream)) return 1;
  }
  return 0;
}

extern "C" int bpx_debug_set_wgrad_k1(int on) { g_k1_dma = (on == 1 || on == 3 || on == 7) ? 1 : 0; g_ct_dma = (on == 1 || on == 5 || on == 7) ? 1 : 0; g_ct_dma_all = on == 7; return 0; }
extern "C" int bpx_debug_set_wgrad_cap(int percent) { g_cap_pct = percent > 0 ? percent : 100; return 0; }

// ---- deferred reductions -------------------------------------------------------------------------------------------------
extern "C" int bpx_wgrad_defer_begin(void) {
  t_defer.active = true;
  t_defer.jobs.clear();
  return 0;
}

extern "C" int bpx_wgrad_defer_flush(bpx_stream_t stream) {
  const char* fn = "bpx_wgrad_defer_flush";
  std::vector<ReduceJob> jobs;
  jobs.swap(t_defer.jobs);
  t_defer.active = false;
  hipStream_t s = (hipStream_t)stream;
  for (size_t base = 0; base < jobs.size(); base += RB_MAX) {
    ReduceBatch b{};
    b.count = (int)std::min<size_t>(RB_MAX, jobs.size() - base);
    int blocks = 0;
    for (int k = 0; k < b.count; ++k) {
      b.job[k] = jobs[base + k];
      b.first_block[k] = blocks;
      blocks += reduce_blocks(b.job[k]);
    }
    b.first_block[b.count] = blocks;
    if (blocks == 0) continue;
    wgrad_reduce_batch_kernel<<<blocks, 1024, 0, s>>>(b);
    BPX_LAUNCH_CHECK(fn);
  }
  return 0;
}
