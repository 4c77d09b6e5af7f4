// Backward of one 3x3x3 convolution in ONE pass over its operands: input gradient (dgrad, fused with the backward of the activation that
// precedes the conv and with the InstanceNorm reductions) AND weight / bias gradient (wgrad).
//
//   g[v][ci]        = sum_{tap,co} W[co][ci][tap] * dy[v - (tap - 1)][co] * act'(scale * t[v][ci] + shift)          (bpx_conv3d_dgrad)
//   dW[tap][ci][co] = sum_u act(scale * t[u][ci] + shift) * dy[u - (tap - 1)][co]                                    (bpx_conv3d_wgrad, shift-dy form)
//
// Both kernels stage the SAME two operands per tile - the haloed dy tile ([voxel][16 ch], 32 B per voxel, raw copy) and the un-haloed raw input
// tile t - and both are bound by that staging chain / by HBM requests at the 16-channel 128^3 layers, not by the matrix unit (DESIGN.md
// section 6): 16 -> 16 @128^3 moves 3 U + 2 U (U = 268 MB, one 16-channel tensor) as two kernels and 3 U here; 16 -> 48 moves 7 U + 4 U against 7 U.
// The kernel is a composition of tested parts: the MFMA step loop and the dgrad epilogue of conv3_lp_kernel (conv3d_lean.hip, 4x4x16 tile) and
// the windowed shift-dy MFMA phase of wgrad_sdm_kernel (wgrad_shared.h); what is new is the staging:
//   * dy halo and raw t tile arrive by `buffer_load ... lds` (no staging VGPRs, no ds_write pass; out-of-volume pieces use an out-of-range
//     buffer offset and land as zeros = the convolution's zero padding);
//   * a transform pass reads the raw t pieces back (each thread the pieces it requested itself: no barrier), applies normalise + activation
//     once per voxel and writes the bf16 MFMA operand of the wgrad phase; the raw copy stays in LDS for the dgrad epilogue (ELU' and the
//     normalised value need the raw t at every output voxel - formerly a second fetch of the tile from L2);
//   * persistent workgroups, XCD-contiguous tile ranges walked in y-strips (conv3d_shared.h decode_tile) so that halo re-reads hit the XCD's L2.
// dy has 16 channels (one chunk); t / g have 16 * CT channels.  Per-workgroup wgrad partials [grid][27][16 CT][16] are reduced by the batched
// fixed-order reduction of wgrad.hip (bpxred), so the parameter gradients stay bit-reproducible.
#include "conv3d_shared.h"
#include "wgrad_shared.h"

using namespace bpxconv;

// workgroups per CU the 16-channel instance is compiled for.  Measured (same box, 16 -> 16 @128^3): 4 per CU at 128 VGPRs 399 us, 3 per CU at
// 162 VGPRs 379 us - the third wave's registers buy more than the fourth workgroup; with 3 per CU there is also LDS for the packed weights.
#ifndef BPX_BWD_OCC1
#define BPX_BWD_OCC1 3
#endif
// CT == 1: the 14 KB of packed dgrad weights (the same for every tile of the persistent workgroup) live in LDS instead of being re-read from
// L2 per step with two loads in flight (the step loop was 4.1 K of the tile's 18 K cycles for 56 MFMAs = 0.9 K: it waited for its operands)
#ifndef BPX_BWD_LDSW
#define BPX_BWD_LDSW 1
#endif
#ifndef BPX_BWD_RS_WPRIO
#define BPX_BWD_RS_WPRIO 1
#endif

namespace {

#ifdef BPX_TICKET_PROBE
__device__ unsigned g_probe_ticket = 0u, g_probe_last = 0u;
#endif
struct BwdParams {
  int N, D, H, W;
  const void* dy; int dy_ld;                       // (N, D, H, W, 16) bf16
  const void* wT;                                  // BPX_PK_K3_T pack of the conv weight: [1][QPAD][Ct][8] bf16
  const void* t; int t_ld; int t_cs; int Ct;       // the conv's raw input (fp16 with TF16, else bf16), Ct = 16 * CT channels, maybe chunk-planar
  const bpx_norm_rec* t_norm; int act;
  void* g; int g_ld;                               // (N, D, H, W, Ct) bf16
  float* red;                                      // [N][tilesPerSample][2][Ct] partials of sum(g), sum(g * xhat)
  float* part; float* dbpart; int want_db;         // [grid][27][Ct][16] weight-gradient partials, [grid][16] bias-gradient partials
  int tilesZ, tilesY, tilesX, tilesPerSample, totalTiles, tilesPerXcd, stripY;
  long long* stamps;                               // profiling: per-workgroup cycle stamps [block][16] of the 5th tile (scripts/bwd_stamps.py), else null
};

// CG = 16-channel chunks of dy (1: the 16-channel layers of level 0; 2: the 32-channel layers of level 1), CT = chunks of t / g this workgroup
// owns; blockIdx.y walks the rest of t's channels (t.C = 16 CT gridDim.y: the 96-channel concat input of level 1 = 3 x CT 2), each y-block
// stages the whole dy halo and its own CT chunks of t.
template <int CG, int CT, int ACTK, bool TF16>
__global__ void __launch_bounds__(256, (CG == 1 && CT == 1) ? BPX_BWD_OCC1 : 2) conv3_bwd_kernel(const BwdParams p) {
  using T = uint16_t;                                                   // gradients, weights, MFMA operands: bf16
  using TT = typename std::conditional<TF16, f16_t, uint16_t>::type;    // storage type of the activation t
  constexpr int TZ = 4, TY = 4, TX = 16, TV = TZ * TY * TX;
  constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;
  constexpr int KPL = 8, VB = 32, NS = CT, MS = 4, STEPS = 14;
  constexpr int NPGT = HV * 2, NPG = (NPGT + 255) / 256;               // 16-byte pieces of the dy halo; piece idx at LDS byte idx * 16
  constexpr int SG_BYTES = HV * VB + 64;                                // + window over-read of the last halo row (sd_mfma_phase)
  constexpr int ST_BYTES = CT * TV * VB;
  constexpr int RED_BYTES = 4 * NS * 16 * 2 * 4;
  constexpr int HSTR = HX * VB;
  constexpr int NKC = TV / 32;
  constexpr bool LDSW = BPX_BWD_LDSW && CT == 1 && CG == 1;
  constexpr bool ALLNS = CG == 1 && CT == 3;                            // the 48-channel instance of level 0: NS accumulator sets at once (below)
  constexpr int SW_BYTES = LDSW ? STEPS * 1024 : 0;
  constexpr int QPADB = 56;                                             // k-groups of a packed chunk (bf16): 14 steps x 4
  __shared__ __attribute__((aligned(16))) unsigned char smem[CG * SG_BYTES + 2 * ST_BYTES + RED_BYTES + CT * 16 * 16 + SW_BYTES];
  unsigned char* sG = smem;                                             // dy halo [CG][HV][32 B]
  unsigned char* sT = smem + CG * SG_BYTES;                             // raw t tile [CT][TV][32 B]
  unsigned char* sA = sT + ST_BYTES;                                    // act(norm(t)) as bf16 [CT][TV][32 B]
  float* red = reinterpret_cast<float*>(sA + ST_BYTES);                 // statistics scratch [wave][NS * 16][2]
  float* sN = red + RED_BYTES / 4;                                      // [CT * 16]{mean, rstd, scale, shift}: the sample's norm records (staging and epilogue)
  unsigned char* sW = reinterpret_cast<unsigned char*>(sN + CT * 16 * 4); // LDSW: packed weights [14 steps][64 lanes][16 B]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;   // (as an SGPR via readfirstlane the 48-channel instance spills 92 bytes per lane: measured, left a VGPR)
  const int j = lane & 15, g = lane >> 4;
  const int D = p.D, H = p.H, W = p.W, Ct = p.Ct;
  const int cb0 = (int)blockIdx.y * CT;                                 // first t / g chunk of this workgroup
  constexpr int Cdy = 16 * CG;

  // ---- per-workgroup constants ------------------------------------------------------------------------------------------------------
  // dgrad phase (conv3_lp_kernel, TX = 16): this lane's output voxel of m-subtile 0 is (wave, 0, j); m-subtile ms adds ms rows
  const int cg_off = (g & 1) * 16;
  const bool hi_tap = (g >> 1) != 0;
  const int hb0 = ((wave * HY) * HX + j) * VB + cg_off;
  const int lbase[4] = {hb0 + (hi_tap ? VB : 0), hb0 + (hi_tap ? HX * VB : 0), hb0 + (hi_tap ? HY * HX * VB : 0), hb0};
  const int evox_rel = (wave * H) * W + j;
  const uint32_t wlane = (uint32_t)((g * Ct + cb0 * 16 + j) * KPL) * 2u;
  // wgrad phase (wgrad_sdm_kernel): transposing-read bases
  const int trl = (j >> 2), trc = (j & 3) * 8;
  const int a_base = g * 8 * VB + trl * VB + trc;
  const int g_lane = (((g >> 1) * HX + (g & 1) * 8) + trl) * VB + trc;
  // staging: this thread's pieces.  The byte offsets and halo coordinates of the dy pieces are RE-DERIVED per tile from an opaque copy of the
  // thread id (a dozen VALU instructions per piece against ~1000 per tile): as per-workgroup constants they were nine more live registers, the
  // compiler spilled them, and every reload in the staging section waited with `s_waitcnt vmcnt(0)` - which also waits for the DMA pieces
  // issued before it (measured: "DMA issue" 2.6 K of a 16-channel tile's 14 K cycles, 9.9 K of 35 K for 48 channels)
  const int sub = tid & 1;
  const int hv0 = tid >> 1;                                                      // halo voxel of this thread's dy piece 0
  const uint32_t hpk0 = (uint32_t)(hv0 / (HX * HY)) | ((uint32_t)((hv0 / HX) % HY) << 8) | ((uint32_t)(hv0 % HX) << 16);
  const uint32_t dy_ld2 = (uint32_t)p.dy_ld * 2u;
  uint32_t rel_t[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int tv = (u * 256 + tid) >> 1;
    rel_t[u] = (uint32_t)((((tv >> 6) * H + ((tv >> 4) & 3)) * W + (tv & 15)) * p.t_ld + sub * KPL) * 2u;
    asm volatile("" : "+v"(rel_t[u]));
  }
  const bool last_ok = (NPG - 1) * 256 + tid < NPGT;
  const uint32_t t_csb = (uint32_t)p.t_cs * 2u;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, (int)0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.t), 0, (int)0x80000000u, 0x00020000);
  constexpr uint32_t OOR = 0x80000000u;   // out of range of the buffer: the DMA writes zeros

  f32x4_t accw[CG][7][CT];                // weight-gradient accumulators of this wave's taps [7 wave, 7 wave + 7), kept over all tiles
#pragma unroll
  for (int k = 0; k < CG; ++k)
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
      for (int c = 0; c < CT; ++c) accw[k][a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool want_b = p.want_db != 0 && blockIdx.y == 0;
  // InstanceNorm-backward statistics sum(g), sum(g xhat): per-lane partial sums kept over ALL tiles of a sample; the 16-lane / 4-wave reduction
  // and the global row are paid once per (workgroup, sample) instead of once per tile (32 DPP adds, an LDS exchange, a barrier and a row store
  // per tile: 1.5 K + part of the epilogue's 4.1 K of the tile's 18 K cycles)
  constexpr bool PSTATS = !ALLNS;    // the 48-channel instance writes one statistics row per tile (below)
  float ps1[NS][4], ps2[NS][4];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ps1[ns][r] = 0.f; ps2[ns][r] = 0.f; }
  int cur_tile = 0;
  auto flush_stats = [&](int nn) {   // workgroup-uniform call
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(ps1[ns][r]), b = row16_sum(ps2[ns][r]);
        if (j == 0) *reinterpret_cast<f32x2_t*>(&red[((wave * NS * 16) + ns * 16 + g * 4 + r) * 2]) = f32x2_t{a, b};
        ps1[ns][r] = 0.f; ps2[ns][r] = 0.f;
      }
    __syncthreads();
    if (tid < NS * 16 * 2) {
      const int c = tid >> 1, k = tid & 1;
      const float a = red[(0 * NS * 16 + c) * 2 + k] + red[(1 * NS * 16 + c) * 2 + k] + red[(2 * NS * 16 + c) * 2 + k] + red[(3 * NS * 16 + c) * 2 + k];
      p.red[(((size_t)nn * gridDim.x + blockIdx.x) * 2 + k) * Ct + cb0 * 16 + c] = a;
    }
    __syncthreads();
  };
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wT), 0, CG * QPADB * Ct * 16, 0x00020000);
  if constexpr (LDSW) {   // this wave's share of the 14 weight pieces; they have landed and are visible after the first tile's staging barrier
    for (int q = wave; q < STEPS; q += 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sW + q * 1024), 16, (uint32_t)(q * 1024 + lane * 16), 0, 0, 0);
  }
  int n_first = -1;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, spx = gridDim.x >> 3;
  int n_cur = -1;
  // cycle stamps: only in a profiling build (bash scripts/ab_build_flags.sh stamps -DBPX_BWD_STAMPS; scripts/bwd_stamps.py) - the pointer and
  // the counters cost the 128-VGPR instance 16 bytes of scratch
#ifdef BPX_BWD_STAMPS
  long long* stamps = (p.stamps && tid == 0) ? p.stamps + (size_t)blockIdx.x * 16 : nullptr;
  int stamp_i = 0, it = 0;
#define BPX_STAMP() do { if (stamps && it == 4 && stamp_i < 15) stamps[stamp_i++] = (long long)__builtin_readcyclecounter(); } while (0)
#else
  int it = 0;
#define BPX_STAMP() do { } while (0)
#endif

  for (int local = slot; local < p.tilesPerXcd; local += spx, ++it) {
    const int tileId = xcd * p.tilesPerXcd + local;
    if (tileId >= p.totalTiles) break;
    int n, tzi, tyi, txi;
    decode_tile(tileId, p.tilesZ, p.tilesY, p.tilesX, p.tilesPerSample, p.stripY, n, tzi, tyi, txi);
    cur_tile = (tzi * p.tilesY + tyi) * p.tilesX + txi;
    const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;
    const bool full = z0 + TZ <= D && y0 + TY <= H && x0 + TX <= W;
    const bool interior = full && z0 >= 1 && z0 + TZ + 1 <= D && y0 >= 1 && y0 + TY + 1 <= H && x0 >= 1 && x0 + TX + 1 <= W;
    const uint32_t base_g = (uint32_t)(((n * D + z0 - 1) * H + (y0 - 1)) * W + (x0 - 1)) * (uint32_t)p.dy_ld * 2u;
    const uint32_t base_t = (uint32_t)(((n * D + z0) * H + y0) * W + x0) * (uint32_t)p.t_ld * 2u;

    BPX_STAMP();   // 0: tile start
    __syncthreads();   // the previous tile's MFMA phases and epilogue are done with sG / sT / sA
    BPX_STAMP();   // 1: top barrier passed
    // ---- staging: LDS-DMA of the raw t tile and of the dy halo ---------------------------------------------------------------------
    bool okt[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int tv = (u * 256 + tid) >> 1;
      okt[u] = full || (z0 + (tv >> 6) < D && y0 + ((tv >> 4) & 3) < H && x0 + (tv & 15) < W);
    }
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t, (lds_ptr_t)(sT + c * TV * VB + (u * 256 + wave * 64) * 16), 16,
                                                 okt[u] ? base_t + rel_t[u] + (uint32_t)(cb0 + c) * t_csb : OOR, 0, 0, 0);
    {
      uint32_t pk = hpk0;
      asm volatile("" : "+v"(pk));   // keeps the piece arithmetic inside the tile loop (see above)
      int hz = (int)(pk & 255u), hy = (int)((pk >> 8) & 255u), hx = (int)(pk >> 16);
      static_assert(HX * HY + HX + 2 == 128 && NPG * 128 >= HV, "piece u + 1 = piece u + 128 voxels = one plane + one row + 2");
#pragma unroll
      for (int u = 0; u < NPG; ++u) {
        // branch-free (bit operators): the short-circuit form compiled to three branches per piece
        const int ok = (int)interior | ((int)((unsigned)(z0 - 1 + hz) < (unsigned)D) & (int)((unsigned)(y0 - 1 + hy) < (unsigned)H) & (int)((unsigned)(x0 - 1 + hx) < (unsigned)W));
        const uint32_t inoff = base_g + (uint32_t)((hz * H + hy) * W + hx) * dy_ld2 + (uint32_t)sub * 16u;
        const uint32_t off = ok ? inoff : OOR;
        if (u < NPG - 1 || last_ok) {
#pragma unroll
          for (int k = 0; k < CG; ++k)       // chunk k of the voxel's dy channels: 32 bytes further
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)(sG + k * SG_BYTES + (u * 256 + wave * 64) * 16), 16, ok ? inoff + (uint32_t)k * 32u : OOR, 0, 0, 0);
        }
        (void)off;
        hx += 2; hy += 1; hz += 1;
        const int cx = hx >= HX; hx -= cx * HX; hy += cx;
        const int cy = hy >= HY; hy -= cy * HY; hz += cy;
      }
    }
    if (n != n_cur) {   // uniform: a workgroup crosses a sample boundary at most N - 1 times
      if (PSTATS && n_cur >= 0) flush_stats(n_cur);
      if (n_cur < 0) n_first = n;
      if (tid < CT * 16) reinterpret_cast<f32x4_t*>(sN)[tid] = *reinterpret_cast<const f32x4_t*>(&p.t_norm[(size_t)n * Ct + cb0 * 16 + tid]);
      n_cur = n;
      __syncthreads();
    }
    BPX_STAMP();   // 2: DMA issued
    // ---- transform: raw t -> normalise + activation -> bf16 operand of the wgrad phase (own pieces only: no barrier before it) --------
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA pieces have landed
    BPX_STAMP();   // 3: DMA landed
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      float psc[KPL], psh[KPL];
      {
        const f32x4_t* q = reinterpret_cast<const f32x4_t*>(sN) + (c * 16 + sub * KPL);
#pragma unroll
        for (int e = 0; e < KPL; ++e) {
          const f32x4_t v = q[e];
          psc[e] = v[2]; psh[e] = v[3];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(sT + c * TV * VB + (u * 256 + tid) * 16);
        if (okt[u]) {   // out-of-volume voxels of ragged tiles stay zero: the conv's padding applies to the ACTIVATED tensor
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float a = fmaf(psc[2 * q], lo16<TT>(v[q]), psh[2 * q]), b = fmaf(psc[2 * q + 1], hi16<TT>(v[q]), psh[2 * q + 1]);
            act_pair<ACTK>(a, b, p.act);
            v[q] = cvt_pk_bf16(a, b);
          }
        }
        *reinterpret_cast<u32x4_t*>(sA + c * TV * VB + (u * 256 + tid) * 16) = v;
      }
    }
    BPX_STAMP();   // 4: transformed
    __syncthreads();   // every wave's DMA pieces and activated pieces are visible
    BPX_STAMP();   // 5: barrier

    // ---- phase A + dgrad epilogue, one 16-channel group of g at a time ------------------------------------------------------------------
    // (conv3_lp_kernel's step loop on the dy halo; one input chunk of 16 dy channels.)  Group by group - not NS accumulator sets at once - so
    // that 16 accumulator registers are live instead of 48: the 48-channel instance then has room for the per-lane statistics (its per-tile
    // statistics row cost 4.5 K of the tile's 29 K cycles) and for fragment reads one stage ahead of the MFMAs; the dy fragments are re-read
    // from LDS per group (90 instead of 30 ds_read_b128 per wave and tile: the LDS has the slack, two workgroups per CU).
    const int vox0 = ((n * D + z0) * H + y0) * W + x0 + evox_rel;
    const bool okzx = full || (z0 + wave < D && x0 + j < W);
    const int yrem = full ? (1 << 20) : H - y0;
    char* __restrict__ yout = reinterpret_cast<char*>(p.g);
    const uint32_t yrow = (uint32_t)(W * p.g_ld) * 2u;
    const uint32_t yb0 = (uint32_t)(vox0 * p.g_ld + cb0 * 16 + g * 4) * 2u;
    const unsigned char* tl = sT + ((wave * MS) * 16 + j) * VB + g * 8;   // raw t of (voxel (wave, ms, j), channels 4 g .. 4 g + 3): + ms * 512 + ns * TV * VB
    // weights of step s and group ns: LDS (CT == 1), or a BUFFER load - resource + one lane-offset VGPR + the step's byte offset in an SGPR +
    // the group as immediate.  (As pointer arithmetic the compiler hoists the 14 x NS 64-bit lane addresses out of the tile loop: 84 VGPRs, spilled.)
    const uint32_t wstep = LDSW ? 1024u : (uint32_t)(4 * Ct * 16);
    // group-by-group form: branch-free rows as in the role-split kernel (masks + buffer stores with out-of-range offsets for out-of-volume rows)
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.g, 0, (int)0xFFFFFFF0u, 0x00020000);
    uint32_t mk[MS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
      mk[ms] = (okzx && ms < yrem) ? 0xFFFFFFFFu : 0u;
      asm volatile("" : "+v"(mk[ms]));
    }
    if constexpr (ALLNS) {
      // NS accumulator sets at once, weights through a one-step register ring (conv3_lp_kernel's form), statistics row per TILE.  Measured
      // alternatives for the 48-channel instance: the per-group form below + per-lane statistics over all tiles (what the 16-channel instance
      // does) wants 418 registers and spills 600-750 bytes per lane at the 256 of two workgroups per CU - not usable.
      f32x4_t acc[MS][NS];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      auto load_w = [&](int s_, int ns) -> u32x4_t {
        return __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wlane + ns * 256u), (int)(s_ * wstep), 0));
      };
      u32x4_t wq[2][NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) wq[0][ns] = load_w(0, ns);
#pragma unroll
      for (int s_ = 0; s_ < STEPS; ++s_) {
        if (s_ + 1 < STEPS) {
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) wq[(s_ + 1) & 1][ns] = load_w(s_ + 1, ns);
        }
        const int cls = s_ < 9 ? 0 : s_ < 12 ? 1 : s_ == 12 ? 2 : 3;
        const int imm = tap_off<HY, HX, VB>(bpx_tap_order_bf16(2 * s_));
        u32x4_t af[MS];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) af[ms] = *reinterpret_cast<const u32x4_t*>(sG + lbase[cls] + ms * HSTR + imm);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = mfma_step<T>(wq[s_ & 1][ns], af[ms], acc[ms][ns]);
      }
      __builtin_amdgcn_sched_barrier(0);
      BPX_STAMP();   // 6: dgrad steps done
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        u32x2_t tv[MS];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) tv[ms] = *reinterpret_cast<const u32x2_t*>(tl + ns * TV * VB + ms * 16 * VB);
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        if (ACTK == 1) {
#pragma unroll
          for (int rp = 0; rp < 4; rp += 2) {
            const f32x4_t* rsrc = reinterpret_cast<const f32x4_t*>(sN) + (ns * 16 + g * 4 + rp);
            const f32x4_t ra = rsrc[0], rb = rsrc[1];
            const f32x2_t sc2{ra[2], rb[2]}, sh2{ra[3], rb[3]}, rs2{ra[1], rb[1]}, nm2{-ra[0] * ra[1], -rb[0] * rb[1]};
            f32x2_t s1p{0.f, 0.f}, s2p{0.f, 0.f};
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) {
              const uint32_t w = tv[ms][rp >> 1];
              const f32x2_t tt{lo16<TT>(w), hi16<TT>(w)};
              const f32x2_t u = __builtin_elementwise_fma(sc2, tt, sh2);
              const f32x2_t xh = __builtin_elementwise_fma(rs2, tt, nm2);
              const f32x2_t e = u * f32x2_t{1.44269504088896341f, 1.44269504088896341f};
              f32x2_t a_{__builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(e[0]), 0.f, 1.f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(e[1]), 0.f, 1.f)};
              const bool in = okzx && ms < yrem;
              const f32x2_t gv = in ? f32x2_t{acc[ms][ns][rp], acc[ms][ns][rp + 1]} * a_ : f32x2_t{0.f, 0.f};
              acc[ms][ns][rp] = gv[0]; acc[ms][ns][rp + 1] = gv[1];
              s1p = s1p + gv;
              s2p = __builtin_elementwise_fma(gv, xh, s2p);
            }
            s1[rp] = s1p[0]; s1[rp + 1] = s1p[1]; s2[rp] = s2p[0]; s2[rp + 1] = s2p[1];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4_t rec = reinterpret_cast<const f32x4_t*>(sN)[ns * 16 + g * 4 + r];
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) {
              const uint32_t w = tv[ms][r >> 1];
              const float tf = (r & 1) ? hi16<TT>(w) : lo16<TT>(w);
              const float u = fmaf(rec[2], tf, rec[3]);
              const float gv = (okzx && ms < yrem) ? acc[ms][ns][r] * apply_act_bwd_rt<T, ACTK>(u, p.act) : 0.f;
              acc[ms][ns][r] = gv;
              s1[r] += gv;
              s2[r] += gv * ((tf - rec[0]) * rec[1]);
            }
          }
        }
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
          if (okzx && ms < yrem)
            *reinterpret_cast<u32x2_t*>(yout + (yb0 + ms * yrow + ns * 32u)) = u32x2_t{cvt_pk_bf16(acc[ms][ns][0], acc[ms][ns][1]), cvt_pk_bf16(acc[ms][ns][2], acc[ms][ns][3])};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a_ = row16_sum(s1[r]), b_ = row16_sum(s2[r]);
          if (j == 0) *reinterpret_cast<f32x2_t*>(&red[((wave * NS * 16) + ns * 16 + g * 4 + r) * 2]) = f32x2_t{a_, b_};
        }
      }
      BPX_STAMP();   // 7: epilogue stores issued
      __syncthreads();
      if (tid < NS * 16 * 2) {
        const int c = tid >> 1, k = tid & 1;
        const float a_ = red[(0 * NS * 16 + c) * 2 + k] + red[(1 * NS * 16 + c) * 2 + k] + red[(2 * NS * 16 + c) * 2 + k] + red[(3 * NS * 16 + c) * 2 + k];
        p.red[(((size_t)n * p.tilesPerSample + cur_tile) * 2 + k) * Ct + c] = a_;
      }
    } else {
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      f32x4_t acc[MS];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) acc[ms] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      // Fragment reads one stage ahead of the MFMAs (CG == 1).  Stages 0..2 = the three dz planes (steps 3 dz .. 3 dz + 2: MS + 2 fragment rows
      // read once and slid over the three dy steps, as in conv3_lp_kernel's REUSE form), stages 3..7 = steps 9..13.  With two dy chunks the
      // second register set does not fit beside twice the weight-gradient accumulators: one set, the K loop runs chunk after chunk.
      u32x4_t rowA[MS + 2], rowB[CG == 1 ? MS + 2 : 1], wA[3], wB[CG == 1 ? 3 : 1];
#pragma unroll
      for (int ck = 0; ck < CG; ++ck) {
        const unsigned char* sGk = sG + ck * SG_BYTES;
        auto load_w = [&](int s_) -> u32x4_t {
          if constexpr (LDSW) return *reinterpret_cast<const u32x4_t*>(sW + lane * 16 + s_ * 1024);
          else return __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wlane + ns * 256u), (int)((ck * QPADB / 4 + s_) * wstep), 0));
        };
        auto load_plane = [&](int dz, u32x4_t* row, u32x4_t* w) {
#pragma unroll
          for (int q = 0; q < 3; ++q) w[q] = load_w(3 * dz + q);
#pragma unroll
          for (int r = 0; r < MS + 2; ++r) row[r] = *reinterpret_cast<const u32x4_t*>(sGk + lbase[0] + r * HSTR + tap_off<HY, HX, VB>(9 * dz));
        };
        auto load_step = [&](int s_, u32x4_t* af, u32x4_t* w) {
          const int cls = s_ < 12 ? 1 : s_ == 12 ? 2 : 3;
          const int imm = tap_off<HY, HX, VB>(bpx_tap_order_bf16(2 * s_));
          w[0] = load_w(s_);
#pragma unroll
          for (int ms = 0; ms < MS; ++ms) af[ms] = *reinterpret_cast<const u32x4_t*>(sGk + lbase[cls] + ms * HSTR + imm);
        };
        auto mma_plane = [&](const u32x4_t* row, const u32x4_t* w) {
#pragma unroll
          for (int dyy = 0; dyy < 3; ++dyy)
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) acc[ms] = mfma_step<T>(w[dyy], row[ms + dyy], acc[ms]);
        };
        auto mma_step = [&](const u32x4_t* af, const u32x4_t* w) {
#pragma unroll
          for (int ms = 0; ms < MS; ++ms) acc[ms] = mfma_step<T>(w[0], af[ms], acc[ms]);
        };
        __builtin_amdgcn_sched_barrier(0);   // (the next group's first fragment reads stay out of this group's epilogue: register budget)
        if constexpr (CG == 1) {
          load_plane(0, rowA, wA);
          load_plane(1, rowB, wB);
          __builtin_amdgcn_sched_barrier(0);
          mma_plane(rowA, wA);
          __builtin_amdgcn_sched_barrier(0);
          load_plane(2, rowA, wA);
          __builtin_amdgcn_sched_barrier(0);
          mma_plane(rowB, wB);
          __builtin_amdgcn_sched_barrier(0);
          load_step(9, rowB, wB);
          __builtin_amdgcn_sched_barrier(0);
          mma_plane(rowA, wA);
          __builtin_amdgcn_sched_barrier(0);
          load_step(10, rowA, wA);
          __builtin_amdgcn_sched_barrier(0);
          mma_step(rowB, wB);
          __builtin_amdgcn_sched_barrier(0);
          load_step(11, rowB, wB);
          __builtin_amdgcn_sched_barrier(0);
          mma_step(rowA, wA);
          __builtin_amdgcn_sched_barrier(0);
          load_step(12, rowA, wA);
          __builtin_amdgcn_sched_barrier(0);
          mma_step(rowB, wB);
          __builtin_amdgcn_sched_barrier(0);
          load_step(13, rowB, wB);
          __builtin_amdgcn_sched_barrier(0);
          mma_step(rowA, wA);
          __builtin_amdgcn_sched_barrier(0);
          mma_step(rowB, wB);
        } else {
#pragma unroll
          for (int dz = 0; dz < 3; ++dz) {
            load_plane(dz, rowA, wA);
            __builtin_amdgcn_sched_barrier(0);
            mma_plane(rowA, wA);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int s_ = 9; s_ < STEPS; ++s_) {
            load_step(s_, rowA, wA);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(rowA, wA);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ns == NS - 1) BPX_STAMP();   // 6: dgrad steps done (last group)

      // ---- epilogue of the group: g = acc * act'(scale t + shift), per-lane partials of sum(g) and sum(g xhat), 8-byte stores ----------------
      u32x2_t tv[MS];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) tv[ms] = *reinterpret_cast<const u32x2_t*>(tl + ns * TV * VB + ms * 16 * VB);
      if (ACTK == 1) {
#pragma unroll
        for (int rp = 0; rp < 4; rp += 2) {
          const f32x4_t* rsrc = reinterpret_cast<const f32x4_t*>(sN) + (ns * 16 + g * 4 + rp);   // from LDS: no L2 latency in the epilogue chain
          const f32x4_t ra = rsrc[0], rb = rsrc[1];
          const f32x2_t sc2{ra[2], rb[2]}, sh2{ra[3], rb[3]}, rs2{ra[1], rb[1]}, nm2{-ra[0] * ra[1], -rb[0] * rb[1]};
          f32x2_t s1p{0.f, 0.f}, s2p{0.f, 0.f};
#pragma unroll
          for (int ms = 0; ms < MS; ++ms) {
            const uint32_t w = tv[ms][rp >> 1];
            const f32x2_t tt{lo16<TT>(w), hi16<TT>(w)};
            const f32x2_t u = __builtin_elementwise_fma(sc2, tt, sh2);
            const f32x2_t xh = __builtin_elementwise_fma(rs2, tt, nm2);
            const f32x2_t e = u * f32x2_t{1.44269504088896341f, 1.44269504088896341f};
            f32x2_t a{__builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(e[0]), 0.f, 1.f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(e[1]), 0.f, 1.f)};
            const f32x2_t gu = f32x2_t{acc[ms][rp], acc[ms][rp + 1]} * a;            // out-of-volume voxels of edge tiles carry no gradient:
            const f32x2_t gv{__uint_as_float(__float_as_uint(gu[0]) & mk[ms]), __uint_as_float(__float_as_uint(gu[1]) & mk[ms])};   // a mask, not a branch
            acc[ms][rp] = gv[0]; acc[ms][rp + 1] = gv[1];
            s1p = s1p + gv;
            s2p = __builtin_elementwise_fma(gv, xh, s2p);
          }
          ps1[ns][rp] += s1p[0]; ps1[ns][rp + 1] += s1p[1]; ps2[ns][rp] += s2p[0]; ps2[ns][rp + 1] += s2p[1];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4_t rec = reinterpret_cast<const f32x4_t*>(sN)[ns * 16 + g * 4 + r];
#pragma unroll
          for (int ms = 0; ms < MS; ++ms) {
            const uint32_t w = tv[ms][r >> 1];
            const float tf = (r & 1) ? hi16<TT>(w) : lo16<TT>(w);
            const float u = fmaf(rec[2], tf, rec[3]);
            const float gv = __uint_as_float(__float_as_uint(acc[ms][r] * apply_act_bwd_rt<T, ACTK>(u, p.act)) & mk[ms]);
            acc[ms][r] = gv;
            ps1[ns][r] += gv;
            ps2[ns][r] += gv * ((tf - rec[0]) * rec[1]);
          }
        }
      }
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)      // (out-of-volume rows: an offset beyond the buffer)
        __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{cvt_pk_bf16(acc[ms][0], acc[ms][1]), cvt_pk_bf16(acc[ms][2], acc[ms][3])}, rs_y,
                                              (int)((yb0 + ms * yrow + ns * 32u) | ~mk[ms]), 0, 0);
    }
    }
    if (CT == 1) BPX_STAMP();   // 7: epilogue stores issued
    BPX_STAMP();   // 8: statistics row written (48-channel instance)
    // ---- phase B: wgrad MFMA steps (windowed shift-dy phase of wgrad_sdm_kernel) on the same staged operands -----------------------------
#pragma unroll
    for (int ck = 0; ck < CG; ++ck) {   // one 16-channel block of dy (= of the conv's output channels) at a time, on the same activated tile
      const unsigned char* sGk = sG + ck * SG_BYTES;
      switch (wave) {   // wave-uniform
        case 0: bpxwg::sd_mfma_phase<0, CT, HY, HX, VB, VB, TV, NKC>(sA, sGk, a_base, g_lane, accw[ck], want_b); break;
        case 1: bpxwg::sd_mfma_phase<1, CT, HY, HX, VB, VB, TV, NKC>(sA, sGk, a_base, g_lane, accw[ck], want_b); break;
        case 2: bpxwg::sd_mfma_phase<2, CT, HY, HX, VB, VB, TV, NKC>(sA, sGk, a_base, g_lane, accw[ck], want_b); break;
        default: bpxwg::sd_mfma_phase<3, CT, HY, HX, VB, VB, TV, NKC>(sA, sGk, a_base, g_lane, accw[ck], want_b); break;
      }
    }
    BPX_STAMP();   // 9: wgrad steps done
  }
#undef BPX_STAMP
  // statistics rows [N][grid][2][Ct]: the last sample's sums, and zeros for the samples this workgroup never touched
  if (PSTATS && n_cur >= 0) flush_stats(n_cur);
  if (PSTATS && tid < NS * 16 * 2) {
    const int c = tid >> 1, k = tid & 1;
    for (int nn = 0; nn < p.N; ++nn)
      if (n_cur < 0 || nn < n_first || nn > n_cur) p.red[(((size_t)nn * gridDim.x + blockIdx.x) * 2 + k) * Ct + cb0 * 16 + c] = 0.f;
  }

  // ---- flush of the weight-gradient partials: lane holds D[ci = 4 g + r][co = j] of its taps ---------------------------------------------
  float* pp = p.part + (size_t)blockIdx.x * 27 * Ct * Cdy;
#pragma unroll
  for (int ck = 0; ck < CG; ++ck)
#pragma unroll
    for (int a = 0; a < 7; ++a) {
      const int tap = 7 * wave + a;
      if (tap >= 27) continue;
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[((size_t)tap * Ct + (cb0 + c) * 16 + 4 * g + r) * Cdy + ck * 16 + j] = accw[ck][a][c][r];
    }
  if (want_b && wave == 3 && g == 0) {
#pragma unroll
    for (int ck = 0; ck < CG; ++ck) p.dbpart[(size_t)blockIdx.x * Cdy + ck * 16 + j] = accw[ck][6][0][0];
  }
#ifdef BPX_TICKET_PROBE
  // measurement only (profiles/r04_ticket_probe.txt): what a last-arriver finalize would add to every workgroup - an agent-scope RELEASE after its
  // last store, one ticket; the last arriver alone pays the ACQUIRE
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (tid == 0) {
    const unsigned t = atomicAdd(&g_probe_ticket, 1u);
    if (t == gridDim.x * gridDim.y - 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); g_probe_ticket = 0u; g_probe_last = (unsigned)blockIdx.x; }
  }
#endif
}


// ---- ROLE-SPLIT form (round 6) ----------------------------------------------------------------------------------------------------------------
// The kernel above runs a tile's phases one after the other in every wave - staging, transform, dgrad MFMA steps, dgrad epilogue, wgrad MFMA steps -
// at two workgroups (8 waves) per CU; its per-phase stamps (profiles/r05_stamps_bwd_fused.txt) show the two MFMA phases at a third of a tile and
// the VALU phases (transform, epilogue) and the exposed DMA latency at the rest.  Here ONE workgroup of 8 waves owns the CU and splits the work by
// ROLE, two tiles in flight in a double-buffered LDS (2 x (dy halo + raw t + act(t)) = 140 KB for 48 channels):
//   waves 0-3 (D): barrier X_i - dgrad MFMA steps on G[b] - dgrad epilogue with the raw T[b] (ELU', per-lane statistics, g stores)
//   waves 4-7 (W): wait for their own DMA pieces of tile i - transform T[b] -> A[b] - barrier X_i - request tile i + 1 into G[b^1], T[b^1] by
//                  `buffer_load ... lds` - wgrad MFMA steps on A[b], G[b]
// One barrier per tile.  At X_i the D waves have finished tile i - 1 and the W waves its wgrad phase, so buffer b ^ 1 is free for the requests of
// tile i + 1, which then have a whole wgrad phase to land.  Each SIMD holds one wave of either role: the VALU sections of one overlap the MFMA
// steps of the other, and neither role carries the other's accumulators (the D waves have room for per-lane statistics over all tiles of a
// sample: no per-tile statistics row, no cross-wave exchange - every D wave writes its own row).  The norm records of the sample are wave-private
// LDS copies (the W waves run one tile ahead of the D waves and may be in the next sample).
// One LDS-DMA piece (64 lanes x 16 bytes -> LDS at `lds` + 16 lane) as INLINE ASSEMBLY: hipcc's wait-count pass treats an LDS-DMA intrinsic as a
// store to "some LDS" and puts `s_waitcnt vmcnt(0)` in front of the next ds_read of ANY LDS address - the role-split kernel's W waves then waited
// for the requests of the NEXT tile (other buffer) at the top of their wgrad phase.  The assembly form is invisible to that pass; every consumer
// of the pieces sits behind an explicit `s_waitcnt vmcnt(0)` and a barrier (conv3_bwd_rs_kernel).
__device__ __forceinline__ void dma16_asm(const u32x4_t& rsrc, uint32_t lds, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds), "v"(voff), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ u32x4_t raw_rsrc(const void* ptr, uint32_t num_records) {
  const uint64_t a = reinterpret_cast<uint64_t>(ptr);
  return u32x4_t{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) & 0xFFFFu, num_records, 0x00020000u};
}

template <int CT, int ACTK, bool TF16>
__global__ void __launch_bounds__(512, CT == 1 ? 4 : 2) conv3_bwd_rs_kernel(const BwdParams p) {
  using T = uint16_t;
  using TT = typename std::conditional<TF16, f16_t, uint16_t>::type;
  constexpr int TZ = 4, TY = 4, TX = 16, TV = TZ * TY * TX;
  constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;
  constexpr int KPL = 8, VB = 32, NS = CT, MS = 4, STEPS = 14;
  constexpr int NPGT = HV * 2, NPG = (NPGT + 255) / 256;
  constexpr int SG_BYTES = HV * VB + 64;
  constexpr int ST_BYTES = CT * TV * VB;
  constexpr int HSTR = HX * VB;
  constexpr int NKC = TV / 32;
  constexpr int SN_BYTES = CT * 16 * 16;                                  // one wave's copy of the sample's norm records
  constexpr int QPADB = 56;
  // dgrad weights (the same 14 x NS KB for every tile): steps [0, WL) live in LDS, the last one in registers of every D wave.  (One step ahead from
  // L2, as the serial kernel does it, costs the role-split form its point: 42 requests per wave and tile keep the CU's texture path busy - the W
  // waves' DMA requests took 300 cycles each - and every step waits for its operands: dgrad steps 6.6 K cycles for 2.7 K of MFMA,
  // profiles/r06_bwd_rs_log.txt #1.)
  constexpr int WL = 13, WR = STEPS - WL;
  constexpr int SW_BYTES = WL * NS * 1024;
  // The activated tile A is NOT double-buffered (that is what makes room for the weights): only the W waves touch it, and they order themselves with
  // a counter in LDS - a W wave writes its pieces of tile i + 1 only after all four have finished the wgrad steps of tile i (wsync below).
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SG_BYTES + 3 * ST_BYTES + 8 * SN_BYTES + SW_BYTES + 16];
  unsigned char* const sG0 = smem;                                        // dy halo [2][HV][32 B]
  unsigned char* const sT0 = smem + 2 * SG_BYTES;                         // raw t tile [2][CT][TV][32 B]
  unsigned char* const sA0 = sT0 + 2 * ST_BYTES;                          // act(norm(t)) as bf16 [CT][TV][32 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;
  float* const sN = reinterpret_cast<float*>(sA0 + ST_BYTES + wave * SN_BYTES);   // [CT * 16]{mean, rstd, scale, shift}, this wave's copy
  unsigned char* const sW = sA0 + ST_BYTES + 8 * SN_BYTES;                 // [WL][NS][64 lanes][16 B]
  unsigned* const wsync = reinterpret_cast<unsigned*>(sW + SW_BYTES);      // W waves that have finished their wgrad steps, summed over tiles
  if (tid == 0) *wsync = 0u;
  __syncthreads();
  const int j = lane & 15, g = lane >> 4;
  const int D = p.D, H = p.H, W = p.W, Ct = p.Ct;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, spx = gridDim.x >> 3;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  constexpr uint32_t OOR = 0x80000000u;   // out of range of the buffers: a load returns zeros, a store is dropped
#ifdef BPX_BWD_STAMPS
  long long* stamps = (p.stamps && (tid & 255) == 0) ? p.stamps + (size_t)blockIdx.x * 16 + (wave >> 2) * 8 : nullptr;
  int stamp_i = 0;
#define BPX_STAMP() do { if (stamps && it == 4 && stamp_i < 8) stamps[stamp_i++] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define BPX_STAMP() do { } while (0)
#endif

  if (wave < 4) {
    // =========================================================== D role ===========================================================================
    const int cg_off = (g & 1) * 16;
    const bool hi_tap = (g >> 1) != 0;
    const int hb0 = ((wave * HY) * HX + j) * VB + cg_off;
    const int lbase[4] = {hb0 + (hi_tap ? VB : 0), hb0 + (hi_tap ? HX * VB : 0), hb0 + (hi_tap ? HY * HX * VB : 0), hb0};
    const int evox_rel = (wave * H) * W + j;
    const uint32_t wlane = (uint32_t)((g * Ct + j) * KPL) * 2u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wT), 0, QPADB * Ct * 16, 0x00020000);
    const uint32_t wstep = (uint32_t)(4 * Ct * 16);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.g, 0, (int)0x80000000u, 0x00020000);
    for (int q = wave; q < WL * NS; q += 4)   // LDS-resident steps: piece (s, ns) at sW + (s NS + ns) KB; visible after the first tile barrier
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sW + q * 1024), 16, wlane + (uint32_t)(q % NS) * 256u, (q / NS) * wstep, 0, 0);
    u32x4_t wres[WR][NS];
#pragma unroll
    for (int k = 0; k < WR; ++k)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        wres[k][ns] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wlane + ns * 256u), (int)((WL + k) * wstep), 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) { uint32_t v = wres[k][ns][e]; asm volatile("" : "+v"(v)); wres[k][ns][e] = v; }   // opaque: kept, not re-requested per tile
      }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's weight pieces are in LDS before it reaches the first barrier
    const int rows = (int)gridDim.x * 4, row = (int)blockIdx.x * 4 + wave;
    float ps1[NS][4], ps2[NS][4];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 4; ++r) { ps1[ns][r] = 0.f; ps2[ns][r] = 0.f; }
    auto flush_stats = [&](int nn) {   // this wave's row of sample nn: sums over its voxels of every tile of the sample so far
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(ps1[ns][r]), b = row16_sum(ps2[ns][r]);
          if (j == 0) {
            float* q = p.red + ((size_t)nn * rows + row) * 2 * Ct + ns * 16 + g * 4 + r;
            q[0] = a; q[Ct] = b;
          }
          ps1[ns][r] = 0.f; ps2[ns][r] = 0.f;
        }
    };
    int n_cur = -1, n_first = -1, it = 0;
    const uint32_t yrow = (uint32_t)(W * p.g_ld) * 2u;
    for (int local = slot; local < p.tilesPerXcd; local += spx, ++it) {
      const int tileId = xcd * p.tilesPerXcd + local;
      if (tileId >= p.totalTiles) break;
      int n, tzi, tyi, txi;
      decode_tile(tileId, p.tilesZ, p.tilesY, p.tilesX, p.tilesPerSample, p.stripY, n, tzi, tyi, txi);
      const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;
      const bool full = z0 + TZ <= D && y0 + TY <= H && x0 + TX <= W;
      const unsigned char* sG = sG0 + (it & 1) * SG_BYTES;
      const unsigned char* sT = sT0 + (it & 1) * ST_BYTES;
      if (n != n_cur) {   // wave-uniform
        if (n_cur >= 0) flush_stats(n_cur);
        if (n_cur < 0) n_first = n;
        int lo = lane;
        asm volatile("" : "+v"(lo));   // (the lane's 64-bit record address stays inside this rare branch: hoisted, it is a spilled register pair)
        if (lo < CT * 16) reinterpret_cast<f32x4_t*>(sN)[lo] = *reinterpret_cast<const f32x4_t*>(&p.t_norm[(size_t)n * Ct + lo]);
        n_cur = n;
      }
      BPX_STAMP();   // D0: tile start
      __syncthreads();   // X_i: tile i's dy halo and raw t have landed (W waves); the D waves are done with tile i - 1
      BPX_STAMP();   // D1: barrier passed
      const int vox0 = ((n * D + z0) * H + y0) * W + x0 + evox_rel;
      const bool okzx = full || (z0 + wave < D && x0 + j < W);
      const int yrem = full ? (1 << 20) : H - y0;
      const uint32_t yb0 = (uint32_t)(vox0 * p.g_ld + g * 4) * 2u;
      const unsigned char* tl = sT + ((wave * MS) * 16 + j) * VB + g * 8;
      // out-of-volume voxels of edge tiles carry no gradient: an AND with an opaque all-ones / zero mask per row.  (As `in ? x : 0` the compiler
      // moved each pair's exp2 under a branch on `in`: 24 basic blocks per tile, each ending in its own wait.)
      uint32_t mk[MS], yoff[MS];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        const bool in = okzx && ms < yrem;
        mk[ms] = in ? 0xFFFFFFFFu : 0u;
        yoff[ms] = in ? yb0 + ms * yrow : OOR;
        asm volatile("" : "+v"(mk[ms]), "+v"(yoff[ms]));
      }
      f32x4_t acc[MS][NS];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      // fragments one step ahead of the MFMAs (the compiler's own order reads a step's fragments and waits for them at once: every step paid the
      // LDS latency, 33 cycles per MFMA instead of 16)
      u32x4_t af[2][MS], wf[2][NS];
      auto load_frag = [&](int s_, u32x4_t* a_, u32x4_t* w_) {
        const int cls = s_ < 9 ? 0 : s_ < 12 ? 1 : s_ == 12 ? 2 : 3;
        const int imm = tap_off<HY, HX, VB>(bpx_tap_order_bf16(2 * s_));
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) a_[ms] = *reinterpret_cast<const u32x4_t*>(sG + lbase[cls] + ms * HSTR + imm);
        if (s_ < WL) {
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) w_[ns] = *reinterpret_cast<const u32x4_t*>(sW + (s_ * NS + ns) * 1024 + lane * 16);
        }
      };
      load_frag(0, af[0], wf[0]);
#pragma unroll
      for (int s_ = 0; s_ < STEPS; ++s_) {
        if (s_ + 1 < STEPS) load_frag(s_ + 1, af[(s_ + 1) & 1], wf[(s_ + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) {
            const u32x4_t w = s_ < WL ? wf[s_ & 1][ns] : wres[s_ - WL >= 0 ? s_ - WL : 0][ns];
            acc[ms][ns] = mfma_step<T>(w, af[s_ & 1][ms], acc[ms][ns]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      BPX_STAMP();   // D2: dgrad steps done
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        u32x2_t tv[MS];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) tv[ms] = *reinterpret_cast<const u32x2_t*>(tl + ns * TV * VB + ms * 16 * VB);
        if (ACTK == 1) {
#pragma unroll
          for (int rp = 0; rp < 4; rp += 2) {
            const f32x4_t* rsrc = reinterpret_cast<const f32x4_t*>(sN) + (ns * 16 + g * 4 + rp);
            const f32x4_t ra = rsrc[0], rb = rsrc[1];
            const f32x2_t sc2{ra[2], rb[2]}, sh2{ra[3], rb[3]}, rs2{ra[1], rb[1]}, nm2{-ra[0] * ra[1], -rb[0] * rb[1]};
            f32x2_t s1p{0.f, 0.f}, s2p{0.f, 0.f};
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) {
              const uint32_t w = tv[ms][rp >> 1];
              const f32x2_t tt{lo16<TT>(w), hi16<TT>(w)};
              const f32x2_t u = __builtin_elementwise_fma(sc2, tt, sh2);
              const f32x2_t xh = __builtin_elementwise_fma(rs2, tt, nm2);
              const f32x2_t e = u * f32x2_t{1.44269504088896341f, 1.44269504088896341f};
              f32x2_t a_{__builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(e[0]), 0.f, 1.f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(e[1]), 0.f, 1.f)};
              const f32x2_t gu = f32x2_t{acc[ms][ns][rp], acc[ms][ns][rp + 1]} * a_;
              const f32x2_t gv{__uint_as_float(__float_as_uint(gu[0]) & mk[ms]), __uint_as_float(__float_as_uint(gu[1]) & mk[ms])};
              acc[ms][ns][rp] = gv[0]; acc[ms][ns][rp + 1] = gv[1];
              s1p = s1p + gv;
              s2p = __builtin_elementwise_fma(gv, xh, s2p);
            }
            ps1[ns][rp] += s1p[0]; ps1[ns][rp + 1] += s1p[1]; ps2[ns][rp] += s2p[0]; ps2[ns][rp + 1] += s2p[1];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4_t rec = reinterpret_cast<const f32x4_t*>(sN)[ns * 16 + g * 4 + r];
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) {
              const uint32_t w = tv[ms][r >> 1];
              const float tf = (r & 1) ? hi16<TT>(w) : lo16<TT>(w);
              const float u = fmaf(rec[2], tf, rec[3]);
              const float gv = __uint_as_float(__float_as_uint(acc[ms][ns][r] * apply_act_bwd_rt<T, ACTK>(u, p.act)) & mk[ms]);
              acc[ms][ns][r] = gv;
              ps1[ns][r] += gv;
              ps2[ns][r] += gv * ((tf - rec[0]) * rec[1]);
            }
          }
        }
        // buffer stores, out-of-volume rows to an out-of-range offset: no predicate, hence no branch - the predicated form cut the epilogue into
        // 36 basic blocks of ~15 instructions, each ending in its own wait (5.4 K cycles for ~600 instructions)
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
          __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{cvt_pk_bf16(acc[ms][ns][0], acc[ms][ns][1]), cvt_pk_bf16(acc[ms][ns][2], acc[ms][ns][3])}, rs_y,
                                                (int)(yoff[ms] + ns * 32u), 0, 0);
      }
      BPX_STAMP();   // D3: epilogue stores issued
    }
    if (n_cur >= 0) flush_stats(n_cur);
    for (int nn = 0; nn < p.N; ++nn)
      if (n_cur < 0 || nn < n_first || nn > n_cur)
        for (int c = lane; c < 2 * Ct; c += 64) p.red[((size_t)nn * rows + row) * 2 * Ct + c] = 0.f;
  } else {
    // =========================================================== W role ===========================================================================
    const int wt = tid & 255, ww = __builtin_amdgcn_readfirstlane(wave) - 4;   // (an SGPR: the pieces' LDS targets are scalar arithmetic)
#if BPX_BWD_RS_WPRIO
    __builtin_amdgcn_s_setprio(BPX_BWD_RS_WPRIO);   // the W waves are the younger half of the workgroup (arbitration losers) and the longer role
#endif
    const int sub = wt & 1;
    const uint32_t dy_ld2 = (uint32_t)p.dy_ld * 2u;
    uint32_t rel_t[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int tv = (u * 256 + wt) >> 1;
      rel_t[u] = (uint32_t)((((tv >> 6) * H + ((tv >> 4) & 3)) * W + (tv & 15)) * p.t_ld + sub * KPL) * 2u;
      asm volatile("" : "+v"(rel_t[u]));
    }
    const bool last_ok = (NPG - 1) * 256 + wt < NPGT;
    const uint32_t t_csb = (uint32_t)p.t_cs * 2u;
    const u32x4_t rs_g = raw_rsrc(p.dy, 0x80000000u), rs_t = raw_rsrc(p.t, 0x80000000u);
    auto lds_off = [](const unsigned char* q) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_ptr_t)const_cast<unsigned char*>(q)); };
    // The WHOLE tile loop is instantiated per W wave (its seven taps are template constants of the MFMA phase): with the four-way switch inside the
    // loop the accumulators met in a phi behind it, the register allocator did not coalesce them - 2 x 84 accumulator registers, 84 moves per tile,
    // and spills whose reloads (scratch loads) waited for the DMA pieces in flight.
    auto w_role = [&](auto wwc) {
    constexpr int WW = decltype(wwc)::value;
    f32x4_t accw[1][7][CT];
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
      for (int c = 0; c < CT; ++c) accw[0][a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool want_b = p.want_db != 0;
    bool okt[2] = {false, false};
    int n_tile = 0;
    // requests of one tile into buffer b, the raw t tile and the dy halo by LDS-DMA (out-of-volume pieces: out-of-range offset = zeros), as NP
    // pieces: `bases` fixes the tile, `piece(q)` requests piece q (t pieces first, then the dy pieces IN ORDER - on edge tiles their halo coordinates
    // are a running counter).  One burst behind the tile barrier: a piece costs its issuer 100-370 cycles wherever it is placed (spread over the
    // K-chunks of the wgrad phase, staged through registers instead, issued by the D waves: all measured, profiles/r06_bwd_rs_log.txt #5, #7, #12).
    constexpr int NP = CT * 2 + NPG;
    uint32_t q_base_g = 0, q_base_t = 0, q_sG = 0, q_sT = 0, q_pk = 0;   // q_pk: halo coordinates of the next dy piece, hz | hy << 8 | hx << 16 (edge tiles only)
    int q_z0 = 0, q_y0 = 0, q_x0 = 0;
    bool q_interior = false, q_live = false;
    // this thread's dy pieces relative to the halo origin (piece u = halo voxel (wt >> 1) + 128 u, half wt & 1): interior tiles - two thirds of a
    // 128^3 volume - add the tile's base and are done; edge tiles also walk the halo coordinates for the bounds test
    uint32_t rel_g[NPG];
    const int hv0 = wt >> 1;
    const uint32_t hpk0 = (uint32_t)(hv0 / (HX * HY)) | ((uint32_t)((hv0 / HX) % HY) << 8) | ((uint32_t)(hv0 % HX) << 16);
#pragma unroll
    for (int u = 0; u < NPG; ++u) {
      const int hv = hv0 + u * 128, hz = hv / (HX * HY), hy = (hv / HX) % HY, hx = hv % HX;
      rel_g[u] = (uint32_t)((hz * H + hy) * W + hx) * dy_ld2 + (uint32_t)sub * 16u;
    }
    // the next tile, in two parts: `decode` (scalar divisions, ~1 K cycles of dependent latency) runs at the TOP of the iteration, beside the
    // transform's VALU work; `bases` - a few multiplies - behind the tile barrier, when the buffer is free
    int d_n = 0, d_z0 = 0, d_y0 = 0, d_x0 = 0;
    auto decode = [&](int tileId) {
      int n, tzi, tyi, txi;
      decode_tile(tileId, p.tilesZ, p.tilesY, p.tilesX, p.tilesPerSample, p.stripY, n, tzi, tyi, txi);
      d_n = n; d_z0 = tzi * TZ; d_y0 = tyi * TY; d_x0 = txi * TX;
    };
    auto bases = [&](int b) {
      const int n = d_n, z0 = d_z0, y0 = d_y0, x0 = d_x0;
      n_tile = n;
      const bool full = z0 + TZ <= D && y0 + TY <= H && x0 + TX <= W;
      q_interior = full && z0 >= 1 && z0 + TZ + 1 <= D && y0 >= 1 && y0 + TY + 1 <= H && x0 >= 1 && x0 + TX + 1 <= W;
      q_base_g = (uint32_t)(((n * D + z0 - 1) * H + (y0 - 1)) * W + (x0 - 1)) * (uint32_t)p.dy_ld * 2u;
      q_base_t = (uint32_t)(((n * D + z0) * H + y0) * W + x0) * (uint32_t)p.t_ld * 2u;
      q_sG = lds_off(sG0 + b * SG_BYTES + ww * 1024);
      q_sT = lds_off(sT0 + b * ST_BYTES + ww * 1024);
      q_z0 = z0; q_y0 = y0; q_x0 = x0;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int tv = (u * 256 + wt) >> 1;
        okt[u] = full || (z0 + (tv >> 6) < D && y0 + ((tv >> 4) & 3) < H && x0 + (tv & 15) < W);
      }
      q_pk = hpk0;
    };
    static_assert(HX * HY + HX + 2 == 128 && NPG * 128 >= HV, "piece u + 1 = piece u + 128 voxels = one plane + one row + 2");
    auto piece = [&](int q) {
      if (q < CT * 2) {
        const int c = q >> 1, u = q & 1;
        const uint32_t off = (okt[u] && q_live) ? q_base_t + rel_t[u] + (uint32_t)c * t_csb : OOR;
        dma16_asm(rs_t, q_sT + (uint32_t)(c * TV * VB + u * 4096), off);
      } else {
        const int u = q - CT * 2;
        const uint32_t inoff = q_base_g + rel_g[u];
        int ok = (int)q_live;
        if (!q_interior) {   // workgroup-uniform
          int hz = (int)(q_pk & 255u), hy = (int)((q_pk >> 8) & 255u), hx = (int)(q_pk >> 16);
          ok &= (int)((unsigned)(q_z0 - 1 + hz) < (unsigned)D) & (int)((unsigned)(q_y0 - 1 + hy) < (unsigned)H) & (int)((unsigned)(q_x0 - 1 + hx) < (unsigned)W);
          hx += 2; hy += 1; hz += 1;
          const int cx = hx >= HX; hx -= cx * HX; hy += cx;
          const int cy = hy >= HY; hy -= cy * HY; hz += cy;
          q_pk = (uint32_t)hz | ((uint32_t)hy << 8) | ((uint32_t)hx << 16);
        }
        if (u < NPG - 1) dma16_asm(rs_g, q_sG + (uint32_t)(u * 4096), ok ? inoff : OOR);
        else if (last_ok) dma16_asm(rs_g, q_sG + (uint32_t)(u * 4096), ok ? inoff : OOR);   // (the tail piece: lanes beyond the halo stay masked - they would write past the buffer)
      }
    };
    int it = 0, n_cur = -1;
    int local = slot;
    bool have = local < p.tilesPerXcd && xcd * p.tilesPerXcd + local < p.totalTiles;
    if (have) {
      decode(xcd * p.tilesPerXcd + local);
      bases(0);
      q_live = true;
#pragma unroll
      for (int q = 0; q < NP; ++q) piece(q);
    }
    for (; have; ++it) {
      const int b = it & 1;
      unsigned char* sG = sG0 + b * SG_BYTES;
      unsigned char* sT = sT0 + b * ST_BYTES;
      unsigned char* sA = sA0;
      BPX_STAMP();   // W0: tile start
      local += spx;
      const bool have_next = local < p.tilesPerXcd && xcd * p.tilesPerXcd + local < p.totalTiles;
      if (have_next) decode(xcd * p.tilesPerXcd + local);
      if (n_tile != n_cur) {   // wave-uniform: this wave's copy of the sample's norm records
        int lo = lane;
        asm volatile("" : "+v"(lo));
        if (lo < CT * 16) reinterpret_cast<f32x4_t*>(sN)[lo] = *reinterpret_cast<const f32x4_t*>(&p.t_norm[(size_t)n_tile * Ct + lo]);
        n_cur = n_tile;
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA pieces of tile i have landed
      while (__hip_atomic_load(wsync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * (unsigned)it) __builtin_amdgcn_s_sleep(1);   // A is free
      BPX_STAMP();   // W1: DMA landed, A free
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        float psc[KPL], psh[KPL];
        {
          const f32x4_t* q = reinterpret_cast<const f32x4_t*>(sN) + (c * 16 + sub * KPL);
#pragma unroll
          for (int e = 0; e < KPL; ++e) {
            const f32x4_t v = q[e];
            psc[e] = v[2]; psh[e] = v[3];
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          u32x4_t v = *reinterpret_cast<const u32x4_t*>(sT + c * TV * VB + (u * 256 + wt) * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) {   // (a select, not a branch: out-of-volume voxels of ragged tiles stay zero - the conv pads the ACTIVATED tensor)
            float a = fmaf(psc[2 * q], lo16<TT>(v[q]), psh[2 * q]), bb = fmaf(psc[2 * q + 1], hi16<TT>(v[q]), psh[2 * q + 1]);
            act_pair<ACTK>(a, bb, p.act);
            v[q] = okt[u] ? cvt_pk_bf16(a, bb) : 0u;
          }
          *reinterpret_cast<u32x4_t*>(sA + c * TV * VB + (u * 256 + wt) * 16) = v;
        }
      }
      BPX_STAMP();   // W2: transformed
      __syncthreads();   // X_i
      BPX_STAMP();   // W3: barrier passed
      have = have_next;
      int lo = lane;
      asm volatile("" : "+v"(lo));
      const int jo = lo & 15, go = lo >> 4, trl = (jo >> 2), trc = (jo & 3) * 8;
      const int a_base = go * 8 * VB + trl * VB + trc;
      const int g_lane = (((go >> 1) * HX + (go & 1) * 8) + trl) * VB + trc;
      if (have) bases(b ^ 1);
      else {           // behind the last tile the pieces are still issued, with out-of-range offsets - zeros into the FREE buffer: no branch in the wgrad phase
        q_sG = lds_off(sG0 + (b ^ 1) * SG_BYTES + ww * 1024);
        q_sT = lds_off(sT0 + (b ^ 1) * ST_BYTES + ww * 1024);
      }
      q_live = have;
      BPX_STAMP();   // W4: next tile set up
#pragma unroll
      for (int q = 0; q < NP; ++q) piece(q);
      __builtin_amdgcn_sched_barrier(0);
      BPX_STAMP();   // W5: pieces requested
      bpxwg::sd_mfma_phase<WW, CT, HY, HX, VB, VB, TV, NKC>(sA, sG, a_base, g_lane, accw[0], want_b);
      BPX_STAMP();   // W6: wgrad steps done
      if (lane == 0) __hip_atomic_fetch_add(wsync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (LDS operations of a wave execute in order: behind its last read of A)
    }
    float* pp = p.part + (size_t)blockIdx.x * 27 * Ct * 16;
#pragma unroll
    for (int a = 0; a < 7; ++a) {
      const int tap = 7 * WW + a;
      if (tap >= 27) continue;
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[((size_t)tap * Ct + c * 16 + 4 * g + r) * 16 + j] = accw[0][a][c][r];
    }
    if (want_b && WW == 3 && g == 0) p.dbpart[(size_t)blockIdx.x * 16 + j] = accw[0][6][0][0];
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): no LDS-DMA piece of this wave (the zero pieces behind the last tile) is in flight when the workgroup's LDS is released
    };
    switch (ww) {   // wave-uniform
      case 0: w_role(std::integral_constant<int, 0>{}); break;
      case 1: w_role(std::integral_constant<int, 1>{}); break;
      case 2: w_role(std::integral_constant<int, 2>{}); break;
      default: w_role(std::integral_constant<int, 3>{}); break;
    }
  }
#undef BPX_STAMP
}

int cu_count_() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

int g_bwd_fused = 1;   // test / A-B hook (bpx_debug_set_bwd_fused)

// bit 0: the (dy 16, t 48) shape takes the role-split kernel, bit 1: (dy 16, t 16) does (test / A-B hook: bpx_debug_set_bwd_rs)
#ifndef BPX_BWD_RS_DEFAULT
#define BPX_BWD_RS_DEFAULT 3
#endif
int g_bwd_rs = BPX_BWD_RS_DEFAULT;

struct BwdPlan { int cg, ct, gy, grid, tilesZ, tilesY, tilesX; bool per_wg_rows; bool rs; };
// instance for (dy.C, t.C): 16 -> {16, 48}: (CG 1, CT 1 | 3); 32 -> {16, 32, ... 128}: (CG 2, CT 1) with t.C / 16 workgroup columns.  (A (CG 2, CT 2)
// instance - two t chunks per workgroup, half the dy staging - needs 2 x 56 weight-gradient accumulators beside the rest and spilled 312 bytes per
// lane at the 256 VGPRs of two workgroups per CU, in the staging section too, where a reload serialises the DMA pieces: not built.)
bool bwd_instance(int Ct, int Cdy, BwdPlan& q) {
  q.gy = 1;
  if (Cdy == 16 && Ct == 16) { q.cg = 1; q.ct = 1; }
  else if (Cdy == 16 && Ct == 48) { q.cg = 1; q.ct = 3; }
  else if (Cdy == 32 && (Ct == 16 || Ct == 32)) { q.cg = 2; q.ct = 1; q.gy = Ct / 16; }   // (t.C = 96 measured: 457 vs 449 us for the two kernels - six
  //                                                                                          columns re-stage the dy halo six times: not taken)
  else return false;
  q.rs = q.cg == 1 && ((q.ct == 3 && (g_bwd_rs & 1)) || (q.ct == 1 && (g_bwd_rs & 2)));
  q.per_wg_rows = q.rs || !(q.cg == 1 && q.ct == 3);
  return true;
}
BwdPlan bwd_plan(int N, int D, int H, int W, int Ct, int Cdy) {
  BwdPlan q{};
  bwd_instance(Ct, Cdy, q);
  q.tilesZ = cdiv(D, 4); q.tilesY = cdiv(H, 4); q.tilesX = cdiv(W, 16);
  const int total = N * q.tilesZ * q.tilesY * q.tilesX;
  const int occ = q.rs ? (q.ct == 1 ? 2 : 1) : (q.cg == 1 && q.ct == 1) ? BPX_BWD_OCC1 : 2;   // role-split: one 8-wave workgroup per CU (two with 16 channels)
  int gx = std::max(8, (cu_count_() * occ / std::max(1, q.gy)) & ~7);    // a multiple of 8: workgroup (x, y) then runs on XCD x % 8 for every y
  gx = std::min(gx, 8 * cdiv(total, 8));
  q.grid = gx;
  return q;
}

int g_bwd_level1 = 1;   // test / A-B hook (bpx_debug_set_bwd_fused bit 1 clears it): the dy.C == 32 instances

bool bwd_supported(int dtype, int N, int D, int H, int W, int Ct, int Cdy) {
  if (!g_bwd_fused) return false;
  if (dtype != BPX_BF16 && dtype != BPX_MIX16) return false;
  BwdPlan q{};
  if (!bwd_instance(Ct, Cdy, q) || (Cdy == 32 && !g_bwd_level1)) return false;
  const int64_t vps = (int64_t)D * H * W, vox = vps * N;
  return W > 8 && vps >= 32768 && vox * std::max(48, Ct) * 2 < (1ll << 31);
}

}  // namespace

// bit 0: the fused backward at all; bit 1 set = without the dy.C == 32 instances (A/B of the level-1 layers)
extern "C" int bpx_debug_set_bwd_fused(int on) { g_bwd_fused = on & 1; g_bwd_level1 = (on & 2) ? 0 : 1; return 0; }
extern "C" int bpx_debug_set_bwd_rs(int mask) { g_bwd_rs = mask & 3; return 0; }

extern "C" int bpx_conv3d_bwd_fused_supported(int dtype, int N, int D, int H, int W, int Ct, int Cdy) { return bwd_supported(dtype, N, D, H, W, Ct, Cdy) ? 1 : 0; }

// rows per sample of red_part_d: one per WORKGROUP (column) of the persistent launch - each sums its tiles of a sample in registers - except for the
// (dy 16, t 48) instance, which writes one row per 4x4x16 tile
extern "C" int bpx_conv3d_bwd_fused_stats_tiles(int N, int D, int H, int W, int Ct, int Cdy) {
  const BwdPlan q = bwd_plan(N, D, H, W, Ct, Cdy);
  return q.rs ? 4 * q.grid : q.per_wg_rows ? q.grid : cdiv(D, 4) * cdiv(H, 4) * cdiv(W, 16);   // role-split: one row per D wave
}

extern "C" int64_t bpx_conv3d_bwd_fused_workspace(int N, int D, int H, int W, int Ct, int Cdy) {
  const BwdPlan q = bwd_plan(N, D, H, W, Ct, Cdy);
  return (int64_t)q.grid * ((int64_t)27 * Ct + 1) * Cdy * 4;
}

extern "C" int bpx_conv3d_bwd_fused(int dtype, int N, int D, int H, int W, bpx_tensor dy, const void* w_packed_T_d, bpx_tensor t,
                                    const bpx_norm_rec* t_norm_d, int act, bpx_tensor g, float* red_part_d, float* dw_d, float* db_d, float* db2_d,
                                    void* ws_d, int64_t ws_bytes, bpx_stream_t stream) {
  const char* fn = "bpx_conv3d_bwd_fused";
  BPX_CHECK(bwd_supported(dtype, N, D, H, W, t.C, dy.C), "%s: unsupported configuration (bpx_conv3d_bwd_fused_supported)", fn);
  BPX_CHECK(act >= 0 && act <= BPX_ACT_SILU, "%s: activation code %d is not built into the fused kernel (none / elu / relu / silu); use bpx_conv3d_dgrad + bpx_conv3d_wgrad", fn, act);
  BPX_CHECK(dy.ptr && t.ptr && g.ptr && w_packed_T_d && t_norm_d && red_part_d && dw_d && ws_d, "%s: null pointer", fn);
  BPX_CHECK(db2_d == nullptr || db_d != nullptr, "%s: db2_d needs db_d", fn);
  BPX_CHECK(dy.cs == 0 && g.cs == 0, "%s: only t may be chunk-planar", fn);
  BPX_CHECK(g.C == t.C && g.ld >= g.C && dy.ld >= dy.C, "%s: g.C %d != t.C %d or bad pitch", fn, g.C, t.C);
  BPX_CHECK(((uintptr_t)dy.ptr % 16) == 0 && ((uintptr_t)t.ptr % 16) == 0 && ((uintptr_t)g.ptr % 8) == 0 && (dy.ld * 2) % 16 == 0 && (t.ld * 2) % 16 == 0 &&
                (g.ld * 2) % 8 == 0 && ((uintptr_t)t_norm_d % 16) == 0,
            "%s: operands must be 16-byte aligned", fn);
  const int64_t vox = (int64_t)N * D * H * W;
  BPX_CHECK(t.cs == 0 || (t.ld >= 16 && t.cs % 8 == 0 && t.cs >= (vox - 1) * t.ld + 16), "%s: bad chunk stride %lld", fn, (long long)t.cs);
  const int64_t tbytes = 2 * (t.cs ? (int64_t)t.cs * (t.C / 16 - 1) + (vox - 1) * t.ld + 16 : vox * (int64_t)t.ld);
  BPX_CHECK(tbytes < (1ll << 31) && vox * dy.ld * 2 < (1ll << 31) && vox * g.ld * 2 < (1ll << 32), "%s: tensors beyond the 32-bit / buffer addressing range", fn);
  const BwdPlan q = bwd_plan(N, D, H, W, t.C, dy.C);
  BPX_CHECK(!q.rs || vox * g.ld * 2 < (1ll << 31), "%s: the role-split kernel stores g through a 2 GB buffer window (pitch %d over %lld voxels is beyond it); "
            "clear the shape's bit with bpx_debug_set_bwd_rs before sizing the workspace", fn, g.ld, (long long)vox);
  const int64_t need = (int64_t)q.grid * ((int64_t)27 * t.C + 1) * dy.C * 4;
  BPX_CHECK(ws_bytes >= need, "%s: workspace too small (%lld < %lld bytes)", fn, (long long)ws_bytes, (long long)need);
  BwdParams p{};
  p.N = N; p.D = D; p.H = H; p.W = W;
  p.dy = dy.ptr; p.dy_ld = dy.ld;
  p.wT = w_packed_T_d;
  p.t = t.ptr; p.t_ld = t.ld; p.t_cs = t.cs ? (int)t.cs : 16; p.Ct = t.C;
  p.t_norm = t_norm_d; p.act = act;
  p.g = g.ptr; p.g_ld = g.ld;
  p.red = red_part_d;
  p.part = reinterpret_cast<float*>(ws_d);
  p.dbpart = p.part + (size_t)q.grid * 27 * t.C * dy.C;
  p.want_db = db_d != nullptr;
  p.tilesZ = q.tilesZ; p.tilesY = q.tilesY; p.tilesX = q.tilesX;
  p.tilesPerSample = q.tilesZ * q.tilesY * q.tilesX;
  p.totalTiles = N * p.tilesPerSample;
  p.tilesPerXcd = cdiv(p.totalTiles, 8);
  p.stripY = strip_rows(q.tilesX);
  p.stamps = g_conv_stamps;
  const bool mix = dtype == BPX_MIX16, elu = act == BPX_ACT_ELU;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)q.grid, (unsigned)q.gy);
#define LB(CG_, CT_)                                                                                  \
  if (q.cg == CG_ && q.ct == CT_) {                                                                   \
    if (mix) { if (elu) conv3_bwd_kernel<CG_, CT_, 1, true><<<grid, 256, 0, s>>>(p); else conv3_bwd_kernel<CG_, CT_, 0, true><<<grid, 256, 0, s>>>(p); }   \
    else { if (elu) conv3_bwd_kernel<CG_, CT_, 1, false><<<grid, 256, 0, s>>>(p); else conv3_bwd_kernel<CG_, CT_, 0, false><<<grid, 256, 0, s>>>(p); }     \
  }
  if (q.rs) {
#define LR(CT_)                                                                                      \
    if (q.ct == CT_) {                                                                                \
      if (mix) { if (elu) conv3_bwd_rs_kernel<CT_, 1, true><<<grid, 512, 0, s>>>(p); else conv3_bwd_rs_kernel<CT_, 0, true><<<grid, 512, 0, s>>>(p); }   \
      else { if (elu) conv3_bwd_rs_kernel<CT_, 1, false><<<grid, 512, 0, s>>>(p); else conv3_bwd_rs_kernel<CT_, 0, false><<<grid, 512, 0, s>>>(p); }     \
    }
    LR(1) LR(3)
#undef LR
  } else {
    LB(1, 1) LB(1, 3) LB(2, 1)
  }
#undef LB
  BPX_LAUNCH_CHECK(fn);
  // dW in the PyTorch layout (Cout = dy.C, Cin = Ct, 3, 3, 3): index = ci * 27 + co * Ct * 27 + tap
  return bpxred::reduce_partials2(fn, p.part, dw_d, q.grid, 27, t.C, dy.C, 27, (int64_t)t.C * 27, 1, p.dbpart, db_d, db2_d, 0, true, s);
}
