// Conv3d 3x3x3 forward, 16-bit storage, 16 output channels: the Z-MARCHING (plane-ring) schedule of the level-0 layers (round 5).
//
// Same GEMM mapping, packed-weight order, tap order, epilogue and statistics layout as conv3_lp_kernel<4, 8, 16, 1, EPI_FWD, ...> (conv3d_lean.hip -
// read its header first): every accumulator sees the same MFMAs in the same order and every statistics row is the same sum, so the output, the
// fused pool and the partial sums are BIT-IDENTICAL to the lean kernel's (tests: check_conv3d_zmarch).  What differs is how the activated input
// gets into LDS:
//   * the lean kernel stages the whole (4+2) x (8+2) x (16+2) halo of every tile: each input voxel is fetched, normalised, activated, converted and
//     written to LDS 2.11 times, and the cycle stamps put ~70 % of a tile's time into that staging (profiles/r04_stamps_fwd_dgrad.txt);
//   * here a persistent workgroup owns a run of consecutive z-steps of ONE (y, x) tile column.  The halo's six z-planes live in a RING: a step
//     stages only its 4 NEW planes, the two overlap planes of the previous step stay where they are (1.41 x per voxel instead of 2.11 x; a run
//     pays the two extra planes once, at its start and where it crosses into the next column);
//   * a step's new planes are fewer pieces per thread (6 instead of 9), so the NEXT stage's pieces fit in registers while this stage's MFMA steps
//     run: the global-load latency that the lean kernel exposes in every chunk stage is off the critical path (the lean kernel's own prefetch
//     experiment failed on registers: 9 pieces + 168-VGPR budget, conv3d_lean.hip header);
//   * plane slots are addressed as (uniform slot base) + (per-lane constant) + immediate: the ring costs eight v_add per stage, no unrolled phases.
// LDS, one 16-channel input chunk (NCH = 1): a ring of 6 plane slots, ring-relative plane q of the step sits in slot (q + off) % 6, off += 4 per step.
// Three chunks (NCH = 3, the decoder's 48 -> 16): all six planes of all chunks would need 104 KB, so per chunk only what must SURVIVE the step is
// private - planes {0, 1} (from the previous step) and {4, 5} (for the next one) in two alternating pairs - and the planes {2, 3}, which no later
// stage reads, share one region W: 2 + 4 * 3 = 14 planes = 81 KB, two workgroups per CU (the lean kernel has three, without any overlap of loads
// and MFMAs inside a workgroup).
#include "conv3d_shared.h"

using namespace bpxconv;

namespace {

#ifndef BPX_ZM_IMG_EARLY
#define BPX_ZM_IMG_EARLY 1
#endif

#ifndef BPX_ZM_RH
#define BPX_ZM_RH 4
#endif
#ifndef BPX_ZM_SCK_AH
#define BPX_ZM_SCK_AH 2
#endif
#ifndef BPX_ZM_SCK_RH
#define BPX_ZM_SCK_RH 2
#endif
constexpr int zm_occ(int nch, bool sck = false) { return (nch == 1 && !sck) ? 3 : 2; }

// SCK: the launch has a fused 1x1x1 shortcut on a raw tensor of 16 .. 48 channels (the decoder's second conv: its 48-channel operand is three times
// the conv's own input).  Its own instance at TWO workgroups per CU: the whole operand of the step - up to 96 VGPRs - is requested behind the prefetch
// and arrives during the MFMA steps (at three per CU there is no room: 144-228 bytes of scratch in every form that was tried), so a workgroup has
// 23 + 48 KB in flight instead of 23 and the shortcut's K steps wait for nothing.
template <int NCH, int ACTK, bool F16, bool SCK = false>
__global__ void __launch_bounds__(256, zm_occ(NCH, SCK)) conv3_zm_kernel(const Conv3Params p) {
  using T = typename std::conditional<F16, f16_t, uint16_t>::type;
  constexpr int TZ = 4, TY = 8, TX = 16, MS = 8, KPL = 8, VB = 32, HY = TY + 2, HX = TX + 2;
  constexpr int PLANE_B = HY * HX * VB;                              // 5760 bytes: one z-plane of a chunk's halo, 32 bytes per voxel
  constexpr int PP = HY * HX * 2;                                     // 360 16-byte pieces per plane
  constexpr int PL = PP - 256;                                        // 104 pieces of a plane beyond one per thread
  static_assert(2 * PL <= 256, "a plane pair's left-over pieces fit one round");
  constexpr int STEPS = 14, QPAD = 56, HSTR = HX * VB;
  constexpr int RH = NCH != 1 ? MS : SCK ? BPX_ZM_SCK_RH : BPX_ZM_RH;                       // m-subtiles per fragment-row batch of the (dz, dy) steps
  constexpr int PLANES = NCH == 1 ? 6 : 2 + 4 * NCH;
  constexpr int RING_B = PLANES * PLANE_B;
  // Weights.  A wave's VMEM operations retire IN ORDER through one counter: a weight fragment requested after the next stage's prefetch could only
  // be waited for together with that prefetch (the first build did exactly that: `s_waitcnt vmcnt(1)` in front of the first MFMA drained the six
  // prefetch loads, i.e. exposed the HBM latency the prefetch exists to hide).  So no weight load may follow the prefetch inside a stage:
  //   NCH == 1: the conv's 14 KB of packed weights are LDS-resident for the whole run (lane-linear [step][lane][16 B] = the packed order itself);
  //   NCH >  1: the stage's 14 fragments are requested into registers at the top of the stage, BEFORE the prefetch (56 VGPRs of the 256 that two
  //             workgroups per CU leave).
  constexpr int WLDS_B = NCH == 1 ? STEPS * 1024 : 0;
  constexpr int RED_B = (NCH == 1 ? 2 : 1) * 4 * 16 * 2 * 4;           // statistics scratch [which][wave][16][2] floats (the pooled tensor's: NCH == 1 only)
  constexpr int TAB_B = NCH * 16 * 8 + 128;                           // {scale, shift} of the column's sample, every input channel; + {bias sum, rank-1 weight} per output channel
  // z-pair exchange of the fused max-pool: packed 16-bit values of the even-x lanes only (after the x-pair maximum lanes j and j ^ 1 hold the same
  // value) = 2 KB.  Every byte counts here: LDS is granted in 2 KB granules, three workgroups per CU need <= 53,248 bytes each, and the first build
  // (54,272) silently ran TWO per CU - found with the workgroup-count sweep of profiles/r05_zmarch_ab.txt, not by the compiler's occupancy remark
  constexpr int POOL_B = NCH == 1 ? 2 * (MS / 2) * 32 * 8 : 0;
  constexpr int WOFF = RING_B, ROFF = WOFF + WLDS_B, TOFF = ROFF + RED_B, POFF = TOFF + TAB_B;
  __shared__ __attribute__((aligned(16))) unsigned char smem[POFF + POOL_B];      // ONE shared object (a second one costs vmcnt(0) waits before LDS reads)
  float* const red = reinterpret_cast<float*>(smem + ROFF);
  float* const tab = reinterpret_cast<float*>(smem + TOFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int D = p.D, H = p.H, W = p.W;
  constexpr int Cout = 16;
  const int cg_off = (g & 1) * 16;
  const bool hi_tap = (g >> 1) != 0;
  const uint32_t hb = (uint32_t)(j * VB + cg_off);                    // this lane's fragment row 0 inside a plane: voxel (row 0, x = j), channel half g & 1

  // profiling: cycle stamps of this workgroup's 5th step (steady state: ring and prefetch running), scripts/zm_stamps.py
#ifdef BPX_ZM_STAMPS   // profiling build only (bash scripts/ab_build_flags.sh zmstamps -DBPX_ZM_STAMPS): the bookkeeping costs the production kernel registers it does not have
  long long* stamps = (p.stamps && tid == 0) ? p.stamps + (size_t)blockIdx.x * 16 : nullptr;
  int stamp_i = 0, it = 0;
#define ZM_STAMP() do { if (stamps && it == 4 && stamp_i < 15) stamps[stamp_i++] = (long long)__builtin_readcyclecounter(); } while (0)
#define ZM_STEP_DONE() ++it
#else
#define ZM_STAMP() do { } while (0)
#define ZM_STEP_DONE() do { } while (0)
#endif

  // ---- staging constants: piece tid of a plane ("A"), and the pair's left-over piece ("L": plane selL of the pair, piece 256 + tid % 104) ----
  const int sub = tid & 1;                                            // 104 is even: both pieces of a thread cover the same 8 channels of the chunk
  const int hvA = tid >> 1, hyA = hvA / HX, hxA = hvA - hyA * HX;
  const int selL = tid / PL, idxL = 256 + (tid - selL * PL), hvL = idxL >> 1, hyL = hvL / HX, hxL = hvL - hyL * HX;
  const bool hasL = tid < 2 * PL;
  const uint32_t HWB = (uint32_t)(H * W * p.x_ld) * 2u;                // bytes between z-planes of the input
  uint32_t relA = (uint32_t)((hyA * W + hxA) * p.x_ld + sub * KPL) * 2u;
  uint32_t relL = (uint32_t)((hyL * W + hxL) * p.x_ld + sub * KPL) * 2u + (hasL ? (uint32_t)selL * HWB : 0u);
  asm volatile("" : "+v"(relA), "+v"(relL));
  const uint32_t ldsA = (uint32_t)tid * 16u, ldsL = (uint32_t)idxL * 16u;
  const char* __restrict__ xin = reinterpret_cast<const char*>(p.x);
  const char* __restrict__ wp = reinterpret_cast<const char*>(p.wp);
  const uint32_t wlane = (uint32_t)((g * Cout + j) * KPL) * 2u;        // this lane's 16-byte operand inside a [4][Cout][8] k-group block
  const uint32_t x_csb = (uint32_t)p.x_cs * 2u, sc_csb = (uint32_t)p.sc_cs * 2u;

  // bias (+ shortcut bias) and the rank-1 shortcut weights: per-workgroup constants, parked in LDS (the epilogue reads its lane's four channels
  // with two ds_read_b128: no registers held across the MFMA steps, no L2 latency per step)
  const bool rank1 = p.sc != nullptr && p.sc_C == 1;
  float* const ektab = tab + NCH * 16 * 2;                            // [16] bias sums, [16] rank-1 weights
  if (tid < 16) {
    float a = 0.f;
    if (p.bias) a += p.bias[tid];
    if (p.sc && p.bias_sc) a += p.bias_sc[tid];
    ektab[tid] = a;
    ektab[16 + tid] = rank1 ? reinterpret_cast<const float*>(p.wsc)[tid] : 0.f;
  }
  if constexpr (NCH == 1) {
#pragma unroll
    for (int q = tid; q < STEPS * 64; q += 256)
      *reinterpret_cast<u32x4_t*>(smem + WOFF + q * 16) = *reinterpret_cast<const u32x4_t*>(wp + (uint32_t)q * 16u);
  }
  // (both visible to every wave after the first step's staging barrier)

  // ---- this workgroup's run of z-steps: XCD x owns the id range [x T/8, (x+1) T/8), its workgroups consecutive pieces of it; ids enumerate
  //      (column = (n, tile row, tile x), z-step) with the z-step fastest ------------------------------------------------------------------------
  const int tilesZ = p.tilesZ, colsPerSample = p.tilesY * p.tilesX;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, spx = gridDim.x >> 3;
  const int T8 = p.tilesPerXcd;
  int L0 = xcd * T8 + (int)(((long long)slot * T8) / spx), L1 = xcd * T8 + (int)(((long long)(slot + 1) * T8) / spx);
  {
    const int cap = min((xcd + 1) * T8, p.totalTiles);
    L1 = min(L1, cap);
  }
  if (L0 >= L1) return;
  int col = L0 / tilesZ, tzi = L0 - col * tilesZ;

  // column state
  int n = 0, y0 = 0, x0 = 0, tyi = 0, txi = 0;
  uint32_t colb = 0;                       // byte offset of halo voxel (z = 0, y0 - 1, x0 - 1) of sample n (may wrap below zero: only in-volume pieces are dereferenced)
  bool okA = false, okL = false;
  int off = 0;                             // NCH == 1: ring offset;  NCH > 1: parity of the private plane pairs
  bool newcol = true;

  // prefetched pieces of the next stage's new planes: pair 0 = planes {2, 3}, pair 1 = planes {4, 5}; three pieces each (A of both planes, L)
  u32x4_t pbuf[2][3];
  uint32_t pmask = 0;

  // slot base (bytes) of ring-relative plane q of chunk c in the CURRENT step
  auto sbase = [&](int q, int c) -> uint32_t {
    if constexpr (NCH == 1) {
      int s = q + off;
      s = s >= 6 ? s - 6 : s;
      return (uint32_t)s * PLANE_B;
    } else {
      if (q == 2 || q == 3) return (uint32_t)(q - 2) * PLANE_B;
      const int par = q < 2 ? off : (off ^ 1);
      return (uint32_t)(2 + (c * 2 + par) * 2 + (q & 1)) * PLANE_B;
    }
  };

  // request the three pieces of the plane pair whose first plane is volume slice z (z, z + 1), chunk c.  Out-of-volume pieces load the tensor's first
  // 16 bytes instead of branching around the load (a branch per load costs the prefetch its batching) and are zeroed when they are stored.
  auto load_pair = [&](u32x4_t* v, uint32_t& mask, int bit, int z, int c) {
    const uint32_t cb = colb + (uint32_t)c * x_csb;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool ok = okA && (unsigned)(z + k) < (unsigned)D;
      v[k] = *reinterpret_cast<const u32x4_t*>(xin + (ok ? cb + (uint32_t)(z + k) * HWB + relA : 0u));
      mask |= ok ? 1u << (bit + k) : 0u;
    }
    const bool ok = okL && (unsigned)(z + selL) < (unsigned)D;
    v[2] = *reinterpret_cast<const u32x4_t*>(xin + (ok ? cb + (uint32_t)z * HWB + relL : 0u));
    mask |= ok ? 1u << (bit + 2) : 0u;
  };

  const bool has_norm = p.in_norm != nullptr;
  // normalise + activate (in-volume pieces only: the zero padding applies to the ACTIVATED tensor) and write a pair's pieces to their plane slots
  auto store_pair = [&](const u32x4_t* v, uint32_t mask, int bit, uint32_t s0, uint32_t s1, const float* psc, const float* psh) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k == 2 && !hasL) break;
      u32x4_t t = v[k];
      const bool in = (mask >> (bit + k)) & 1u;
      if (!in) t = u32x4_t{0u, 0u, 0u, 0u};
      if (has_norm && in) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float a = fmaf(psc[2 * i], lo16<T>(t[i]), psh[2 * i]), b = fmaf(psc[2 * i + 1], hi16<T>(t[i]), psh[2 * i + 1]);
          act_pair<ACTK>(a, b, p.act);
          t[i] = pk16<T>(a, b);
        }
      }
      const uint32_t dst = k == 0 ? s0 + ldsA : k == 1 ? s1 + ldsA : (selL ? s1 : s0) + ldsL;
      *reinterpret_cast<u32x4_t*>(smem + dst) = t;
    }
  };
  auto load_tab = [&](int c, float* psc, float* psh) {
    if (!has_norm) return;
    const f32x4_t* tp = reinterpret_cast<const f32x4_t*>(tab + (c * 16 + sub * KPL) * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4_t q = tp[i];
      psc[2 * i] = q[0]; psh[2 * i] = q[1]; psc[2 * i + 1] = q[2]; psh[2 * i + 1] = q[3];
    }
  };

  char* __restrict__ yout = reinterpret_cast<char*>(p.y);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)0xFFFFFFF0u, 0x00020000);   // (epilogue rows: out-of-volume rows store beyond it)
  const uint32_t yrow = (uint32_t)(W * p.y_ld) * 2u;

  for (int L = L0; L < L1; ++L) {
    if (newcol) {
      n = col / colsPerSample;
      const int r = col - n * colsPerSample;
      tyi = r / p.tilesX; txi = r - tyi * p.tilesX;
      y0 = tyi * TY; x0 = txi * TX;
      colb = (uint32_t)(((n * D) * H + (y0 - 1)) * W + (x0 - 1)) * (uint32_t)p.x_ld * 2u;
      okA = (unsigned)(y0 - 1 + hyA) < (unsigned)H && (unsigned)(x0 - 1 + hxA) < (unsigned)W;
      okL = hasL && (unsigned)(y0 - 1 + hyL) < (unsigned)H && (unsigned)(x0 - 1 + hxL) < (unsigned)W;
      off = 0;
    }
    ZM_STAMP();   // 0: step start
    const int z0 = tzi * TZ;
    const bool more = (L + 1 < L1) && (tzi + 1 < tilesZ);             // the next step continues this column: its new planes can be prefetched

    f32x4_t acc[MS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) acc[ms] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float img[MS];
    u32x4_t bq[SCK ? 3 : 1][SCK ? MS : 1], bw[SCK ? 3 : 1];
    const int sck_n = SCK ? p.sc_C / 16 : 0;

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // NCH > 1: this stage's 14 weight fragments, requested before anything else of the stage (behind them in the VMEM queue: only the pieces
      // prefetched during the previous stage, which the stage needs first anyway)
      u32x4_t wall[NCH == 1 ? 1 : STEPS];
      if constexpr (NCH > 1) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
          uint32_t so = (uint32_t)((c * QPAD + s * 4) * Cout * 16);
          asm volatile("" : "+s"(so));                                 // opaque and scalar: (uniform pointer + SGPR) + this lane's 32-bit offset, not a hoisted 64-bit VGPR address per step
          wall[s] = *reinterpret_cast<const u32x4_t*>(wp + so + wlane);
        }
      }
      if (c > 0) __syncthreads();                                     // the shared planes {2, 3} of the previous chunk are no longer read
      float psc[KPL], psh[KPL];
      if (newcol) {
        // ---- a run's first step in a column: all six planes, nothing prefetched (once per ~20-30 steps) ----
        if (c == 0 && has_norm && tid < NCH * 16) {
          const f32x2_t ss = *reinterpret_cast<const f32x2_t*>(&p.in_norm[(size_t)n * p.Cin + tid].scale);
          *reinterpret_cast<f32x2_t*>(tab + tid * 2) = ss;
        }
        u32x4_t v0[3];
        uint32_t m0 = 0;
        pmask = 0;
        load_pair(v0, m0, 0, z0 - 1, c);
        load_pair(pbuf[0], pmask, 0, z0 + 1, c);
        load_pair(pbuf[1], pmask, 3, z0 + 3, c);
        if (c == 0 && has_norm) __syncthreads();                      // the table is visible
        load_tab(c, psc, psh);
        store_pair(v0, m0, 0, sbase(0, c), sbase(1, c), psc, psh);
      } else {
        load_tab(c, psc, psh);
      }
#ifdef BPX_ZM_SPLIT    // profiling build: an explicit wait for the prefetched pieces (the BPX_ZM_SPLIT stores issued after them may stay outstanding) + a stamp
      if (c == 0) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BPX_ZM_SPLIT) : "memory"); ZM_STAMP(); }
#endif
      store_pair(pbuf[0], pmask, 0, sbase(2, c), sbase(3, c), psc, psh);
      store_pair(pbuf[1], pmask, 3, sbase(4, c), sbase(5, c), psc, psh);
      if (c == 0) ZM_STAMP();   // 1: chunk 0 transformed and written
      __syncthreads();                                                // this stage's planes are in LDS
      if (c == 0) ZM_STAMP();   // 2: barrier
      // ---- the next stage's new planes fly while this stage's MFMA steps run (nothing the MFMA steps wait for is requested after them) ----
      pmask = 0;
      if (!newcol && c + 1 < NCH) {
        load_pair(pbuf[0], pmask, 0, z0 + 1, c + 1);
        load_pair(pbuf[1], pmask, 3, z0 + 3, c + 1);
      } else if (c + 1 == NCH && more) {
        load_pair(pbuf[0], pmask, 0, z0 + TZ + 1, 0);
        load_pair(pbuf[1], pmask, 3, z0 + TZ + 3, 0);
      }
      // SCK: request chunk q of the shortcut operand (8 m-subtiles x 16 bytes per lane) and its weight fragment.  Out-of-volume lanes read the
      // tensor's first bytes: their accumulator columns are never stored or summed.
      auto sc_request = [&](int q) {
        if constexpr (SCK) {
          // (no per-load control flow: a chunk beyond the operand's last one re-reads the last chunk and is skipped at the MFMAs)
          const int qq = q < sck_n ? q : sck_n - 1;
          const int vox0 = ((n * D + z0 + wave) * H + y0) * W + x0 + j;
          const bool okzx = (z0 + wave < D) & (x0 + j < W);
          const uint32_t sb0 = (uint32_t)(vox0 * p.sc_ld) * 2u + (uint32_t)cg_off + (uint32_t)qq * sc_csb, srow = (uint32_t)(W * p.sc_ld) * 2u;
#pragma unroll
          for (int ms = 0; ms < MS; ++ms) {
            const uint32_t o = (okzx & (y0 + ms < H)) ? sb0 + ms * srow : 0u;
            bq[q][ms] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(p.sc) + o);
          }
          bw[q] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(p.wsc) + (uint32_t)(qq * 4 * Cout * 16) + wlane);
        }
      };
      if (SCK && c + 1 == NCH) { sc_request(0); sc_request(1); }        // chunks 0 and 1 behind the prefetch; chunk 2 in front of the last five steps (below)
      if (!SCK && BPX_ZM_IMG_EARLY && c + 1 == NCH && rank1) {
        // rank-1 shortcut: the image value of this lane's eight voxels, requested behind the prefetch (8 VGPRs across the MFMA steps).  Measured and
        // dropped (profiles/r05_zmarch_ab.txt): two requests per lane handed round with v_permlane16/32_swap - the same time, and in combination with
        // the 16-byte stores below a few rows came out wrong, differently from run to run (an unexplained permlane hazard: scripts/probes/zm_diff.py)
        const int vox0 = ((n * D + z0 + wave) * H + y0) * W + x0 + j;
        const bool okzx = z0 + wave < D && x0 + j < W;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
          img[ms] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.sc) + ((okzx && y0 + ms < H) ? (uint32_t)(vox0 + ms * W) * 4u : 0u));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (c == 0) ZM_STAMP();   // 3: prefetch requested

      // ---- 14 MFMA steps on planes wave .. wave + 2 of the ring ----
      const uint32_t pz[3] = {sbase(wave, c), sbase(wave + 1, c), sbase(wave + 2, c)};   // uniform
      const uint32_t hb0 = hb + (hi_tap ? VB : 0), hb1 = hb + (hi_tap ? HX * VB : 0);
      const uint32_t wl = WOFF + (uint32_t)lane * 16u;
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        // fragment rows ms + dy of plane dz: read once per plane and slid over the three dy steps (conv3d_lean.hip REUSE); RH m-subtiles at a time
        const uint32_t b0 = pz[dz] + hb0;
        u32x4_t w3[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
          w3[dy] = NCH == 1 ? *reinterpret_cast<const u32x4_t*>(smem + wl + (3 * dz + dy) * 1024) : wall[NCH == 1 ? 0 : 3 * dz + dy];
#pragma unroll
        for (int h = 0; h < MS; h += RH) {
          u32x4_t row[RH + 2];
#pragma unroll
          for (int r = 0; r < RH + 2; ++r) row[r] = *reinterpret_cast<const u32x4_t*>(smem + b0 + (h + r) * HSTR);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int ms = 0; ms < RH; ++ms) acc[h + ms] = mfma_step<T>(w3[dy], row[ms + dy], acc[h + ms]);
        }
      }
      if (SCK && c + 1 == NCH) {       // the fragment-row registers of the (dz, dy) steps are free from here on
        __builtin_amdgcn_sched_barrier(0);
        sc_request(2);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 9; s < STEPS; ++s) {
        // steps 9-11: taps (dz, 0, 2) + (dz, 1, 2); step 12: taps (0, 2, 2) + (1, 2, 2); step 13: tap (2, 2, 2) alone
        const uint32_t base = s < 12 ? pz[s < 12 ? s - 9 : 0] + hb1 : s == 12 ? (hi_tap ? pz[1] : pz[0]) + hb : pz[2] + hb;
        const int imm = s < 12 ? 2 * VB : (2 * HX + 2) * VB;
        const u32x4_t ws = NCH == 1 ? *reinterpret_cast<const u32x4_t*>(smem + wl + s * 1024) : wall[NCH == 1 ? 0 : s];
        constexpr int AH = SCK ? BPX_ZM_SCK_AH : MS;                   // m-subtiles per batch (the wide-shortcut instance has its operand live here)
#pragma unroll
        for (int h = 0; h < MS; h += AH) {
          u32x4_t af[AH];
#pragma unroll
          for (int ms = 0; ms < AH; ++ms) af[ms] = *reinterpret_cast<const u32x4_t*>(smem + base + (h + ms) * HSTR + imm);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ms = 0; ms < AH; ++ms) acc[h + ms] = mfma_step<T>(ws, af[ms], acc[h + ms]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (c == 0) ZM_STAMP();   // 4: chunk 0 MFMA steps
    }
    __builtin_amdgcn_sched_barrier(0);
    ZM_STAMP();   // 5: all chunks

    // ---- fused 1x1x1 shortcut on a second raw tensor: extra K steps, operand straight from global memory (as the lean kernel) ----
    const bool full = z0 + TZ <= D && y0 + TY <= H && x0 + TX <= W;
    const int vox0 = ((n * D + z0 + wave) * H + y0) * W + x0 + j;     // this lane's voxel for m-subtile 0
    const bool okzx = full || (z0 + wave < D && x0 + j < W);
    const int yrem = full ? (1 << 20) : H - y0;                        // m-subtile ms is inside the volume iff ms < yrem
    if constexpr (SCK) {
      // the operand arrived during the MFMA steps; per accumulator the chunks are added in order 0, 1, 2 after the conv's own steps - the lean kernel's bits
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (q < sck_n) {
#pragma unroll
          for (int ms = 0; ms < MS; ++ms) acc[ms] = mfma_step<T>(bw[q], bq[q][ms], acc[ms]);
        }
      }
    }
    ZM_STAMP();   // 6: wide shortcut
    // ---- epilogue (the lean kernel's, one output-channel group): bias / rank-1 shortcut, statistics, one 8-byte store per m-subtile ----
    const uint32_t yb0 = (uint32_t)(vox0 * p.y_ld + g * 4) * 2u;
    if (!(BPX_ZM_IMG_EARLY) || !rank1) {
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
        img[ms] = (rank1 && okzx && ms < yrem) ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.sc) + (uint32_t)(vox0 + ms * W) * 4u) : 0.f;
    }
    const f32x4_t addk = *reinterpret_cast<const f32x4_t*>(ektab + g * 4), w1k = *reinterpret_cast<const f32x4_t*>(ektab + 16 + g * 4);
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) + lgkmcnt(0) here, outside the predicated row blocks (conv3d_lean.hip: stores must not wait for stores)
    auto flush_stats = [&](const float* s1, const float* s2, int which) {
      if ((which ? (float*)p.pool_part : p.part) == nullptr) return;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[r]), b = row16_sum(s2[r]);
        if (j == 0) *reinterpret_cast<f32x2_t*>(&red[which * 4 * 16 * 2 + (wave * 16 + g * 4 + r) * 2]) = f32x2_t{a, b};
      }
    };
    {
      const int co = g * 4;
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
      u32x2_t pk[MS];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        // branch-free rows (round 6): out-of-volume rows contribute zeros and store to an offset beyond the buffer.  (Not the one-chunk instance without a
        // wide shortcut - the A/B partner of conv3_zs_kernel: at its 168 registers of three workgroups per CU the straight-line form spills 36 bytes.)
        constexpr bool BFREE = NCH == 3 || SCK;
        const bool in = okzx && ms < yrem;
        pk[ms] = u32x2_t{0u, 0u};
        if (BFREE || in) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = acc[ms][r] + addk[r] + img[ms] * w1k[r];
            v[r] = (!BFREE || in) ? t : 0.f;
            s1[r] += v[r];
            s2[r] += v[r] * v[r];
          }
          pk[ms] = u32x2_t{pk16s<T>(v[0], v[1]), pk16s<T>(v[2], v[3])};
          if constexpr (BFREE) __builtin_amdgcn_raw_buffer_store_b64(pk[ms], rs_y, (int)(in ? yb0 + ms * yrow : 0xFFFFFFFFu), 0, 0);
          else *reinterpret_cast<u32x2_t*>(yout + (yb0 + ms * yrow)) = pk[ms];
        }
      }
      // (Measured and removed: 16-byte stores - lanes (j, g) and (j, g ^ 1) exchange their 8-byte halves with v_permlane16_swap, 4 stores per lane
      //  instead of 8 - bit-identical on their own and FLAT: 236.0 / 385.2 / 625.9 us with, 235.2 / 383.6 / 627.8 without, profiles/r05_zmarch_ab.txt.)
      if constexpr (NCH == 1) {
        if (p.pool != nullptr) {
          // fused MaxPool3d (pool_sz, 2, 2): y pairs = two m-subtiles of this lane, x pairs = lanes j / j^1 (DPP), z pairs = waves w / w + 1 (LDS)
          float m[MS / 2][4];
#pragma unroll
          for (int k = 0; k < MS / 2; ++k) {
            const u32x2_t a = pk[2 * k], b = pk[2 * k + 1];
            m[k][0] = fmaxf(lo16<T>(a[0]), lo16<T>(b[0])); m[k][1] = fmaxf(hi16<T>(a[0]), hi16<T>(b[0]));
            m[k][2] = fmaxf(lo16<T>(a[1]), lo16<T>(b[1])); m[k][3] = fmaxf(hi16<T>(a[1]), hi16<T>(b[1]));
#pragma unroll
            for (int r = 0; r < 4; ++r)
              m[k][r] = fmaxf(m[k][r], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m[k][r]), 0xB1, 0xF, 0xF, true)));
          }
          if (p.pool_sz == 2) {
            // [wave pair][k][lane / 2], as packed 16-bit pairs: every m is a maximum of stored 16-bit values, so the round trip is exact; lanes j and
            // j ^ 1 write the same value to the same slot.  The region is this exchange's alone and was last read before the previous step's closing barrier.
            u32x2_t* ex = reinterpret_cast<u32x2_t*>(smem + POFF);
            if (wave & 1) {
#pragma unroll
              for (int k = 0; k < MS / 2; ++k) ex[((wave >> 1) * (MS / 2) + k) * 32 + (lane >> 1)] = u32x2_t{pk16<T>(m[k][0], m[k][1]), pk16<T>(m[k][2], m[k][3])};
            }
            __syncthreads();
            if (!(wave & 1)) {
#pragma unroll
              for (int k = 0; k < MS / 2; ++k) {
                const u32x2_t o = ex[((wave >> 1) * (MS / 2) + k) * 32 + (lane >> 1)];
                m[k][0] = fmaxf(m[k][0], lo16<T>(o[0])); m[k][1] = fmaxf(m[k][1], hi16<T>(o[0]));
                m[k][2] = fmaxf(m[k][2], lo16<T>(o[1])); m[k][3] = fmaxf(m[k][3], hi16<T>(o[1]));
              }
            }
          }
          float q1[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};
          if ((p.pool_sz == 1 || !(wave & 1)) && !(j & 1) && z0 + wave < D && x0 + j < W) {
            const int Dp = D / p.pool_sz, Hp = H >> 1, Wp = W >> 1;
            const int pz = (z0 + wave) / p.pool_sz, px = (x0 + j) >> 1;
            char* __restrict__ pout = reinterpret_cast<char*>(p.pool);
#pragma unroll
            for (int k = 0; k < MS / 2; ++k) {
              if (y0 + 2 * k < H) {
                const int py = (y0 >> 1) + k;
                *reinterpret_cast<u32x2_t*>(pout + (uint32_t)((((n * Dp + pz) * Hp + py) * Wp + px) * p.pool_ld + co) * 2u) =
                    u32x2_t{pk16<T>(m[k][0], m[k][1]), pk16<T>(m[k][2], m[k][3])};
#pragma unroll
                for (int r = 0; r < 4; ++r) { q1[r] += m[k][r]; q2[r] += m[k][r] * m[k][r]; }
              }
            }
          }
          flush_stats(q1, q2, 1);
        }
      }
      flush_stats(s1, s2, 0);
    }

    ZM_STAMP();   // 7: epilogue (stores issued, statistics in LDS)
    // ---- closing barrier of the step: every wave is done with the ring's four oldest planes AND the statistics scratch is complete ----
    __syncthreads();
    ZM_STAMP();   // 8: closing barrier
    if (p.part != nullptr || p.pool_part != nullptr) {
      if (tid < 2 * 16 * 2) {
        const int which = tid >> 5, q = tid & 31;
        const int ch = q >> 1, k = q & 1;
        float* dst = which ? p.pool_part : p.part;
        if (dst != nullptr) {
          const float* rd = red + which * 4 * 16 * 2;
          const float a = rd[(0 * 16 + ch) * 2 + k] + rd[(1 * 16 + ch) * 2 + k] + rd[(2 * 16 + ch) * 2 + k] + rd[(3 * 16 + ch) * 2 + k];
          const int tile = (tzi * p.tilesY + tyi) * p.tilesX + txi;
          dst[(((size_t)n * p.tilesPerSample + tile) * 2 + k) * Cout + ch] = a;
        }
      }
    }

    ZM_STAMP();   // 9: statistics row stored
    ZM_STEP_DONE();
    // ---- advance ----
    if constexpr (NCH == 1) { off += 4; off = off >= 6 ? off - 6 : off; } else { off ^= 1; }
    newcol = false;
    if (++tzi == tilesZ) { tzi = 0; ++col; newcol = true; }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// conv3_zs_kernel: the z-march with the workgroup's waves SPLIT into roles (round 5, the structure DESIGN.md section 8 names as "not tried").
// One input chunk, no wide shortcut (the encoder's second conv: image shortcut, fused pool).  512 threads: waves 0-3 are CONSUMERS (one z-slice
// each: MFMA steps, epilogue, stores), waves 4-7 PRODUCERS (request, normalise + activate, LDS write of the NEXT step's four new planes, and the
// previous step's statistics row).  The two halves meet at ONE barrier per step (two with the fused z-pooling).  The ring has ten plane slots:
// the six planes the consumers read and the four the producers write are disjoint, plane a of a column run lives in slot a % 10.  Why: in
// conv3_zm_kernel every wave runs [stage -> MFMA -> epilogue] as one dependent chain and the stamps show each phase ~2.5 K cycles for a LONE
// workgroup - nothing is saturated, the chains just do not overlap; here the staging chain and the MFMA chain of a step run side by side.
// Two workgroups of eight waves per CU (76 KB LDS, <= 128 VGPRs: the accumulators and the prefetched pieces share registers - a thread is either).
// Same MFMAs in the same order, same epilogue and statistics rows: bit-identical to the other two kernels.
template <int ACTK, bool F16>
__global__ void __launch_bounds__(512, 4) conv3_zs_kernel(const Conv3Params p) {
  using T = typename std::conditional<F16, f16_t, uint16_t>::type;
  constexpr int TZ = 4, TY = 8, TX = 16, MS = 8, KPL = 8, VB = 32, HY = TY + 2, HX = TX + 2;
  constexpr int PLANE_B = HY * HX * VB, PP = HY * HX * 2, PL = PP - 256;
  constexpr int STEPS = 14, HSTR = HX * VB, RH = 4, SLOTS = 10;
  constexpr int RING_B = SLOTS * PLANE_B;                              // 57,600
  constexpr int WOFF = RING_B, ROFF = WOFF + STEPS * 1024;             // weights: [step][lane][16 B]
  constexpr int RED_B = 2 * 2 * 4 * 16 * 2 * 4;                        // statistics scratch [step parity][which][wave][16][2]: written by the consumers, read by a producer wave one step later
  constexpr int TOFF = ROFF + RED_B, TAB_B = 16 * 8 + 128, POFF = TOFF + TAB_B, POOL_B = 2 * (MS / 2) * 32 * 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[POFF + POOL_B];
  float* const tab = reinterpret_cast<float*>(smem + TOFF);
  float* const ektab = tab + 32;

  const int tid = threadIdx.x, lane = tid & 63;
  const bool producer = tid >= 256;                                    // wave-uniform
  const int ptid = tid & 255;                                          // thread id inside the role
  const int wave = __builtin_amdgcn_readfirstlane(ptid >> 6);          // consumers: z-slice
  const int j = lane & 15, g = lane >> 4;
  const int D = p.D, H = p.H, W = p.W;
  constexpr int Cout = 16;
  const int cg_off = (g & 1) * 16;
  const bool hi_tap = (g >> 1) != 0;
  const uint32_t hb = (uint32_t)(j * VB + cg_off);

  // producers' staging constants (conv3_zm_kernel): piece ptid of a plane ("A") and the pair's left-over piece ("L")
  const int sub = ptid & 1;
  const int hvA = ptid >> 1, hyA = hvA / HX, hxA = hvA - hyA * HX;
  const int selL = ptid / PL, idxL = 256 + (ptid - selL * PL), hvL = idxL >> 1, hyL = hvL / HX, hxL = hvL - hyL * HX;
  const bool hasL = ptid < 2 * PL;
  const uint32_t HWB = (uint32_t)(H * W * p.x_ld) * 2u;
  uint32_t relA = (uint32_t)((hyA * W + hxA) * p.x_ld + sub * KPL) * 2u;
  uint32_t relL = (uint32_t)((hyL * W + hxL) * p.x_ld + sub * KPL) * 2u + (hasL ? (uint32_t)selL * HWB : 0u);
  asm volatile("" : "+v"(relA), "+v"(relL));
  const uint32_t ldsA = (uint32_t)ptid * 16u, ldsL = (uint32_t)idxL * 16u;
  const char* __restrict__ xin = reinterpret_cast<const char*>(p.x);
  const char* __restrict__ wp = reinterpret_cast<const char*>(p.wp);

  const bool rank1 = p.sc != nullptr && p.sc_C == 1;
  if (tid < 16) {
    float a = 0.f;
    if (p.bias) a += p.bias[tid];
    if (p.sc && p.bias_sc) a += p.bias_sc[tid];
    ektab[tid] = a;
    ektab[16 + tid] = rank1 ? reinterpret_cast<const float*>(p.wsc)[tid] : 0.f;
  }
  for (int q = tid; q < STEPS * 64; q += 512)
    *reinterpret_cast<u32x4_t*>(smem + WOFF + q * 16) = *reinterpret_cast<const u32x4_t*>(wp + (uint32_t)q * 16u);

  const int tilesZ = p.tilesZ, colsPerSample = p.tilesY * p.tilesX;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, spx = gridDim.x >> 3;
  const int T8 = p.tilesPerXcd;
  int L0 = xcd * T8 + (int)(((long long)slot * T8) / spx), L1 = xcd * T8 + (int)(((long long)(slot + 1) * T8) / spx);
  {
    const int cap = min((xcd + 1) * T8, p.totalTiles);
    L1 = min(L1, cap);
  }
  if (L0 >= L1) return;
  int col = L0 / tilesZ, tzi = L0 - col * tilesZ;
  int n = 0, y0 = 0, x0 = 0, tyi = 0, txi = 0;
  uint32_t colb = 0;
  bool okA = false, okL = false;
  int sb = 0;                               // slot of ring-relative plane 0 of the current step
  int par = 0;                              // statistics scratch half of the current step
  bool newcol = true;
  int prev_row = -1;                        // row index (n * tilesPerSample + tile) of the step whose statistics wait in red[par ^ 1]; -1: none

  // Role-shared registers: the consumers' eight accumulators / the producers' six prefetched pieces (+ validity mask).
  f32x4_t acc[MS];
  uint32_t pmask = 0;
  auto PB = [&](int k) -> f32x4_t& { return acc[k]; };   // producer view: pieces 0..2 = planes {2, 3} of the step being prepared, 3..5 = planes {4, 5}

  auto slot_of = [&](int q) -> uint32_t { int s = sb + q; s = s >= SLOTS ? s - SLOTS : s; return (uint32_t)s * PLANE_B; };
  const bool has_norm = p.in_norm != nullptr;

  auto load3 = [&](f32x4_t* v, uint32_t& mask, int bit, int z) {      // the three pieces of the plane pair (z, z + 1)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool ok = okA && (unsigned)(z + k) < (unsigned)D;
      v[k] = __builtin_bit_cast(f32x4_t, *reinterpret_cast<const u32x4_t*>(xin + (ok ? colb + (uint32_t)(z + k) * HWB + relA : 0u)));
      mask |= ok ? 1u << (bit + k) : 0u;
    }
    const bool ok = okL && (unsigned)(z + selL) < (unsigned)D;
    v[2] = __builtin_bit_cast(f32x4_t, *reinterpret_cast<const u32x4_t*>(xin + (ok ? colb + (uint32_t)z * HWB + relL : 0u)));
    mask |= ok ? 1u << (bit + 2) : 0u;
  };
  auto store3 = [&](const f32x4_t* v, uint32_t mask, int bit, uint32_t s0, uint32_t s1, const float* psc, const float* psh) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k == 2 && !hasL) break;
      u32x4_t t = __builtin_bit_cast(u32x4_t, v[k]);
      const bool in = (mask >> (bit + k)) & 1u;
      if (!in) t = u32x4_t{0u, 0u, 0u, 0u};
      if (has_norm && in) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float a = fmaf(psc[2 * i], lo16<T>(t[i]), psh[2 * i]), b = fmaf(psc[2 * i + 1], hi16<T>(t[i]), psh[2 * i + 1]);
          act_pair<ACTK>(a, b, p.act);
          t[i] = pk16<T>(a, b);
        }
      }
      const uint32_t dst = k == 0 ? s0 + ldsA : k == 1 ? s1 + ldsA : (selL ? s1 : s0) + ldsL;
      *reinterpret_cast<u32x4_t*>(smem + dst) = t;
    }
  };
  auto load_tab = [&](float* psc, float* psh) {
    if (!has_norm) return;
    const f32x4_t* tp = reinterpret_cast<const f32x4_t*>(tab + (sub * KPL) * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4_t q = tp[i];
      psc[2 * i] = q[0]; psh[2 * i] = q[1]; psc[2 * i + 1] = q[2]; psh[2 * i + 1] = q[3];
    }
  };
  // the statistics row of the step that finished at the last barrier (producer wave 0 of the role)
  auto store_row = [&](int row, int half) {
    if (row < 0 || !producer || ptid >= 64) return;
    const int which = ptid >> 5, q = ptid & 31, ch = q >> 1, k = q & 1;
    float* dst = which ? p.pool_part : p.part;
    if (dst == nullptr) return;
    const float* rd = reinterpret_cast<const float*>(smem + ROFF) + half * 256 + which * 128;
    const float a = rd[(0 * 16 + ch) * 2 + k] + rd[(1 * 16 + ch) * 2 + k] + rd[(2 * 16 + ch) * 2 + k] + rd[(3 * 16 + ch) * 2 + k];
    dst[((size_t)row * 2 + k) * Cout + ch] = a;
  };

  char* __restrict__ yout = reinterpret_cast<char*>(p.y);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)0xFFFFFFF0u, 0x00020000);   // (epilogue rows: out-of-volume rows store beyond it)
  const uint32_t yrow = (uint32_t)(W * p.y_ld) * 2u;
  const bool pool2 = p.pool != nullptr && p.pool_sz == 2;

  // Per-step bookkeeping, identical in both roles (the roles run SEPARATE loops so that each gets its own register allocation - as one loop with
  // role branches inside, the compiler kept both roles' state live everywhere: 300 bytes of scratch at the 128-VGPR budget; the barriers of the
  // two loops pair up dynamically: every step has [two at a column start] + [one with the fused z-pooling] + [one at its end] in both)
#define ZS_STEP_BEGIN()                                                                                                  \
    if (newcol) {                                                                                                        \
      n = col / colsPerSample;                                                                                           \
      const int r_ = col - n * colsPerSample;                                                                            \
      tyi = r_ / p.tilesX; txi = r_ - tyi * p.tilesX;                                                                    \
      y0 = tyi * TY; x0 = txi * TX;                                                                                      \
      sb = 0;                                                                                                            \
    }                                                                                                                    \
    const int z0 = tzi * TZ;
#define ZS_STEP_END()                                                                                                    \
    prev_row = n * p.tilesPerSample + (tzi * p.tilesY + tyi) * p.tilesX + txi;                                           \
    par ^= 1;                                                                                                            \
    sb += 4; sb = sb >= SLOTS ? sb - SLOTS : sb;                                                                         \
    newcol = false;                                                                                                      \
    if (++tzi == tilesZ) { tzi = 0; ++col; newcol = true; }

  if (producer) {
    // ================================================== PRODUCER WAVES ==================================================
    for (int L = L0; L < L1; ++L) {
      ZS_STEP_BEGIN()
      if (newcol) {
        colb = (uint32_t)(((n * D) * H + (y0 - 1)) * W + (x0 - 1)) * (uint32_t)p.x_ld * 2u;
        okA = (unsigned)(y0 - 1 + hyA) < (unsigned)H && (unsigned)(x0 - 1 + hxA) < (unsigned)W;
        okL = hasL && (unsigned)(y0 - 1 + hyL) < (unsigned)H && (unsigned)(x0 - 1 + hxL) < (unsigned)W;
      }
      const bool next1 = (L + 1 < L1) && (tzi + 1 < tilesZ);           // the next step continues this column
      const bool next2 = (L + 2 < L1) && (tzi + 2 < tilesZ);
      if (newcol) {
        // a run's first step in a column: all six planes while the consumers wait (once per column)
        f32x4_t v0[3];
        uint32_t m0 = 0;
        if (has_norm && ptid < 16) *reinterpret_cast<f32x2_t*>(tab + ptid * 2) = *reinterpret_cast<const f32x2_t*>(&p.in_norm[(size_t)n * p.Cin + ptid].scale);
        pmask = 0;
        load3(v0, m0, 0, z0 - 1);
        load3(&PB(0), pmask, 0, z0 + 1);
        load3(&PB(3), pmask, 3, z0 + 3);
        __syncthreads();                                               // the table is visible; every consumer is past the previous column
        float psc[KPL], psh[KPL];
        load_tab(psc, psh);
        store3(v0, m0, 0, slot_of(0), slot_of(1), psc, psh);
        store3(&PB(0), pmask, 0, slot_of(2), slot_of(3), psc, psh);
        store3(&PB(3), pmask, 3, slot_of(4), slot_of(5), psc, psh);
        pmask = 0;
        if (next1) { load3(&PB(0), pmask, 0, z0 + TZ + 1); load3(&PB(3), pmask, 3, z0 + TZ + 3); }
        __syncthreads();                                               // the six planes are in LDS
      }
      // phase 1: the statistics row of the step that ended at the last barrier, then the NEXT step's four new planes
      store_row(prev_row, par ^ 1);
      if (next1) {
        float psc[KPL], psh[KPL];
        load_tab(psc, psh);
        // planes {2 .. 5} of step L + 1 = ring-relative planes 6 .. 9 of this step: the four slots the consumers do not read now
        store3(&PB(0), pmask, 0, slot_of(6), slot_of(7), psc, psh);
        store3(&PB(3), pmask, 3, slot_of(8), slot_of(9), psc, psh);
      }
      if (pool2) __syncthreads();                                      // (the consumers' z-pair exchange)
      // phase 2: request the planes of step L + 2; they fly across the barrier and the consumers' next MFMA steps
      pmask = 0;
      if (next2) { load3(&PB(0), pmask, 0, z0 + 2 * TZ + 1); load3(&PB(3), pmask, 3, z0 + 2 * TZ + 3); }
      __syncthreads();                                                 // end of the step
      ZS_STEP_END()
    }
    store_row(prev_row, par ^ 1);
    return;
  }

  // ==================================================== CONSUMER WAVES ====================================================
  for (int L = L0; L < L1; ++L) {
    ZS_STEP_BEGIN()
    if (newcol) {
      __syncthreads();
      __syncthreads();                                                 // the six planes are in LDS
    }
    const bool full = z0 + TZ <= D && y0 + TY <= H && x0 + TX <= W;
    const int vox0 = ((n * D + z0 + wave) * H + y0) * W + x0 + j;
    const bool okzx = full || (z0 + wave < D && x0 + j < W);
    const int yrem = full ? (1 << 20) : H - y0;
    float m[MS / 2][4];                                                // the pooled y-pair / x-pair maxima, carried over the pool barrier
    {
      float img[MS];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
        img[ms] = rank1 ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.sc) + ((okzx && ms < yrem) ? (uint32_t)(vox0 + ms * W) * 4u : 0u)) : 0.f;
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) acc[ms] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const uint32_t pz[3] = {slot_of(wave), slot_of(wave + 1), slot_of(wave + 2)};
      const uint32_t hb0 = hb + (hi_tap ? VB : 0), hb1 = hb + (hi_tap ? HX * VB : 0);
      const uint32_t wl = WOFF + (uint32_t)lane * 16u;
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        const uint32_t b0 = pz[dz] + hb0;
        u32x4_t w3[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) w3[dy] = *reinterpret_cast<const u32x4_t*>(smem + wl + (3 * dz + dy) * 1024);
#pragma unroll
        for (int h = 0; h < MS; h += RH) {
          u32x4_t row[RH + 2];
#pragma unroll
          for (int r = 0; r < RH + 2; ++r) row[r] = *reinterpret_cast<const u32x4_t*>(smem + b0 + (h + r) * HSTR);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int ms = 0; ms < RH; ++ms) acc[h + ms] = mfma_step<T>(w3[dy], row[ms + dy], acc[h + ms]);
        }
      }
#pragma unroll
      for (int s = 9; s < STEPS; ++s) {
        const uint32_t base = s < 12 ? pz[s < 12 ? s - 9 : 0] + hb1 : s == 12 ? (hi_tap ? pz[1] : pz[0]) + hb : pz[2] + hb;
        const int imm = s < 12 ? 2 * VB : (2 * HX + 2) * VB;
        const u32x4_t ws = *reinterpret_cast<const u32x4_t*>(smem + wl + s * 1024);
#pragma unroll
        for (int h = 0; h < MS; h += 4) {
          u32x4_t af[4];
#pragma unroll
          for (int ms = 0; ms < 4; ++ms) af[ms] = *reinterpret_cast<const u32x4_t*>(smem + base + (h + ms) * HSTR + imm);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ms = 0; ms < 4; ++ms) acc[h + ms] = mfma_step<T>(ws, af[ms], acc[h + ms]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // epilogue, first half: bias / rank-1 shortcut, statistics, one 8-byte store per m-subtile (conv3_zm_kernel's arithmetic, two m-subtiles at a time)
      const uint32_t yb0 = (uint32_t)(vox0 * p.y_ld + g * 4) * 2u;
      const f32x4_t addk = *reinterpret_cast<const f32x4_t*>(ektab + g * 4), w1k = *reinterpret_cast<const f32x4_t*>(ektab + 16 + g * 4);
      __builtin_amdgcn_s_waitcnt(0x0070);
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < MS / 2; ++k) {
        u32x2_t pk[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ms = 2 * k + h;
          const bool in = okzx && ms < yrem;      // branch-free rows (round 6)
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = acc[ms][r] + addk[r] + img[ms] * w1k[r];
            v[r] = in ? t : 0.f;
            s1[r] += v[r];
            s2[r] += v[r] * v[r];
          }
          pk[h] = u32x2_t{pk16s<T>(v[0], v[1]), pk16s<T>(v[2], v[3])};
          __builtin_amdgcn_raw_buffer_store_b64(pk[h], rs_y, (int)(in ? yb0 + ms * yrow : 0xFFFFFFFFu), 0, 0);
        }
        if (p.pool != nullptr) {
          const u32x2_t a = pk[0], b = pk[1];
          m[k][0] = fmaxf(lo16<T>(a[0]), lo16<T>(b[0])); m[k][1] = fmaxf(hi16<T>(a[0]), hi16<T>(b[0]));
          m[k][2] = fmaxf(lo16<T>(a[1]), lo16<T>(b[1])); m[k][3] = fmaxf(hi16<T>(a[1]), hi16<T>(b[1]));
#pragma unroll
          for (int r = 0; r < 4; ++r)
            m[k][r] = fmaxf(m[k][r], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m[k][r]), 0xB1, 0xF, 0xF, true)));
        }
      }
      if (p.part != nullptr) {   // the output's statistics go to LDS here, before the pool barrier: nothing of them is carried over it
        float* red = reinterpret_cast<float*>(smem + ROFF) + par * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(s1[r]), b = row16_sum(s2[r]);
          if (j == 0) *reinterpret_cast<f32x2_t*>(&red[(wave * 16 + g * 4 + r) * 2]) = f32x2_t{a, b};
        }
      }
      if (pool2 && (wave & 1)) {
        u32x2_t* ex = reinterpret_cast<u32x2_t*>(smem + POFF);
#pragma unroll
        for (int k = 0; k < MS / 2; ++k) ex[((wave >> 1) * (MS / 2) + k) * 32 + (lane >> 1)] = u32x2_t{pk16<T>(m[k][0], m[k][1]), pk16<T>(m[k][2], m[k][3])};
      }
    }
    if (pool2) __syncthreads();                                        // the odd z-slices' pooled rows are in LDS
    if (p.pool != nullptr) {
      float q1[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};
      if (pool2 && !(wave & 1)) {
        const u32x2_t* ex = reinterpret_cast<const u32x2_t*>(smem + POFF);
#pragma unroll
        for (int k = 0; k < MS / 2; ++k) {
          const u32x2_t o = ex[((wave >> 1) * (MS / 2) + k) * 32 + (lane >> 1)];
          m[k][0] = fmaxf(m[k][0], lo16<T>(o[0])); m[k][1] = fmaxf(m[k][1], hi16<T>(o[0]));
          m[k][2] = fmaxf(m[k][2], lo16<T>(o[1])); m[k][3] = fmaxf(m[k][3], hi16<T>(o[1]));
        }
      }
      if ((p.pool_sz == 1 || !(wave & 1)) && !(j & 1) && z0 + wave < D && x0 + j < W) {
        const int Dp = D / p.pool_sz, Hp = H >> 1, Wp = W >> 1;
        const int pz_ = (z0 + wave) / p.pool_sz, px = (x0 + j) >> 1;
        char* __restrict__ pout = reinterpret_cast<char*>(p.pool);
#pragma unroll
        for (int k = 0; k < MS / 2; ++k) {
          if (y0 + 2 * k < H) {
            const int py = (y0 >> 1) + k;
            *reinterpret_cast<u32x2_t*>(pout + (uint32_t)((((n * Dp + pz_) * Hp + py) * Wp + px) * p.pool_ld + g * 4) * 2u) =
                u32x2_t{pk16<T>(m[k][0], m[k][1]), pk16<T>(m[k][2], m[k][3])};
#pragma unroll
            for (int r = 0; r < 4; ++r) { q1[r] += m[k][r]; q2[r] += m[k][r] * m[k][r]; }
          }
        }
      }
      if (p.pool_part != nullptr) {
        float* red = reinterpret_cast<float*>(smem + ROFF) + par * 256 + 128;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(q1[r]), b = row16_sum(q2[r]);
          if (j == 0) *reinterpret_cast<f32x2_t*>(&red[(wave * 16 + g * 4 + r) * 2]) = f32x2_t{a, b};
        }
      }
    }
    __syncthreads();                                                   // end of the step: done with the six planes; the next four are written; red[par] complete
    ZS_STEP_END()
  }
#undef ZS_STEP_BEGIN
#undef ZS_STEP_END
}

int g_zm_mode = -1;   // -1: from the environment (BPX_CONV_ZM, default 1); 0 off; 1 on where it applies; 2 on even for short runs (tests)
int g_zm_wgs = 0;     // tests: cap on the number of workgroups (long runs that cross columns on small volumes); 0 = none
int g_zm_launches = 0;

}  // namespace

extern "C" int bpx_debug_conv_zm_launches(void) { return g_zm_launches; }
// tests / profiling: workgroups of the fp16 ELU instance (nch input chunks) the runtime keeps resident per CU (the compiler's occupancy remark does not
// see the LDS allocation granule)
extern "C" int bpx_debug_conv_zm_occupancy(int nch) {
  int n = 0;
  hipError_t e = nch == 0   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3_zs_kernel<1, true>, 512, 0)     // 0: the role-split form (512 threads)
                 : nch == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3_zm_kernel<1, 1, true>, 256, 0)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3_zm_kernel<3, 1, true>, 256, 0);
  return e == hipSuccess ? n : -1;
}
extern "C" int bpx_debug_set_conv_zm(int mode) {
  g_zm_mode = mode < 0 ? -1 : (mode & 0xFF);
  g_zm_wgs = mode < 0 ? 0 : (mode >> 8);
  return 0;
}

namespace bpxconv {

static int zm_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return n;
}

// 0 = launched; 1 = not applicable (the caller takes the lean kernel)
int launch_conv3_zm(const Conv3Params& p0, const TileCfg& c, hipStream_t s) {
  int mode = g_zm_mode, wgs = g_zm_wgs;
  if (mode < 0) {   // environment: same encoding as the hook (bits 8.. = workgroup cap)
    static const char* e = getenv("BPX_CONV_ZM");
    const int v = e ? atoi(e) : 1;
    mode = v & 0xFF; wgs = v >> 8;
  }
  const bool no_zs = (mode & 4) != 0;   // bit 2: without the role-split form (conv3_zs_kernel) - the test / A-B selector of conv3_zm_kernel<1, .., false>
  mode &= 3;
  if (mode == 0) return 1;
  if (!(c.tz == 4 && c.ty == 8 && c.tx == 16 && c.ns == 1) || p0.Cout != 16 || p0.ps > 1) return 1;
  if (!(p0.Cin == 16 || p0.Cin == 48)) return 1;
  if (p0.pool != nullptr && p0.Cin != 16) return 1;   // the fused MaxPool epilogue exists in the one-chunk kernels only: 48 -> 16 + pool takes the lean kernel
  {   // A/B aid: BPX_CONV_ZM_MASK bit 0 = one chunk without a wide shortcut, bit 1 = one chunk + shortcut of 16 .. 48 channels, bit 2 = three chunks
    static const char* e = getenv("BPX_CONV_ZM_MASK");
    static const int mask = e ? atoi(e) : 7;
    const int kind = p0.Cin == 48 ? 4 : (p0.sc != nullptr && p0.sc_C >= 16) ? 2 : 1;
    if (!(mask & kind)) return 1;
  }
  Conv3Params p = p0;
  p.tilesZ = cdiv(p.D, c.tz);
  p.tilesY = cdiv(p.H, c.ty);
  p.tilesX = cdiv(p.W, c.tx);
  p.tilesPerSample = p.tilesZ * p.tilesY * p.tilesX;
  p.totalTiles = p.N * p.tilesPerSample;
  p.tilesPerXcd = cdiv(p.totalTiles, 8);
  p.stamps = g_conv_stamps;
  p.dbg = 0;
  const int nch = p.Cin / 16;
  const bool sck = p.sc != nullptr && p.sc_C >= 16;
  {   // the role-split form (conv3_zs_kernel) of the one-chunk layers without a wide shortcut; BPX_CONV_ZS = 0 (or mode bit 2): conv3_zm_kernel
    static const char* e = getenv("BPX_CONV_ZS");
    static const int zs = e ? atoi(e) : 1;
    if (zs && !no_zs && nch == 1 && !sck) {
      int gz = std::max(8, (zm_cu_count() * 2) & ~7);
      if (wgs > 0) gz = std::max(8, std::min(gz, wgs & ~7));
      gz = std::min(gz, 8 * p.tilesPerXcd);
      if (mode == 1 && (p.totalTiles < 4 * gz || p.tilesZ < 4)) return 1;
      const bool elu_ = p.act == BPX_ACT_ELU;
      ++g_zm_launches;
      dim3 gridz((unsigned)gz, 1);
      if (p.f16) { if (elu_) conv3_zs_kernel<1, true><<<gridz, 512, 0, s>>>(p); else conv3_zs_kernel<0, true><<<gridz, 512, 0, s>>>(p); }
      else { if (elu_) conv3_zs_kernel<1, false><<<gridz, 512, 0, s>>>(p); else conv3_zs_kernel<0, false><<<gridz, 512, 0, s>>>(p); }
      return 0;
    }
  }
  if (sck && (p.sc_C > 48 || nch != 1)) return 1;                      // wider shortcuts / three chunks + shortcut: the lean kernel
  const int occ = zm_occ(nch, sck);
  int gx = std::max(8, (zm_cu_count() * occ) & ~7);
  if (wgs > 0) gx = std::max(8, std::min(gx, wgs & ~7));
  gx = std::min(gx, 8 * p.tilesPerXcd);
  // the ring pays off over runs of several z-steps; short runs (small volumes) stay on the lean kernel.  Both kernels give the same bits, so
  // the choice may depend on the batch size.
  if (mode == 1 && (p.totalTiles < 4 * gx || p.tilesZ < 4)) return 1;
  const bool elu = p.act == BPX_ACT_ELU;
  dim3 grid((unsigned)gx, 1);
#define Z(NCH, SCK)                                                                    \
  if (nch == NCH && sck == SCK) {                                                      \
    ++g_zm_launches;                                                                   \
    if (p.f16) {                                                                       \
      if (elu) conv3_zm_kernel<NCH, 1, true, SCK><<<grid, 256, 0, s>>>(p);             \
      else conv3_zm_kernel<NCH, 0, true, SCK><<<grid, 256, 0, s>>>(p);                 \
    } else {                                                                           \
      if (elu) conv3_zm_kernel<NCH, 1, false, SCK><<<grid, 256, 0, s>>>(p);            \
      else conv3_zm_kernel<NCH, 0, false, SCK><<<grid, 256, 0, s>>>(p);                \
    }                                                                                  \
    return 0;                                                                          \
  }
  Z(1, false) Z(1, true) Z(3, false)
#undef Z
  return 1;
}

}  // namespace bpxconv
