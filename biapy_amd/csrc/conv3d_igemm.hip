// Conv3d 3x3x3 "same" as an implicit GEMM on the gfx950 matrix cores.
//
//   GEMM view : D[co][v] = sum_{tap,ci} Wp[co][(tap,ci)] * A[(tap,ci)][v]
//               MFMA "A" operand = packed weights (16 output channels x K), "B" operand = activations
//               (K x 16 voxels) so that a lane ends up holding 4 CONSECUTIVE CHANNELS of one voxel
//               -> one 8-byte (bf16) / 16-byte (f32) store per lane, no LDS transpose in the epilogue.
//   tile      : TZ x TY x TX output voxels per 256-thread workgroup (4 waves x MS m-subtiles of 16
//               voxels), 16*NS output channels (blockIdx.y walks the rest of Cout).
//   K loop    : input channels in chunks of 16.  For each chunk the (TZ+2)(TY+2)(TX+2) x 16 halo is
//               staged ONCE in LDS ([voxel][16ch], 32 B (bf16) / 64 B (f32) per voxel - conflict-free
//               for ds_read_b128 at 32 B) and read 27x by the taps; the normalisation
//               + activation that precedes the convolution in the reference graph is applied while
//               staging (fp32 math), so the normalised tensor never exists in HBM.
//   weights   : pre-packed in fragment order [chunk][kgroup q][Cout][KPL]; a lane's 16-byte operand
//               is one coalesced global load (L1/L2 resident - a whole chunk is <= 57 KB).
//   epilogue  : FWD   : + bias (+ fused 1x1x1 shortcut conv of a second raw tensor, + its bias),
//                       store, per-(n,c) sum / sum^2 partials for the NEXT InstanceNorm.
//               DGRAD : g = acc * act'(scale*t+shift); store g; partials of sum(g), sum(g*xhat).
#include "conv3d_shared.h"

using namespace bpxconv;

namespace bpxconv { long long* g_conv_stamps = nullptr; }  // profiling hook: device buffer for per-workgroup cycle stamps
#define g_stamps bpxconv::g_conv_stamps
extern "C" int bpx_debug_set_conv_stamps(void* p) { g_stamps = (long long*)p; return 0; }
static int g_use_ws = 0;  // bf16 kernel selection, see bpx_debug_set_conv_ws below
static int g_conv_kg = -1; // two K groups in the small-tile kernel: -1 = environment (BPX_CONV_KG, default on), 0 / 1 (bpx_debug_set_conv_kg)
extern "C" int bpx_debug_set_conv_kg(int on) { g_conv_kg = on; return 0; }
static int64_t g_lean_min_vps = 32768;   // voxels per sample from which the lean kernel is used (test hook: bpx_debug_set_conv_ws 10 = 64^3 as in round 2, 11 = 32^3)

namespace {

// Stage an EZ x EY x EX block of voxels (16 channels of chunk `chunk`) into LDS as [voxel][16ch].
// Voxel (0,0,0) of the block sits at volume coordinate (oz,oy,ox); out-of-volume voxels are ZERO
// (Conv3d zero padding applies to the activated tensor, so the zero is written after the prologue).
template <typename T, int EZ, int EY, int EX>
__device__ __forceinline__ void stage_block(unsigned char* smem, const T* __restrict__ src, int ld, int cs, int chunk, int n, int D, int H,
                                            int W, int oz, int oy, int ox, const bpx_norm_rec* __restrict__ norm, int C_norm,
                                            int act, int tid) {
  constexpr int KPL = ElemTraits<T>::KPL, GPT = 16 / KPL, VB = 16 * (int)sizeof(T);
  constexpr int PIECES = EZ * EY * EX * GPT;
  constexpr int UNR = 4;
  const int sub = tid % GPT;
  float sc[KPL], sh[KPL];
  if (norm) {
#pragma unroll
    for (int e = 0; e < KPL; ++e) {
      bpx_norm_rec r = norm[(size_t)n * C_norm + chunk * 16 + sub * KPL + e];
      sc[e] = r.scale; sh[e] = r.shift;
    }
  }
  const T* sbase = src + (size_t)chunk * cs + sub * KPL;
  for (int base = 0; base < PIECES; base += 256 * UNR) {
    u32x4_t buf[UNR];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      int idx = base + u * 256 + tid;
      ok[u] = false;
      buf[u] = u32x4_t{0u, 0u, 0u, 0u};
      if (idx < PIECES) {
        int hv = idx / GPT;
        int hx = hv % EX, hy = (hv / EX) % EY, hz = hv / (EX * EY);
        int gz = oz + hz, gy = oy + hy, gx = ox + hx;
        if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) {
          ok[u] = true;
          buf[u] = *reinterpret_cast<const u32x4_t*>(sbase + ((((size_t)n * D + gz) * H + gy) * W + gx) * (size_t)ld);
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
          bpx_act_vec<std::is_same<T, float>::value, KPL>(f, act);   // (stage_block serves the fused shortcut only, which has no activation)
          v = pack16<T>(f);
        }
        *reinterpret_cast<u32x4_t*>(smem + (size_t)(idx / GPT) * VB + sub * 16) = v;
      }
    }
  }
}

// number of next-chunk weight fragments already re-loaded before step s (at most `per` per step from step np on, and
// never fragment k before step k has consumed the current one)
__host__ __device__ constexpr int wreg_next(int s, int np, int per) {
  int k = 0;
  for (int t = np; t < s; ++t) {
    int c = t + 1 - k;
    k += c < per ? (c > 0 ? c : 0) : per;
  }
  return k;
}

// TT: element type of the dgrad epilogue's activation operand t (BPX_MIX16: fp16 beside bf16 gradients), else T
// KG = 2 (round 6, the <= 16^3 layers of the bottom of the U): TWO K groups of four waves in one 512-thread workgroup.  Those launches have 64-128
// workgroups for 256 CUs and a serial chain of Cin / 16 chunk rounds (stage - barrier - 14 MFMA steps) per workgroup: they are latency-bound, and
// splitting K across WORKGROUPS costs a device-scope fence per workgroup (round 4: 2.5-5x slower).  Inside a workgroup it costs one LDS exchange:
// group kg takes the chunks kg, kg + 2, ... through its own pair of halo buffers (the chunk barriers are shared), group 1 hands its accumulators
// to group 0 through LDS at the end, group 0 runs the shortcut steps and the epilogue.  The chain is half as long and every SIMD holds two waves.
template <typename T, int TZ, int TY, int TX, int NS, int EPI, int ACTK, typename TT = T, int KG = 1>
__global__ void __launch_bounds__(256 * KG, 2) conv3_kernel(const Conv3Params p) {
  using Tr = ElemTraits<T>;
  constexpr int KPL = Tr::KPL, GPT = 16 / KPL, VB = 16 * (int)sizeof(T);
  constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;
  constexpr int QTOT = 27 * GPT, STEPS = (QTOT + 3) / 4, QPAD = STEPS * 4;
  constexpr int MT = TZ * TY * TX / 16, MS = MT / 4;
  static_assert(MT % 4 == 0 && MS >= 1, "tile must give every wave at least one m-subtile");
  static_assert((HV * GPT + 255) / 256 <= STEPS, "one staged piece per MFMA step");
  static_assert(sizeof(T) != 2 || wreg_next(STEPS, (HV * GPT + 255) / 256, (STEPS + (STEPS - (HV * GPT + 255) / 256) - 1) / (STEPS - (HV * GPT + 255) / 256)) == STEPS, "weight re-load schedule must cover every fragment");
  constexpr int RED_BYTES = 4 * NS * 16 * 2 * 4;
  constexpr int BUFB = HV * VB;                 // one halo buffer; two of them (double buffering) + reduction scratch
  static_assert(NS * 16 * 2 * 4 * 4 <= RED_BYTES, "reduction scratch");
  constexpr int NBUF = 2;
  static_assert(KG == 1 || (TX == 8 && sizeof(T) == 2), "two K groups: the small-tile 16-bit instances only");
  constexpr int XCH_BYTES = KG == 2 ? MS * NS * 256 * 16 : 0;     // the accumulator hand-over [ms][ns][thread] f32x4 re-uses the halo buffers
  static_assert(XCH_BYTES <= KG * NBUF * BUFB, "accumulator exchange fits the halo buffers");
  __shared__ __attribute__((aligned(16))) unsigned char smem_all[KG * NBUF * BUFB + RED_BYTES];

  const int kg = KG == 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;      // K group of this wave
  unsigned char* const smem = smem_all + kg * NBUF * BUFB;                                     // this group's pair of halo buffers
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x % p.tilesPerSample, n = blockIdx.x / p.tilesPerSample;
  const int txi = tile % p.tilesX, tyi = (tile / p.tilesX) % p.tilesY, tzi = tile / (p.tilesX * p.tilesY);
  const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;
  const int co_base = blockIdx.y * 16 * NS;
  const int Cout = p.Cout;

  int stamp_i = 0;
#define BPX_STAMP() do { if (p.stamps && tid == 0 && kg == 0 && blockIdx.y == 0 && stamp_i < 15) p.stamps[(size_t)blockIdx.x * 16 + stamp_i++] = (long long)__builtin_readcyclecounter(); } while (0)
  BPX_STAMP();  // 0: start
  if (p.stamps && tid == 0 && kg == 0 && blockIdx.y == 0)  // slot 15: where the workgroup ran (HW_ID | XCC_ID << 32)
    p.stamps[(size_t)blockIdx.x * 16 + 15] =
        (long long)(((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4));
  f32x4_t acc[MS][NS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // A wave's MS m-subtiles are consecutive rows (TX = 16) or row pairs (TX = 8) of ONE z-slice of the tile, so the LDS
  // address of subtile ms is the address of subtile 0 plus a compile-time stride: every ds_read below is
  // VGPR base + immediate.
  static_assert(TZ == 4 && MS * 16 == TY * TX && (TX == 16 || TX == 8), "wave = z-slice mapping");
  constexpr int HSTR = (TX >= 16 ? 1 : 16 / TX) * HX * VB;      // halo-buffer stride between m-subtiles
  constexpr int TSTR = 16 * VB;                                  // same for the un-haloed [voxel][16ch] block (shortcut)
  int hb0, tb0;
  {
    int t = (wave * MS) * 16 + j;
    int tz = t / (TY * TX), ty = (t / TX) % TY, tx = t % TX;
    hb0 = ((tz * HY + ty) * HX + tx) * VB;
    tb0 = t * VB;
  }
  // Per-lane K decomposition (DESIGN.md "K order").  bf16: a step covers two taps x 16 channels; lanes g<2 take tap A,
  // lanes g>=2 tap B.  The taps are paired (bpx_tap_order_bf16) so that addr(B)-addr(A) is one of three constants;
  // adding it to per-lane bases once makes every ds_read address = VGPR base + compile-time immediate.
  const int cg_off = (GPT == 2 ? (g & 1) : g) * 16;
  const bool hi_tap = (GPT == 2) && (g >> 1);
  constexpr int NCLS = (GPT == 2) ? 4 : 1;
  int lbase[NCLS];
  if (GPT == 2) {
    lbase[0] = hb0 + cg_off + (hi_tap ? VB : 0);
    lbase[1 % NCLS] = hb0 + cg_off + (hi_tap ? HX * VB : 0);
    lbase[2 % NCLS] = hb0 + cg_off + (hi_tap ? HY * HX * VB : 0);
    lbase[NCLS - 1] = hb0 + cg_off;
  } else {
    lbase[0] = hb0 + cg_off;
  }

  // ---- staging plan: which 16-byte pieces of the halo this thread moves (fixed for the tile) --------
  constexpr int NPIECE = HV * GPT;                 // LDS image is piece-linear: piece idx lives at byte idx*16
  constexpr int NP = (NPIECE + 255) / 256;
  const int sub = tid % GPT;
  uint32_t goff[NP];
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    int idx = u * 256 + tid;
    goff[u] = 0xFFFFFFFFu;
    if (idx < NPIECE) {
      int hv = idx / GPT;
      int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
      int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      if (gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
        goff[u] = (uint32_t)((((uint32_t)gz * p.H + gy) * p.W + gx) * (uint32_t)p.x_ld);
    }
  }
  const T* __restrict__ xin = reinterpret_cast<const T*>(p.x) + (size_t)n * p.D * p.H * p.W * p.x_ld + sub * KPL;
  const T* __restrict__ wp = reinterpret_cast<const T*>(p.wp);
  const bpx_norm_rec* __restrict__ nrec = p.in_norm ? p.in_norm + (size_t)n * p.Cin + sub * KPL : nullptr;
  const int nchunks = p.Cin / 16;   // KG == 2: a multiple of 2 (launcher); group kg walks kg, kg + KG, ...

  u32x4_t pbuf[NP];
  float psc[KPL], psh[KPL];

  // One staged piece: normalise + activate (fp32) the raw 16 bytes held in pbuf[u] and store them into the LDS buffer
  // at byte offset `wbuf`; then re-use the registers for the same piece of channel chunk `next_chunk` (if any).
#define BPX_STAGE_PIECE(u, wbuf, next_chunk)                                                                  \
  do {                                                                                                        \
    const int idx_ = (u) * 256 + tid;                                                                         \
    if (idx_ < NPIECE) {                                                                                      \
      u32x4_t v_ = pbuf[u];                                                                                   \
      if (nrec && goff[u] != 0xFFFFFFFFu && !(p.dbg & 2)) {                                                   \
        float f_[KPL];                                                                                        \
        unpack16<T>(v_, f_);                                                                                  \
        if (ACTK == 1) { _Pragma("unroll") for (int e_ = 0; e_ < KPL; ++e_) f_[e_] = apply_act_rt<T, 1>(fmaf(psc[e_], f_[e_], psh[e_]), p.act); } \
        else { _Pragma("unroll") for (int e_ = 0; e_ < KPL; ++e_) f_[e_] = fmaf(psc[e_], f_[e_], psh[e_]);                    \
               bpx_act_vec<std::is_same<T, float>::value, KPL, ACTK == 2>(f_, p.act); }                                               \
        v_ = pack16<T>(f_);                                                                                   \
      }                                                                                                       \
      *reinterpret_cast<u32x4_t*>(smem + (wbuf) + (size_t)idx_ * 16) = v_;                                    \
      if ((next_chunk) < nchunks && goff[u] != 0xFFFFFFFFu && !(p.dbg & 4))                                   \
        pbuf[u] = *reinterpret_cast<const u32x4_t*>(xin + goff[u] + (size_t)(next_chunk) * p.x_cs);                       \
    }                                                                                                         \
  } while (0)
#define BPX_LOAD_NORM(chunk_)                                                                                 \
  do {                                                                                                        \
    if (nrec && (chunk_) < nchunks) {                                                                         \
      _Pragma("unroll") for (int e_ = 0; e_ < KPL; ++e_) {                                                    \
        bpx_norm_rec r_ = nrec[(chunk_) * 16 + e_];                                                           \
        psc[e_] = r_.scale; psh[e_] = r_.shift;                                                               \
      }                                                                                                       \
    }                                                                                                         \
  } while (0)

  BPX_STAMP();  // 1: index math done
  {
  // ---- prologue: chunk 0 -> LDS buffer 0, chunk 1 -> registers ------------------------------------------------
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    pbuf[u] = u32x4_t{0u, 0u, 0u, 0u};
    if (goff[u] != 0xFFFFFFFFu) pbuf[u] = *reinterpret_cast<const u32x4_t*>(xin + goff[u] + (size_t)kg * p.x_cs);
  }
  BPX_LOAD_NORM(kg);
#pragma unroll
  for (int u = 0; u < NP; ++u) BPX_STAGE_PIECE(u, 0, kg + KG);
  BPX_LOAD_NORM(kg + KG);
  BPX_STAMP();  // 2: chunk 0 transformed + written (includes the HBM latency of its loads)
  __syncthreads();
  BPX_STAMP();  // 3: first barrier

  // ---- main loop: the MFMAs of chunk c read LDS buffer c&1 while the SAME wave, between its MFMA steps, transforms
  //      chunk c+1 (already in registers) into the other buffer and re-issues the global loads of chunk c+2.  VALU
  //      (normalise/ELU/convert), the matrix pipe, LDS and HBM latency overlap inside every wave; one barrier per chunk.
  // Weight operands.  VMEM results retire IN ORDER (one vmcnt queue), so a wait for an L1-resident weight fragment
  // issued after a halo prefetch stalls for the full HBM latency of that prefetch.  WREG kernels therefore keep the
  // whole chunk's fragments in registers: they are (re)loaded for chunk c+1 during the last steps of chunk c, AFTER the
  // step loop has issued its halo re-loads, and are first needed a barrier later - no wait inside the step loop.
  constexpr bool WREG = (sizeof(T) == 2) && (STEPS * NS * 4 <= 112);
  constexpr int NWR = WREG ? STEPS : 1;
  constexpr int WSTEPS = STEPS - NP;                                  // steps without a staging piece
  constexpr int PER = WREG ? (STEPS + WSTEPS - 1) / WSTEPS : 1;       // fragments re-loaded per such step
  u32x4_t wreg[NWR][NS];
  if (WREG) {
    const T* wl0 = wp + ((size_t)kg * QPAD * Cout + (size_t)g * Cout + co_base + j) * KPL;
#pragma unroll
    for (int s = 0; s < NWR; ++s)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) wreg[s][ns] = *reinterpret_cast<const u32x4_t*>(wl0 + ((size_t)s * 4 * Cout + ns * 16) * KPL);
  }
  // Round 4 (cycle stamps of the <= 16^3 layers, one workgroup per CU = one wave per SIMD): a step was ONE exposed latency - 486 cycles per step for
  // the 8 MFMAs (128 cycles) of the NS = 4 instances, whose next step's weights were requested one step ahead (L2 latency ~500 cycles), and 264
  // cycles per step for the 4 MFMAs of the register-resident (WREG) instances, which waited for the step's own LDS fragment reads.  Now: the
  // fragment reads run one step ahead of the MFMAs (two register sets), and the non-WREG weights come through a ring of WD + 1 = 4 steps
  // (three loads in flight per output-channel group; the ring restarts per chunk: the first WD steps of the next chunk are requested after the
  // last step's MFMAs and fly across the chunk barrier).
  // Only for the 4x4x8 tiles of the <= 16^3 layers (two m-subtiles per wave): the 16-wide tiles (the lean kernel's fallback, the fp32 mode) keep one
  // fragment set and a one-step ring - with four or eight m-subtiles the second set spills.
  constexpr bool SMALL = TX == 8;
  constexpr int WD = SMALL ? 3 : 1;
  constexpr int NAF = SMALL ? 2 : 1;
  u32x4_t wq[WREG ? 1 : WD + 1][NS];
  if (!WREG) {
    const T* wl0 = wp + ((size_t)kg * QPAD * Cout + (size_t)g * Cout + co_base + j) * KPL;
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) wq[d][ns] = *reinterpret_cast<const u32x4_t*>(wl0 + ((size_t)d * 4 * Cout + ns * 16) * KPL);
  }
  int cur = 0;
  for (int chunk = kg; chunk < nchunks; chunk += KG) {
    const int nxt = BUFB - cur;
    const bool stage_next = chunk + KG < nchunks;
    const T* wl = wp + ((size_t)chunk * QPAD * Cout + (size_t)g * Cout + co_base + j) * KPL;
    const T* wl1 = wl + (size_t)KG * QPAD * Cout * KPL;   // this group's next chunk
    u32x4_t af[NAF][MS];
    auto read_frags = [&](int s_, u32x4_t* f) {
      const int tapA = (GPT == 2) ? bpx_tap_order_bf16(2 * s_) : s_;
      const int cls = (GPT == 2) ? (s_ < 9 ? 0 : s_ < 12 ? 1 : s_ == 12 ? 2 : 3) : 0;
      const int imm = tap_off<HY, HX, VB>(tapA);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) f[ms] = *reinterpret_cast<const u32x4_t*>(smem + lbase[cls] + ms * HSTR + imm);
    };
    if (NAF == 2) read_frags(0, af[0]);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (!WREG && s + WD < STEPS) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          wq[WREG ? 0 : (s + WD) % (WD + 1)][ns] = *reinterpret_cast<const u32x4_t*>(wl + ((size_t)(s + WD) * 4 * Cout + ns * 16) * KPL);
      }
      // the NEXT step's LDS reads are issued before this step's MFMAs; sched_barrier keeps the compiler from re-serialising to save VGPRs
      if (NAF == 2) { if (s + 1 < STEPS) read_frags(s + 1, af[(s + 1) % NAF]); }
      else read_frags(s, af[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = mfma_step<T>(WREG ? wreg[WREG ? s : 0][ns] : wq[WREG ? 0 : s % (WD + 1)][ns], af[s % NAF][ms], acc[ms][ns]);
      }
      if (s < NP && stage_next) BPX_STAGE_PIECE(s < NP ? s : 0, nxt, chunk + 2 * KG);
      if (WREG && s >= NP && stage_next) {
        // re-load fragments [k0, k1) of the NEXT chunk; never a fragment this chunk has not consumed yet (k <= s)
        const int k0 = wreg_next(s, NP, PER), k1 = wreg_next(s + 1, NP, PER);
#pragma unroll
        for (int k = k0; k < k1; ++k) {
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
            wreg[WREG ? (k < STEPS ? k : 0) : 0][ns] = *reinterpret_cast<const u32x4_t*>(wl1 + ((size_t)k * 4 * Cout + ns * 16) * KPL);
        }
      }
    }
    if (!WREG && stage_next) {   // the ring's first WD steps of the next chunk: in flight across the chunk barrier
#pragma unroll
      for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) wq[WREG ? 0 : d][ns] = *reinterpret_cast<const u32x4_t*>(wl1 + ((size_t)d * 4 * Cout + ns * 16) * KPL);
    }
    BPX_STAMP();  // 4,6,8..: step loop of the chunk done
    BPX_LOAD_NORM(chunk + 2 * KG);
    // flip the read buffer: every ds_read base register moves by +-BUFB (cheaper than an add per read)
    const int flip = cur ? -BUFB : BUFB;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) lbase[c] += flip;
    cur = nxt;
    __syncthreads();  // buffer `nxt` is complete and nobody reads the old one any more
    BPX_STAMP();  // 5,7,9..: barrier passed
  }
  }
#undef BPX_STAGE_PIECE
#undef BPX_LOAD_NORM
  if constexpr (KG == 2) {   // group 1's accumulators -> group 0 (the loop's last barrier has released every halo buffer)
    f32x4_t* xch = reinterpret_cast<f32x4_t*>(smem_all);
    if (kg == 1) {
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) xch[(ms * NS + ns) * 256 + tid] = acc[ms][ns];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ms][ns] += xch[(ms * NS + ns) * 256 + tid];
    }
  }

  // ---- fused 1x1x1 shortcut on a second raw tensor (EPI_FWD only) ------------------------------
  if (EPI == EPI_FWD && p.sc != nullptr && p.sc_C >= 16) {
    const T* __restrict__ scin = reinterpret_cast<const T*>(p.sc);
    const T* __restrict__ wsc = reinterpret_cast<const T*>(p.wsc);
    const int nch = p.sc_C / 16;
    for (int chunk = 0; chunk < nch; ++chunk) {
      __syncthreads();
      if (kg == 0) stage_block<T, TZ, TY, TX>(smem, scin, p.sc_ld, p.sc_cs, chunk, n, p.D, p.H, p.W, z0, y0, x0, nullptr, 0, 0, tid);
      __syncthreads();
      if (kg != 0) continue;
      const T* wl = wsc + ((size_t)chunk * 4 * Cout + (size_t)g * Cout + co_base + j) * KPL;
      u32x4_t wf[NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) wf[ns] = *reinterpret_cast<const u32x4_t*>(wl + (size_t)ns * 16 * KPL);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        u32x4_t af = *reinterpret_cast<const u32x4_t*>(smem + tb0 + ms * TSTR + cg_off);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = mfma_step<T>(wf[ns], af, acc[ms][ns]);
      }
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  float s1[NS][4], s2[NS][4];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[ns][r] = s2[ns][r] = 0.f;

  if (kg == 0) {
  T* __restrict__ yout = reinterpret_cast<T*>(p.y);
  // Every operand of the epilogue is requested first, unconditionally (out-of-volume lanes read the tile's first voxel) and awaited ONCE:
  // with the loads inside the predicated per-voxel blocks the compiler put an `s_waitcnt vmcnt(0)` into each block, and that counter also holds
  // the stores - the epilogue was a chain of MS x NS [load, wait for it AND the previous store's acknowledgement, store] rounds.
  using TRaw = typename std::conditional<sizeof(TT) == 2, u32x2_t, f32x4_t>::type;
  const bool rank1 = EPI == EPI_FWD && p.sc != nullptr && p.sc_C == 1;
  const bool has_t = EPI != EPI_FWD && p.t_norm != nullptr;
  const size_t vox00 = (((size_t)n * p.D + z0) * p.H + y0) * p.W + x0;
  bool okv[MS];
  size_t voxv[MS];
  float imgv[MS];
  TRaw traw[MS][NS];
  float addv[NS][4], w1v[NS][4];
  bpx_norm_rec recv[(EPI == EPI_FWD) ? 1 : NS][4];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int t = (wave * MS + ms) * 16 + j;
    const int z = z0 + t / (TY * TX), y = y0 + (t / TX) % TY, x = x0 + t % TX;
    okv[ms] = z < p.D && y < p.H && x < p.W;
    voxv[ms] = okv[ms] ? (((size_t)n * p.D + z) * p.H + y) * p.W + x : vox00;
    imgv[ms] = rank1 ? reinterpret_cast<const float*>(p.sc)[voxv[ms]] : 0.f;
    if (has_t) {
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const int co = co_base + ns * 16 + g * 4;
        traw[ms][ns] = *reinterpret_cast<const TRaw*>(reinterpret_cast<const TT*>(p.t) + voxv[ms] * (size_t)p.t_ld + (size_t)(co >> 4) * p.t_cs + (co & 15));
      }
    }
  }
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int co = co_base + ns * 16 + g * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      addv[ns][r] = 0.f; w1v[ns][r] = 0.f;
      if (EPI == EPI_FWD) {
        if (p.bias) addv[ns][r] += p.bias[co + r];
        if (p.sc && p.bias_sc) addv[ns][r] += p.bias_sc[co + r];
        if (rank1) w1v[ns][r] = reinterpret_cast<const float*>(p.wsc)[co + r];
      } else if (has_t) {
        recv[ns][r] = p.t_norm[(size_t)n * Cout + co + r];
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), once
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int co = co_base + ns * 16 + g * 4;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
      if (okv[ms]) {
        float v[4];
        if (EPI == EPI_FWD) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[ms][ns][r] + addv[ns][r] + imgv[ms] * w1v[ns][r];
            s1[ns][r] += v[r];
            s2[ns][r] += v[r] * v[r];
          }
        } else {
          if (has_t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float tv;
              if constexpr (sizeof(TT) == 2) tv = (r & 1) ? hi16<TT>(traw[ms][ns][r >> 1]) : lo16<TT>(traw[ms][ns][r >> 1]);
              else tv = traw[ms][ns][r];
              const bpx_norm_rec& rc = recv[(EPI == EPI_FWD) ? 0 : ns][r];
              float u = fmaf(rc.scale, tv, rc.shift);
              v[r] = acc[ms][ns][r] * apply_act_bwd_rt<T, ACTK>(u, p.t_act);
              float xh = (tv - rc.mean) * rc.rstd;
              s1[ns][r] += v[r];
              s2[ns][r] += v[r] * xh;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ms][ns][r];
          }
        }
        T* yp = yout + voxv[ms] * (size_t)p.y_ld + (size_t)(co >> 4) * p.y_cs + (co & 15);
        if (p.dbg & 8) continue;
        if constexpr (std::is_same<T, float>::value) {
          *reinterpret_cast<f32x4_t*>(yp) = f32x4_t{v[0], v[1], v[2], v[3]};
        } else {
          *reinterpret_cast<u32x2_t*>(yp) = u32x2_t{pk16s<T>(v[0], v[1]), pk16s<T>(v[2], v[3])};
        }
      }
    }
  }

  }   // kg == 0
  BPX_STAMP();  // epilogue stores issued
  if (p.part != nullptr) {
    float* red = reinterpret_cast<float*>(smem_all + KG * NBUF * BUFB);  // [wave][NS*16][2]
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[ns][r], b = s2[ns][r];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
        if (j == 0 && kg == 0) {
          red[((wave * NS * 16) + ns * 16 + g * 4 + r) * 2 + 0] = a;
          red[((wave * NS * 16) + ns * 16 + g * 4 + r) * 2 + 1] = b;
        }
      }
    __syncthreads();
    if (tid < NS * 16 * 2 && kg == 0) {
      int c = tid >> 1, k = tid & 1;
      float a = red[(0 * NS * 16 + c) * 2 + k] + red[(1 * NS * 16 + c) * 2 + k] + red[(2 * NS * 16 + c) * 2 + k] +
                red[(3 * NS * 16 + c) * 2 + k];
      p.part[(((size_t)n * p.tilesPerSample + tile) * 2 + k) * Cout + co_base + c] = a;
    }
  }
}

template <typename T, int EPI, typename TT = T>
int launch_conv3(const Conv3Params& p0, const TileCfg& c, hipStream_t s) {
  Conv3Params p = p0;
  int tilesZ = cdiv(p.D, c.tz);
  p.tilesY = cdiv(p.H, c.ty);
  p.tilesX = cdiv(p.W, c.tx);
  p.tilesPerSample = tilesZ * p.tilesY * p.tilesX;
  { const char* e = getenv("BPX_CONV_DBG"); p.dbg = e ? atoi(e) : 0; }
  p.stamps = g_stamps;  // profiling ablations: 1 no MFMA, 2 no transform, 4 no re-loads, 8 no stores
  dim3 grid((unsigned)(p.N * p.tilesPerSample), (unsigned)(p.Cout / (16 * c.ns)));
  const bool elu = (EPI == EPI_FWD ? p.act : p.t_act) == BPX_ACT_ELU;
  const bool ext = (EPI == EPI_FWD ? p.act : p.t_act) > BPX_ACT_SILU;   // leaky_relu ... softplus: the ACTK = 2 instances
  // two K groups per workgroup (KG = 2) for the small-tile 16-bit launches with an even number of >= 4 input chunks: the <= 16^3 layers
  if constexpr (sizeof(T) == 2) {
    static const bool kg_env = getenv("BPX_CONV_KG") == nullptr || atoi(getenv("BPX_CONV_KG")) != 0;   // A/B: BPX_CONV_KG=0
    const int nchunks = p.Cin / 16;
    if ((g_conv_kg < 0 ? kg_env : g_conv_kg != 0) && c.tz == 4 && c.ty == 4 && c.tx == 8 && (c.ns == 2 || c.ns == 4) && nchunks >= 4 && nchunks % 2 == 0) {
#define LK(NS)                                                                                   \
      if (c.ns == NS) {                                                                          \
        if (elu) conv3_kernel<T, 4, 4, 8, NS, EPI, 1, TT, 2><<<grid, 512, 0, s>>>(p);            \
        else if (ext) conv3_kernel<T, 4, 4, 8, NS, EPI, 2, TT, 2><<<grid, 512, 0, s>>>(p);       \
        else conv3_kernel<T, 4, 4, 8, NS, EPI, 0, TT, 2><<<grid, 512, 0, s>>>(p);                \
        return 0;                                                                                \
      }
      LK(2) LK(4)
#undef LK
    }
  }
#define L(TZ, TY, TX, NS)                                                        \
  if (c.tz == TZ && c.ty == TY && c.tx == TX && c.ns == NS) {                    \
    if (elu) conv3_kernel<T, TZ, TY, TX, NS, EPI, 1, TT><<<grid, 256, 0, s>>>(p);    \
    else if (ext) conv3_kernel<T, TZ, TY, TX, NS, EPI, 2, TT><<<grid, 256, 0, s>>>(p);   \
    else conv3_kernel<T, TZ, TY, TX, NS, EPI, 0, TT><<<grid, 256, 0, s>>>(p);        \
    return 0;                                                                    \
  }
  if constexpr (sizeof(T) == 2) {  // the 512-voxel tile's fp32 halo (69 KB) exceeds static LDS; bf16 only
    L(4, 8, 16, 1)
  }
  L(4, 4, 16, 1) L(4, 4, 16, 2) L(4, 4, 16, 3) L(4, 4, 16, 4) L(4, 4, 8, 1) L(4, 4, 8, 2) L(4, 4, 8, 3) L(4, 4, 8, 4)
#undef L
  return 1;
}

}  // namespace

// bf16 kernel selection: 0 = automatic (lean persistent kernel of conv3d_lean.hip for >= 32^3 volumes, the double-buffered
// kernel of this file below that), 4 = always double-buffered, 5 = always lean persistent.
// Record of two schedules that were measured in round 1 and removed from the library in round 2 (wave-specialised 8-wave
// persistent | persistent 4-wave with a cross-tile stage pipeline), against the double-buffered kernel, cfg-2 layers, B=4 (us):
//   fwd 48->16@128^3: 781 | 868 | 1006     fwd 16->16+img@128^3: 463 | 481 | 484     dgrad 16->48@128^3: 1077 | 2373 | 1205
// Both lost: they hold a second stage in registers and either spill or drop to fewer co-resident workgroups.  The lean kernel
// goes the other way (3-4 workgroups per CU, nothing pipelined inside a workgroup) and wins wherever a CU gets >= 4 rounds of
// tiles: 725 / 342 / 944 us on the same three layers.
extern "C" int bpx_debug_set_conv_ws(int on) {
  if (on == 10 || on == 11) { g_lean_min_vps = on == 10 ? 262144 : 32768; return 0; }
  g_use_ws = on;
  return 0;
}

// Lean persistent kernel (conv3d_lean.hip) where a CU gets >= 4 rounds of tiles; it uses 32-bit element offsets and
// 16-byte vector loads of the per-channel parameter arrays.
static bool use_lean(int dtype, const Conv3Params& p) {
  // >= 32^3 voxels per sample (64^3 until round 3: after the epilogue / fragment-reuse work on the lean kernel it wins at 32^3 too - B = 4, same process: fwd 192 -> 64 98 -> 88 us, 64 -> 64 + sc192 59 -> 50, dgrad 64 -> 64 43.8 -> 35.2, 64 -> 32 27.6 -> 21.0)
  // The choice must NOT depend on the batch size: a sample's result has to be the same bits whatever batch it travels in (sharded sliding windows
  // compare checksums across different batch compositions).
  const int64_t vps = (int64_t)p.D * p.H * p.W, vox = vps * p.N;
  if ((dtype != BPX_BF16 && dtype != BPX_F16) || !(g_use_ws == 5 || (g_use_ws == 0 && vps >= g_lean_min_vps))) return false;
  if ((p.in_norm && p.act > BPX_ACT_SILU) || (p.t_norm && p.t_act > BPX_ACT_SILU)) return false;   // the round-4 activation codes: plain kernel only
  const int64_t ldmax = std::max<int64_t>(std::max(p.x_ld, p.y_ld), std::max(p.sc ? p.sc_ld : 0, p.t ? p.t_ld : 0));
  auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  // 32-bit byte offsets: a chunk-planar tensor extends over (channels / 16) planes
  auto ext = [](int cs, int C) { return cs == 16 ? (int64_t)0 : (int64_t)cs * (C / 16); };
  const int64_t planar = std::max(std::max(ext(p.x_cs, p.Cin), ext(p.y_cs, p.Cout)), std::max(p.sc ? ext(p.sc_cs, p.sc_C) : 0, p.t ? ext(p.t_cs, p.Cout) : 0));
  return p.W > 8 && vox * ldmax < (1ll << 31) && planar < (1ll << 31) && vox < (1ll << 30) && al(p.bias) && al(p.bias_sc) && al(p.wsc) && al(p.in_norm) && al(p.t_norm);
}

extern "C" int bpx_conv3d_stats_tiles(int dtype, int N, int D, int H, int W, int Cout) {
  TileCfg c = pick_cfg(dtype, D, H, W, Cout);
  int t = cdiv(D, c.tz) * cdiv(H, c.ty) * cdiv(W, c.tx);
  (void)N;
  return t;
}

static int check_tensor(const char* fn, const char* name, const bpx_tensor& t, int esize, bool need16) {
  BPX_CHECK(t.ptr != nullptr, "%s: %s.ptr is null", fn, name);
  BPX_CHECK(t.C >= 1 && t.ld >= (t.cs ? 16 : t.C), "%s: %s has ld %d < C %d", fn, name, t.ld, t.C);
  if (need16) {
    BPX_CHECK(t.C % 16 == 0, "%s: %s.C = %d must be a multiple of 16", fn, name, t.C);
    BPX_CHECK(((uintptr_t)t.ptr % 16) == 0 && ((size_t)t.ld * esize) % 16 == 0, "%s: %s must be 16-byte aligned (ptr and ld)", fn, name);
  }
  return 0;
}

// chunk-planar operand (bpx_tensor.cs != 0): planes must hold every voxel and keep 16-byte alignment
static int check_planar(const char* fn, const char* name, const bpx_tensor& t, int N, int D, int H, int W) {
  if (t.ptr == nullptr || t.cs == 0) return 0;
  const int64_t vox = (int64_t)N * D * H * W;
  BPX_CHECK(t.ld >= 16 && t.cs % 8 == 0 && t.cs >= (vox - 1) * t.ld + 16 && t.cs < (1ll << 31), "%s: %s has chunk stride %lld for %lld voxels of pitch %d", fn,
            name, (long long)t.cs, (long long)vox, t.ld);
  return 0;
}

static int conv3d_fwd_impl(const char* fn, int dtype, int N, int D, int H, int W, bpx_tensor x, const bpx_norm_rec* in_norm_d, int act,
                           const void* w_packed_d, const float* bias_d, bpx_tensor sc, const void* w_sc_d,
                           const float* bias_sc_d, bpx_tensor y, float* stats_part_d, int pool_sz, bpx_tensor pooled,
                           float* pool_stats_part_d, bpx_stream_t stream) {
  BPX_CHECK(dtype == BPX_BF16 || dtype == BPX_F32 || dtype == BPX_F16, "%s: dtype must be BF16, F16 (forward only) or F32", fn);
  int es = (int)dtype_size(dtype);
  BPX_CHECK(N > 0 && D > 0 && H > 0 && W > 0, "%s: empty volume", fn);
  if (check_tensor(fn, "x", x, es, true) || check_tensor(fn, "y", y, es, true)) return 1;
  BPX_CHECK(w_packed_d != nullptr, "%s: packed weights are null", fn);
  if (sc.ptr) {
    BPX_CHECK(w_sc_d != nullptr, "%s: shortcut weights are null", fn);
    if (sc.C != 1 && check_tensor(fn, "sc", sc, es, true)) return 1;
  }
  Conv3Params p{};
  p.N = N; p.D = D; p.H = H; p.W = W;
  p.x = x.ptr; p.x_ld = x.ld; p.Cin = x.C; p.in_norm = in_norm_d; p.act = act;
  p.wp = w_packed_d; p.bias = bias_d;
  p.sc = sc.ptr; p.sc_ld = sc.ld; p.sc_C = sc.ptr ? sc.C : 0; p.wsc = w_sc_d; p.bias_sc = bias_sc_d;
  p.y = y.ptr; p.y_ld = y.ld; p.Cout = y.C; p.part = stats_part_d;
  if (check_planar(fn, "x", x, N, D, H, W) || check_planar(fn, "sc", sc, N, D, H, W) || check_planar(fn, "y", y, N, D, H, W)) return 1;
  p.x_cs = chunk_stride(x); p.sc_cs = chunk_stride(sc); p.y_cs = chunk_stride(y); p.t_cs = 16;
  p.f16 = dtype == BPX_F16 ? 1 : 0;
  TileCfg c = pick_cfg(dtype, D, H, W, y.C);
  if (pool_sz) {
    BPX_CHECK(use_lean(dtype, p) && c.tx == 16, "%s: the fused pooling needs the lean bf16 kernel (bpx_conv3d_fwd_pool_supported)", fn);
    BPX_CHECK(pool_sz == 1 || pool_sz == 2, "%s: pool z stride must be 1 or 2 (got %d)", fn, pool_sz);
    BPX_CHECK(D % pool_sz == 0 && H % 2 == 0 && W % 2 == 0, "%s: extents must be divisible by the pooling window", fn);
    if (check_tensor(fn, "pooled", pooled, es, true)) return 1;
    BPX_CHECK(pooled.C == y.C, "%s: pooled.C %d != y.C %d", fn, pooled.C, y.C);
    p.pool = pooled.ptr; p.pool_ld = pooled.ld; p.pool_sz = pool_sz; p.pool_part = pool_stats_part_d;
  }
  const bool lean_ok = use_lean(dtype, p) && c.tx == 16;
  // the 16-output-channel layers of the big tile march in z (conv3d_zmarch.hip; same bits as the lean kernel); 1 = not applicable
  int rc = !lean_ok ? 1 : launch_conv3_zm(p, c, (hipStream_t)stream);
  if (rc != 0)
    rc = lean_ok ? launch_conv3_lean(EPI_FWD, p, c, (hipStream_t)stream)
           : (dtype == BPX_BF16) ? launch_conv3<uint16_t, EPI_FWD>(p, c, (hipStream_t)stream)
           : (dtype == BPX_F16)  ? launch_conv3<f16_t, EPI_FWD>(p, c, (hipStream_t)stream)
                                 : launch_conv3<float, EPI_FWD>(p, c, (hipStream_t)stream);
  BPX_CHECK(rc == 0, "%s: no kernel for tile config", fn);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

extern "C" int bpx_conv3d_fwd(int dtype, int N, int D, int H, int W, bpx_tensor x, const bpx_norm_rec* in_norm_d, int act,
                              const void* w_packed_d, const float* bias_d, bpx_tensor sc, const void* w_sc_d,
                              const float* bias_sc_d, bpx_tensor y, float* stats_part_d, bpx_stream_t stream) {
  return conv3d_fwd_impl("bpx_conv3d_fwd", dtype, N, D, H, W, x, in_norm_d, act, w_packed_d, bias_d, sc, w_sc_d, bias_sc_d, y, stats_part_d, 0,
                         bpx_tensor{nullptr, 0, 0}, nullptr, stream);
}

// Conv3d k = 3 (16 -> 16 * s^3 channels) + 3-D pixel shuffle by s in one pass: the up-scaling stage of the RCAN super-resolution network
// (biapy/models/rcan.py:317-319 `conv(filters, filters * scale**2) + nn.PixelShuffle(scale)` is 2-D only; the 3-D form - s^3 sub-positions,
// out[n, c, s z + a, s y + b, s x + e] = conv[n, c s^3 + (a s + b) s + e, z, y, x] - is defined here, see rcan.py of this package).
// w_packed_d: BPX_PK_K3 of the weight with its output channels re-ordered [sub-position][channel] (the engine does that), bias likewise.
// y: the (N, sD, sH, sW, 16) tensor.  Large volumes only (the lean kernel: D*H*W >= 32^3, W > 8).
extern "C" int bpx_conv3d_fwd_shuffle(int dtype, int N, int D, int H, int W, bpx_tensor x, const bpx_norm_rec* in_norm_d, int act, const void* w_packed_d,
                                      const float* bias_d, int s, bpx_tensor y, bpx_stream_t stream) {
  const char* fn = "bpx_conv3d_fwd_shuffle";
  BPX_CHECK(dtype == BPX_BF16 || dtype == BPX_F16, "%s: dtype must be BF16 or F16", fn);
  BPX_CHECK(s >= 2 && s <= 4, "%s: shuffle factor %d (2..4)", fn, s);
  BPX_CHECK(x.cs == 0 && y.cs == 0 && y.C == 16 && y.ld == 16 && x.C % 16 == 0, "%s: dense tensors, 16 output channels (got C=%d ld=%d)", fn, y.C, y.ld);
  if (check_tensor(fn, "x", x, 2, true) || check_tensor(fn, "y", y, 2, true)) return 1;
  BPX_CHECK(w_packed_d != nullptr, "%s: packed weights are null", fn);
  Conv3Params p{};
  p.N = N; p.D = D; p.H = H; p.W = W;
  p.x = x.ptr; p.x_ld = x.ld; p.Cin = x.C; p.in_norm = in_norm_d; p.act = act;
  p.wp = w_packed_d; p.bias = bias_d;
  p.y = y.ptr; p.y_ld = 16; p.Cout = 16 * s * s * s; p.part = nullptr;
  p.x_cs = 16; p.sc_cs = 16; p.y_cs = 16; p.t_cs = 16;
  p.f16 = dtype == BPX_F16 ? 1 : 0;
  p.ps = s;
  TileCfg c = pick_cfg(dtype, D, H, W, p.Cout);
  BPX_CHECK(use_lean(dtype, p) && c.tx == 16 && (int64_t)N * D * H * W * s * s * s * 16 < (1ll << 30), "%s: needs the lean kernel (volume >= 32^3, W > 8, output < 2 GB)", fn);
  BPX_CHECK(launch_conv3_lean(EPI_FWD, p, c, (hipStream_t)stream) == 0, "%s: no kernel for tile config", fn);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

extern "C" int bpx_conv3d_fwd_pool(int dtype, int N, int D, int H, int W, bpx_tensor x, const bpx_norm_rec* in_norm_d, int act,
                                   const void* w_packed_d, const float* bias_d, bpx_tensor sc, const void* w_sc_d,
                                   const float* bias_sc_d, bpx_tensor y, float* stats_part_d, int pool_sz, bpx_tensor pooled,
                                   float* pool_stats_part_d, bpx_stream_t stream) {
  BPX_CHECK(pooled.cs == 0, "bpx_conv3d_fwd_pool: the pooled output cannot be chunk-planar");
  return conv3d_fwd_impl("bpx_conv3d_fwd_pool", dtype, N, D, H, W, x, in_norm_d, act, w_packed_d, bias_d, sc, w_sc_d, bias_sc_d, y, stats_part_d,
                         pool_sz, pooled, pool_stats_part_d, stream);
}

extern "C" int bpx_conv3d_fwd_pool_supported(int dtype, int N, int D, int H, int W, int x_ld, int y_ld, int Cout) {
  if ((dtype != BPX_BF16 && dtype != BPX_F16) || g_use_ws != 0 || (int64_t)D * H * W < 262144 || W <= 8) return 0;
  TileCfg c = pick_cfg(dtype, D, H, W, Cout);
  return c.tx == 16 && (int64_t)N * D * H * W * std::max(x_ld, y_ld) < (1ll << 31) ? 1 : 0;
}

extern "C" int bpx_conv3d_dgrad(int dtype, int N, int D, int H, int W, bpx_tensor dy, const void* w_packed_T_d, bpx_tensor t,
                                const bpx_norm_rec* t_norm_d, int act, bpx_tensor g, float* red_part_d, bpx_stream_t stream) {
  BPX_CHECK(dy.cs == 0 && g.cs == 0, "bpx_conv3d_dgrad: only t may be chunk-planar");
  const char* fn = "bpx_conv3d_dgrad";
  BPX_CHECK(dtype == BPX_BF16 || dtype == BPX_F32 || dtype == BPX_MIX16, "%s: dtype must be BF16, F32 or MIX16 (t fp16; dy, weights, g bf16)", fn);
  const bool mix = dtype == BPX_MIX16;
  if (mix) dtype = BPX_BF16;
  int es = (int)dtype_size(dtype);
  if (check_tensor(fn, "dy", dy, es, true) || check_tensor(fn, "g", g, es, true)) return 1;
  BPX_CHECK(w_packed_T_d != nullptr, "%s: packed weights are null", fn);
  if (t_norm_d) {
    if (check_tensor(fn, "t", t, es, false)) return 1;
    BPX_CHECK(t.C == g.C, "%s: t.C %d != g.C %d", fn, t.C, g.C);
  }
  Conv3Params p{};
  p.N = N; p.D = D; p.H = H; p.W = W;
  p.x = dy.ptr; p.x_ld = dy.ld; p.Cin = dy.C; p.in_norm = nullptr; p.act = 0;
  p.wp = w_packed_T_d;
  p.y = g.ptr; p.y_ld = g.ld; p.Cout = g.C; p.part = t_norm_d ? red_part_d : nullptr;
  p.t = t.ptr; p.t_ld = t.ld; p.t_norm = t_norm_d; p.t_act = act;
  if (t_norm_d && check_planar(fn, "t", t, N, D, H, W)) return 1;
  p.x_cs = 16; p.sc_cs = 16; p.y_cs = 16; p.t_cs = chunk_stride(t);
  p.t_f16 = (mix && t_norm_d) ? 1 : 0;
  {   // buffer addressing of the LDS-DMA: every byte of t the kernel touches within 2 GB of t.ptr
    const int64_t vox = (int64_t)N * D * H * W;
    const int64_t tbytes = 2 * (t.cs ? (int64_t)t.cs * (g.C / 16 - 1) + (vox - 1) * t.ld + 16 : vox * (int64_t)t.ld);
    p.t_dma = (t_norm_d && tbytes < (1ll << 31)) ? 1 : 0;
  }
  TileCfg c = pick_cfg(dtype, D, H, W, g.C);
  const bool lean_ok = use_lean(dtype, p) && c.tx == 16 && !(c.ns >= 2 && t_norm_d && !p.t_dma);
  int rc = lean_ok ? launch_conv3_lean(EPI_DGRAD, p, c, (hipStream_t)stream)
           : p.t_f16             ? launch_conv3<uint16_t, EPI_DGRAD, f16_t>(p, c, (hipStream_t)stream)
           : (dtype == BPX_BF16) ? launch_conv3<uint16_t, EPI_DGRAD>(p, c, (hipStream_t)stream)
                               : launch_conv3<float, EPI_DGRAD>(p, c, (hipStream_t)stream);
  BPX_CHECK(rc == 0, "%s: no kernel for tile config", fn);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}
