// Conv3d 3x3x3 implicit GEMM, 16-bit storage: the "DMA-pipelined" schedule for the large layers (>= 64^3), round 3.
//
// Same GEMM mapping, LDS halo layout, packed-weight order and epilogue semantics as conv3_lp_kernel (conv3d_lean.hip - read its
// header first).  What round 2 measured about that kernel: MFMA busy 17 %, VALU 32 %, SQ_WAIT_ANY 47 % - it
// is bound by the dependent chain  global load -> normalise+ELU -> LDS -> barrier -> 14 MFMA steps  of each workgroup, i.e. by HBM
// latency that three co-resident workgroups do not cover, and a register prefetch of the next chunk does not fit its VGPR budget.
// This schedule removes the chain instead of adding residency:
//   * the halo goes global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, zero fill of out-of-volume
//     pieces by the buffer range check (offset 0x80000000 against num_records = 2^31);
//   * TWO halo buffers: the DMA of the NEXT stage (next input-channel chunk, or the first chunk of the workgroup's next tile) is
//     issued right after the barrier that starts this stage's MFMA phase and lands during it - the load latency is off the critical
//     path without holding a byte of it in registers;
//   * forward: InstanceNorm + activation are applied IN PLACE in LDS by the thread that owns the piece (ds_read -> fp32 math ->
//     ds_write); dgrad stages raw dy: no prologue, no VALU, no ds_write at all;
//   * the chunk's 14 x NS weight fragments live in registers (VMEM results retire in order: a weight load issued after the DMA of
//     the next stage would wait for that DMA, so no VMEM instruction is issued between the DMA and the end of the MFMA phase);
//     layers with ONE input chunk (16 -> 16) load them once per workgroup;
//   * 74 KB LDS -> two workgroups per CU -> 256 VGPRs per lane: the register budget that makes the two points above possible,
//     and the 4x8x16 tile for 32 output channels (NS = 2) too.
#include "conv3d_shared.h"

using namespace bpxconv;

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int TZ, int TY, int TX, int NS, int EPI, int ACTK, bool F16, bool TF16>
__global__ void __launch_bounds__(256, 2) conv3_dma_kernel(const Conv3Params p) {
  using T = typename std::conditional<F16, f16_t, uint16_t>::type;
  using TT = typename std::conditional<TF16, f16_t, uint16_t>::type;
  constexpr int KPL = 8, VB = 32;
  constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;
  constexpr int STEPS = 14, QPAD = 56;
  constexpr int MT = TZ * TY * TX / 16, MS = MT / 4;
  static_assert(TZ == 4 && MS * 16 == TY * TX && TX == 16, "wave = z-slice mapping");
  constexpr int NPIECE = HV * 2, NP = (NPIECE + 255) / 256;      // 16-byte pieces of the halo; piece idx lives at LDS byte idx*16
  constexpr int BUFB = NP * 4096;                                 // whole 1 KB wave-pieces: the DMA writes every lane's 16 bytes
  constexpr int RED_BYTES = 2 * 4 * NS * 16 * 2 * 4;
  constexpr int HSTR = HX * VB;                                   // LDS stride between m-subtiles (= tile rows) of the halo image
  constexpr int NTAB_BYTES = 2 * 16 * 2 * 4;                      // {scale, shift} of the 16 channels of a chunk, two stages
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB + RED_BYTES + NTAB_BYTES];   // ONE LDS object (a second one makes hipcc drain vmcnt before LDS reads)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int co_base = blockIdx.y * 16 * NS;
  const int Cout = p.Cout, D = p.D, H = p.H, W = p.W;

  // ---- per-workgroup constants ------------------------------------------------------------------------------------
  const int ex = j;
  const int cg_off = (g & 1) * 16;
  const bool hi_tap = (g >> 1) != 0;
  const int hb0 = ((wave * HY) * HX + ex) * VB + cg_off;
  int lbase[4] = {hb0 + (hi_tap ? VB : 0), hb0 + (hi_tap ? HX * VB : 0), hb0 + (hi_tap ? HY * HX * VB : 0), hb0};
  const int evox_rel = (wave * H) * W + ex;

  const int sub = tid & 1;
  uint32_t rel[NP], hc[NP];   // byte offset of this thread's piece u relative to the halo origin of a tile; its halo coordinates (hz | hy << 8 | hx << 16)
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int idx = u * 256 + tid;
    const int hv = idx >> 1;
    const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
    rel[u] = (uint32_t)(((hz * H + hy) * W + hx) * p.x_ld + sub * KPL) * 2u;
    hc[u] = idx < NPIECE ? ((uint32_t)hz | ((uint32_t)hy << 8) | ((uint32_t)hx << 16)) : 0xFFu;   // hz = 255: never inside the volume
  }
  const char* __restrict__ wp = reinterpret_cast<const char*>(p.wp);
  const uint32_t wlane = (uint32_t)((g * Cout + co_base + j) * KPL) * 2u;
  const int nchunks = p.Cin / 16;
  const uint32_t x_csb = (uint32_t)p.x_cs * 2u, sc_csb = (uint32_t)p.sc_cs * 2u, y_csb = (uint32_t)p.y_cs * 2u, t_csb = (uint32_t)p.t_cs * 2u;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, spx = gridDim.x >> 3;
  // every byte offset of x is < 2^31 (checked by the launcher): a piece outside the volume gets offset 2^31 = out of range = zeros
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)0x80000000u, 0x00020000);
  const bool has_norm = EPI == EPI_FWD && p.in_norm != nullptr;

  struct TileInfo { int n, tile, z0, y0, x0; };
  auto decode = [&](int local) {
    const int tileId = xcd * p.tilesPerXcd + local;
    TileInfo t;
    t.n = tileId / p.tilesPerSample; t.tile = tileId - t.n * p.tilesPerSample;
    const int txi = t.tile % p.tilesX, tyi = (t.tile / p.tilesX) % p.tilesY, tzi = t.tile / (p.tilesX * p.tilesY);
    t.z0 = tzi * TZ; t.y0 = tyi * TY; t.x0 = txi * TX;
    return t;
  };
  auto has_tile = [&](int local) { return local < p.tilesPerXcd && xcd * p.tilesPerXcd + local < p.totalTiles; };

  // DMA of one stage (tile t, input-channel chunk) into halo buffer `buf`; returns the mask of this thread's in-volume pieces
  auto issue = [&](const TileInfo& t, int chunk, int buf) {
    const uint32_t base_b = (uint32_t)(((t.n * D + t.z0 - 1) * H + (t.y0 - 1)) * W + (t.x0 - 1)) * (uint32_t)p.x_ld * 2u + (uint32_t)chunk * x_csb;
    const bool interior = t.z0 >= 1 && t.z0 + TZ + 1 <= D && t.y0 >= 1 && t.y0 + TY + 1 <= H && t.x0 >= 1 && t.x0 + TX + 1 <= W;
    uint32_t vm = 0;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      bool ok = (u < NP - 1) || (NP - 1) * 256 + tid < NPIECE;
      if (!interior) {
        const uint32_t c = hc[u];
        ok = ok && (unsigned)(t.z0 - 1 + (int)(c & 255u)) < (unsigned)D && (unsigned)(t.y0 - 1 + (int)((c >> 8) & 255u)) < (unsigned)H &&
             (unsigned)(t.x0 - 1 + (int)(c >> 16)) < (unsigned)W;
      }
      const uint32_t off = ok ? base_b + rel[u] : 0x80000000u;
      vm |= ok ? (1u << u) : 0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(smem + buf * BUFB + u * 4096 + wave * 1024), 16, off, 0, 0, 0);
    }
    return vm;
  };
  // the chunk's weight fragments: [step][NS] 16-byte operands of this lane
  u32x4_t wreg[STEPS][NS];
  auto load_w = [&](int chunk) {
    const char* wl = wp + (size_t)chunk * QPAD * Cout * 16;
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) wreg[s][ns] = *reinterpret_cast<const u32x4_t*>(wl + (size_t)s * 4 * Cout * 16 + (wlane + ns * 256u));
  };
  // InstanceNorm {scale, shift} of a stage's 16 channels travel through a small LDS table, one stage ahead: lanes 0..15 request the NEXT
  // stage's records at the top of a stage (in front of the wait for this stage's DMA, which covers them), write them behind it, and the
  // stage's barrier publishes them.  A register prefetch behind the DMA is no alternative: hipcc waits for such a load at once (vmcnt is
  // in order, so that wait would drain the DMA before the MFMA phase it is meant to overlap).
  f32x2_t* ntab = reinterpret_cast<f32x2_t*>(smem + 2 * BUFB + RED_BYTES);   // [2][16]
  auto norm_rec = [&](int n, int chunk) {
    return *reinterpret_cast<const f32x2_t*>(&p.in_norm[(size_t)n * p.Cin + chunk * 16 + (tid & 15)].scale);
  };

  if (!has_tile(slot)) return;
  // profiling (scripts/dma_stamps.py): cycle stamps of one steady-state stage of this workgroup, the last chunk of a tile
  long long* stamps = (p.stamps && tid == 0 && blockIdx.y == 0) ? p.stamps + (size_t)blockIdx.x * 16 : nullptr;
  int stamp_i = 0, it = 0;
  const int stamp_it = 2 * nchunks + nchunks - 1;
#define BPX_STAMP() do { if (stamps && it == stamp_it && stamp_i < 15) stamps[stamp_i++] = (long long)__builtin_readcyclecounter(); } while (0)
  int local = slot, chunk = 0, cbuf = 0;
  TileInfo cur = decode(local);
  uint32_t vm_cur = issue(cur, 0, 0), vm_next = 0;
  if (nchunks == 1) load_w(0);
  int sp = 0;                                        // parity of the stage: which half of the table holds its records
  if (has_norm) {
    if (tid < 16) ntab[tid] = norm_rec(cur.n, 0);
    __syncthreads();
  }

  f32x4_t acc[MS][NS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (;;) {
    // ---- what comes after this stage ---------------------------------------------------------------------------------
    int nlocal = local, nchunk = chunk + 1;
    bool hasnext = true;
    TileInfo nxt = cur;
    if (nchunk == nchunks) {
      nchunk = 0; nlocal = local + spx;
      hasnext = has_tile(nlocal);
      if (hasnext) nxt = decode(nlocal);
    }
    BPX_STAMP();                                     // 0: top of the stage
    if (nchunks > 1) load_w(chunk);                  // before the wait below: it covers them too
    f32x2_t nrec_next = f32x2_t{0.f, 0.f};
    if (has_norm && hasnext && tid < 16) nrec_next = norm_rec(nxt.n, nchunk);
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): this wave's pieces of the stage are in LDS (and the weights in registers)
    asm volatile("" ::: "memory");
    BPX_STAMP();                                     // 1: DMA of this stage (and the weights) landed
    if (has_norm) {
      if (hasnext && tid < 16) ntab[(sp ^ 1) * 16 + tid] = nrec_next;
      float nsc[KPL], nsh[KPL];
      {
        const f32x4_t* q = reinterpret_cast<const f32x4_t*>(ntab + sp * 16 + sub * KPL);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4_t v = q[e];
          nsc[2 * e] = v[0]; nsh[2 * e] = v[1]; nsc[2 * e + 1] = v[2]; nsh[2 * e + 1] = v[3];
        }
      }
      // in-place prologue: every thread transforms the pieces its own wave's DMA wrote (no barrier needed in between)
      unsigned char* hb = smem + cbuf * BUFB;
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        if ((vm_cur >> u) & 1u) {                    // zero padding applies to the ACTIVATED tensor: out-of-volume pieces stay 0
          u32x4_t v = *reinterpret_cast<const u32x4_t*>(hb + (size_t)(u * 256 + tid) * 16);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float a = fmaf(nsc[2 * i], lo16<T>(v[i]), nsh[2 * i]), b = fmaf(nsc[2 * i + 1], hi16<T>(v[i]), nsh[2 * i + 1]);
            act_pair<ACTK>(a, b, p.act);
            v[i] = pk16<T>(a, b);
          }
          *reinterpret_cast<u32x4_t*>(hb + (size_t)(u * 256 + tid) * 16) = v;
        }
      }
    }
    BPX_STAMP();                                     // 2: in-place prologue done
    __syncthreads();   // the whole halo image of this stage is in LDS; every wave is done reading the other buffer
    BPX_STAMP();                                     // 3: barrier
    if (hasnext) {
      vm_next = issue(nxt, nchunk, cbuf ^ 1);
    }
    // Operands of the epilogue, requested BEHIND the DMA and consumed after the MFMA phase (hipcc waits at the first use, i.e. there; by then
    // the DMA in front of them has landed as well): the un-normalised activation t of the dgrad epilogue, the image of the rank-1 shortcut.
    // Without this the epilogue starts with a full HBM round trip that two workgroups per CU do not cover (A/B: BPX_CONV_DBG=1 turns it off).
    const bool last_chunk = chunk == nchunks - 1;
    const bool pre_epi = last_chunk && !(p.dbg & 1);
    u32x2_t tv0[MS];
    float img0[MS];
    if (pre_epi) {
      const int n = cur.n, z0 = cur.z0, y0 = cur.y0, x0 = cur.x0;
      const bool full = z0 + TZ <= D && y0 + TY <= H && x0 + TX <= W;
      const int vox0 = ((n * D + z0) * H + y0) * W + x0 + evox_rel;
      const bool okzx = full || (z0 + wave < D && x0 + ex < W);
      const int yrem = full ? (1 << 20) : H - y0;
      if (EPI == EPI_DGRAD && p.t_norm != nullptr) {
        const char* __restrict__ tin = reinterpret_cast<const char*>(p.t);
        const uint32_t trow = (uint32_t)(W * p.t_ld) * 2u, tb = (uint32_t)(vox0 * p.t_ld + g * 4) * 2u + (uint32_t)(co_base >> 4) * t_csb;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
          tv0[ms] = u32x2_t{0u, 0u};
          if (okzx && ms < yrem) tv0[ms] = *reinterpret_cast<const u32x2_t*>(tin + (tb + ms * trow));
        }
      }
      if (EPI == EPI_FWD && p.sc != nullptr && p.sc_C == 1) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
          img0[ms] = (okzx && ms < yrem) ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.sc) + (uint32_t)(vox0 + ms * W) * 4u) : 0.f;
      }
    }
    BPX_STAMP();                                     // 4: next stage's DMA issued
    // ---- 14 MFMA steps over the staged chunk: no VMEM instruction in here ----------------------------------------------
    {
      // software pipeline over the steps (the 256-VGPR budget of two workgroups per CU pays for a second fragment set): the LDS reads of
      // step s + 1 are in flight while the MFMAs of step s issue
      const unsigned char* hb = smem + cbuf * BUFB;
      u32x4_t af[2][MS];
      auto fetch = [&](int s, u32x4_t* dst) {
        const int cls = s < 9 ? 0 : s < 12 ? 1 : s == 12 ? 2 : 3;
        const int imm = tap_off<HY, HX, VB>(bpx_tap_order_bf16(2 * s));
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) dst[ms] = *reinterpret_cast<const u32x4_t*>(hb + lbase[cls] + ms * HSTR + imm);
      };
      fetch(0, af[0]);
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        if (s + 1 < STEPS && !(p.dbg & 2)) fetch(s + 1, af[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = mfma_step<T>(wreg[s][ns], af[s & 1][ms], acc[ms][ns]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < STEPS && (p.dbg & 2)) fetch(s + 1, af[(s + 1) & 1]);   // A/B (BPX_CONV_DBG=2): no overlap of the fragment reads
      }
    }

    BPX_STAMP();                                     // 5: MFMA phase
    if (last_chunk) {
      // =================================================== epilogue of tile `cur` ===================================================
      const int n = cur.n, tile = cur.tile, z0 = cur.z0, y0 = cur.y0, x0 = cur.x0;
      __builtin_amdgcn_sched_barrier(0);
      const bool full = z0 + TZ <= D && y0 + TY <= H && x0 + TX <= W;
      const int vox0 = ((n * D + z0) * H + y0) * W + x0 + evox_rel;  // this lane's voxel for m-subtile 0
      const bool okzx = full || (z0 + wave < D && x0 + ex < W);
      const int yrem = full ? (1 << 20) : H - y0;                    // m-subtile ms (= tile row) is inside the volume iff ms < yrem
      // ---- fused 1x1x1 shortcut on a second raw tensor (EPI_FWD only): extra K steps, operands straight from global memory ----
      if (EPI == EPI_FWD && p.sc != nullptr && p.sc_C >= 16) {
        const char* __restrict__ scin = reinterpret_cast<const char*>(p.sc);
        const char* __restrict__ wsc = reinterpret_cast<const char*>(p.wsc);
        const uint32_t sb0 = (uint32_t)(vox0 * p.sc_ld) * 2u + (uint32_t)cg_off, srow = (uint32_t)(W * p.sc_ld) * 2u;
        const int nch = p.sc_C / 16;
        for (int c = 0; c < nch; ++c) {
          u32x4_t bq[MS], wf[NS];
#pragma unroll
          for (int ms = 0; ms < MS; ++ms) {
            bq[ms] = u32x4_t{0u, 0u, 0u, 0u};
            if (okzx && ms < yrem) bq[ms] = *reinterpret_cast<const u32x4_t*>(scin + (sb0 + ms * srow + (uint32_t)c * sc_csb));
          }
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) wf[ns] = *reinterpret_cast<const u32x4_t*>(wsc + (size_t)c * 4 * Cout * 16 + (wlane + ns * 256u));
#pragma unroll
          for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = mfma_step<T>(wf[ns], bq[ms], acc[ms][ns]);
        }
      }

      BPX_STAMP();                                   // 6: fused shortcut
      char* __restrict__ yout = reinterpret_cast<char*>(p.y);
      const uint32_t yrow = (uint32_t)(W * p.y_ld) * 2u;                                       // bytes between m-subtiles
      const uint32_t yb0 = (uint32_t)(vox0 * p.y_ld + g * 4) * 2u + (uint32_t)(co_base >> 4) * y_csb;
      float* red = reinterpret_cast<float*>(smem + 2 * BUFB);
      auto flush_stats = [&](int ns, const float* s1, const float* s2, int which = 0) {
        if ((which ? (float*)p.pool_part : p.part) == nullptr) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(s1[r]), b = row16_sum(s2[r]);
          if (j == 0) *reinterpret_cast<f32x2_t*>(&red[which * 4 * NS * 16 * 2 + ((wave * NS * 16) + ns * 16 + g * 4 + r) * 2]) = f32x2_t{a, b};
        }
      };

      if (EPI == EPI_FWD) {
        const bool rank1 = p.sc != nullptr && p.sc_C == 1;
        float img[MS];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
          if (pre_epi) img[ms] = rank1 ? img0[ms] : 0.f;
          else img[ms] = (rank1 && okzx && ms < yrem) ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.sc) + (uint32_t)(vox0 + ms * W) * 4u) : 0.f;
        }
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          const int co = co_base + ns * 16 + g * 4;
          f32x4_t add = f32x4_t{0.f, 0.f, 0.f, 0.f}, w1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if (p.bias) add += *reinterpret_cast<const f32x4_t*>(p.bias + co);
          if (p.sc && p.bias_sc) add += *reinterpret_cast<const f32x4_t*>(p.bias_sc + co);
          if (rank1) w1 = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p.wsc) + co);
          float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
          u32x2_t pk[MS];
#pragma unroll
          for (int ms = 0; ms < MS; ++ms) {
            pk[ms] = u32x2_t{0u, 0u};
            if (okzx && ms < yrem) {
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v[r] = acc[ms][ns][r] + add[r] + img[ms] * w1[r];
                s1[r] += v[r];
                s2[r] += v[r] * v[r];
              }
              pk[ms] = u32x2_t{pk16<T>(v[0], v[1]), pk16<T>(v[2], v[3])};
              *reinterpret_cast<u32x2_t*>(yout + (yb0 + ms * yrow + ns * y_csb)) = pk[ms];
            }
          }
          if (p.pool != nullptr) {
            // fused MaxPool3d (pool_sz,2,2) of the values just written: y pairs = two m-subtiles of this lane, x pairs = lanes j / j^1 (DPP),
            // z pairs = waves w / w+1 (through LDS, in the halo buffer this stage just finished reading)
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
              f32x4_t* exb = reinterpret_cast<f32x4_t*>(smem + cbuf * BUFB);  // [wave pair][k][lane]
              __syncthreads();                                 // the halo image (or the previous group's exchange) is no longer read
              if (wave & 1) {
#pragma unroll
                for (int k = 0; k < MS / 2; ++k) exb[((wave >> 1) * (MS / 2) + k) * 64 + lane] = f32x4_t{m[k][0], m[k][1], m[k][2], m[k][3]};
              }
              __syncthreads();
              if (!(wave & 1)) {
#pragma unroll
                for (int k = 0; k < MS / 2; ++k) {
                  const f32x4_t o = exb[((wave >> 1) * (MS / 2) + k) * 64 + lane];
#pragma unroll
                  for (int r = 0; r < 4; ++r) m[k][r] = fmaxf(m[k][r], o[r]);
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
            flush_stats(ns, q1, q2, 1);
          }
          flush_stats(ns, s1, s2);
        }
      } else {
        const bool has_t = p.t_norm != nullptr;
        const char* __restrict__ tin = reinterpret_cast<const char*>(p.t);
        const uint32_t trow = (uint32_t)(W * p.t_ld) * 2u, tb = (uint32_t)(vox0 * p.t_ld + g * 4) * 2u + (uint32_t)(co_base >> 4) * t_csb;
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          u32x2_t tv[MS];
          if (has_t) {
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) {
              if (pre_epi && ns == 0) { tv[ms] = tv0[ms]; continue; }
              tv[ms] = u32x2_t{0u, 0u};
              if (okzx && ms < yrem) tv[ms] = *reinterpret_cast<const u32x2_t*>(tin + (tb + ms * trow + ns * t_csb));
            }
          }
          float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
          if (has_t && ACTK == 1) {
            // ELU, two adjacent channels at a time in packed fp32 math (see conv3d_lean.hip): u = scale*t + shift, xhat = rstd*t - mean*rstd,
            // ELU'(u) = med3(exp(u), 0, 1), g = acc * ELU'(u), S1 += g, S2 += g * xhat
#pragma unroll
            for (int rp = 0; rp < 4; rp += 2) {
              const f32x4_t ra = *reinterpret_cast<const f32x4_t*>(&p.t_norm[(size_t)n * Cout + co_base + ns * 16 + g * 4 + rp]);
              const f32x4_t rb = *reinterpret_cast<const f32x4_t*>(&p.t_norm[(size_t)n * Cout + co_base + ns * 16 + g * 4 + rp + 1]);
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
                const bool in = okzx && ms < yrem;            // out-of-volume voxels of edge tiles carry no gradient
                const f32x2_t gv = in ? f32x2_t{acc[ms][ns][rp], acc[ms][ns][rp + 1]} * a : f32x2_t{0.f, 0.f};
                acc[ms][ns][rp] = gv[0]; acc[ms][ns][rp + 1] = gv[1];
                s1p = s1p + gv;
                s2p = __builtin_elementwise_fma(gv, xh, s2p);
              }
              s1[rp] = s1p[0]; s1[rp + 1] = s1p[1]; s2[rp] = s2p[0]; s2[rp + 1] = s2p[1];
            }
          } else if (has_t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const f32x4_t rec = *reinterpret_cast<const f32x4_t*>(&p.t_norm[(size_t)n * Cout + co_base + ns * 16 + g * 4 + r]);
#pragma unroll
              for (int ms = 0; ms < MS; ++ms) {
                const uint32_t w = tv[ms][r >> 1];
                const float tf = (r & 1) ? hi16<TT>(w) : lo16<TT>(w);
                const float u = fmaf(rec[2], tf, rec[3]);
                const float gv = (okzx && ms < yrem) ? acc[ms][ns][r] * apply_act_bwd_rt<T, ACTK>(u, p.t_act) : 0.f;
                acc[ms][ns][r] = gv;
                s1[r] += gv;
                s2[r] += gv * ((tf - rec[0]) * rec[1]);
              }
            }
          }
#pragma unroll
          for (int ms = 0; ms < MS; ++ms)
            if (okzx && ms < yrem)
              *reinterpret_cast<u32x2_t*>(yout + (yb0 + ms * yrow + ns * y_csb)) =
                  u32x2_t{pk16<T>(acc[ms][ns][0], acc[ms][ns][1]), pk16<T>(acc[ms][ns][2], acc[ms][ns][3])};
          flush_stats(ns, s1, s2);
        }
      }

      BPX_STAMP();                                   // 7: epilogue math + stores issued
      // ---- statistics partials: 4 waves (LDS) -> global [n][tile][2][Cout] ------------------------------------------------
      if (p.part != nullptr || (EPI == EPI_FWD && p.pool_part != nullptr)) {
        __syncthreads();
        if (tid < 2 * NS * 16 * 2) {
          const int which = tid / (NS * 16 * 2), q = tid % (NS * 16 * 2);
          const int c = q >> 1, k = q & 1;
          float* dst = which ? (EPI == EPI_FWD ? p.pool_part : nullptr) : p.part;
          if (dst != nullptr) {
            const float* rd = red + which * 4 * NS * 16 * 2;
            const float a = rd[(0 * NS * 16 + c) * 2 + k] + rd[(1 * NS * 16 + c) * 2 + k] + rd[(2 * NS * 16 + c) * 2 + k] + rd[(3 * NS * 16 + c) * 2 + k];
            dst[(((size_t)n * p.tilesPerSample + tile) * 2 + k) * Cout + co_base + c] = a;
          }
        }
      }
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    BPX_STAMP();                                     // 8: stage done
    if (!hasnext) break;
    ++it;
    // ---- advance: the buffers swap roles (every ds_read base moves by +-BUFB) -----------------------------------------------
    cbuf ^= 1; sp ^= 1; vm_cur = vm_next; chunk = nchunk; local = nlocal; cur = nxt;
  }
#undef BPX_STAMP
}

int cu_count_dma() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

template <int EPI>
int launch_dma(const Conv3Params& p0, const TileCfg& c, hipStream_t s) {
  Conv3Params p = p0;
  const int tilesZ = cdiv(p.D, c.tz);
  p.tilesY = cdiv(p.H, c.ty);
  p.tilesX = cdiv(p.W, c.tx);
  p.tilesPerSample = tilesZ * p.tilesY * p.tilesX;
  p.totalTiles = p.N * p.tilesPerSample;
  p.tilesPerXcd = cdiv(p.totalTiles, 8);
  p.stamps = g_conv_stamps;
  { static const char* e = getenv("BPX_CONV_DBG"); p.dbg = e ? atoi(e) : 0; }
  const int gy = p.Cout / (16 * c.ns);
  const bool elu = (EPI == EPI_FWD ? p.act : p.t_act) == BPX_ACT_ELU;
  int gx = std::max(8, (cu_count_dma() * 2 / gy) & ~7);
  gx = std::min(gx, 8 * p.tilesPerXcd);
  dim3 grid((unsigned)gx, (unsigned)gy);
#define L(TY, NS)                                                                                      \
  if (c.tz == 4 && c.ty == TY && c.tx == 16 && c.ns == NS) {                                           \
    if (p.f16) {                                                                                       \
      if constexpr (EPI == EPI_FWD) {                                                                  \
        if (elu) conv3_dma_kernel<4, TY, 16, NS, EPI_FWD, 1, true, true><<<grid, 256, 0, s>>>(p);       \
        else conv3_dma_kernel<4, TY, 16, NS, EPI_FWD, 0, true, true><<<grid, 256, 0, s>>>(p);           \
        return 0;                                                                                      \
      }                                                                                                \
      return 1;                                                                                        \
    }                                                                                                  \
    if (p.t_f16) {                                                                                     \
      if constexpr (EPI == EPI_DGRAD) {                                                                \
        if (elu) conv3_dma_kernel<4, TY, 16, NS, EPI_DGRAD, 1, false, true><<<grid, 256, 0, s>>>(p);    \
        else conv3_dma_kernel<4, TY, 16, NS, EPI_DGRAD, 0, false, true><<<grid, 256, 0, s>>>(p);        \
        return 0;                                                                                      \
      }                                                                                                \
      return 1;                                                                                        \
    }                                                                                                  \
    if (elu) conv3_dma_kernel<4, TY, 16, NS, EPI, 1, false, false><<<grid, 256, 0, s>>>(p);             \
    else conv3_dma_kernel<4, TY, 16, NS, EPI, 0, false, false><<<grid, 256, 0, s>>>(p);                 \
    return 0;                                                                                          \
  }
  L(8, 1)   // NS = 2 (32 output channels, 4x4x16 tile) compiles with scratch at 256 VGPRs (112 weight registers): those layers stay on conv3_lp_kernel
#undef L
  return 1;
}

}  // namespace

namespace bpxconv {
int g_conv_dma = 1;   // the DMA-pipelined kernel where it applies (bpx_debug_set_conv_ws: 7 = off, 6 = on)

// byte offsets < 2^31 for the operand that goes through the buffer descriptor; 4x8x16 tiles, 16 output channels
bool conv3_dma_applies(const Conv3Params& p, const TileCfg& c) {
  const int64_t vox = (int64_t)p.N * p.D * p.H * p.W;
  const int64_t xbytes = (p.x_cs == 16 ? vox * p.x_ld : (int64_t)p.x_cs * (p.Cin / 16)) * 2;
  return g_conv_dma != 0 && c.tz == 4 && c.tx == 16 && c.ty == 8 && c.ns == 1 && xbytes < (1ll << 31);
}
int launch_conv3_dma(int epi, const Conv3Params& p, const TileCfg& c, hipStream_t s) {
  return epi == EPI_FWD ? launch_dma<EPI_FWD>(p, c, s) : launch_dma<EPI_DGRAD>(p, c, s);
}
}  // namespace bpxconv
