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
__global__ void __launch_bounds__(256, TY == 8 ? 2 : 3) conv3_dma_kernel(const Conv3Params p) {
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
  // Halo coordinates of this thread's pieces: piece u is halo voxel (u * 256 + tid) / 2, i.e. 128 voxels = (7 rows, 2 columns) past piece
  // u - 1.  Only piece 0's coordinates are kept; a stage walks them forward with two conditional wraps per piece (the kernel is bound by
  // the number of instructions a wave issues - ~4 cycles each whatever the unit - so neither divisions per piece nor nine more
  // loop-invariant registers are affordable in the forward instances)
  static_assert(128 / HX == 7 && 128 % HX == 2 && HY >= 6 && HY <= 15, "piece walk: +2 columns, +7 rows (+1 with the column wrap), then rows modulo HY");
  const int hv0 = tid >> 1;
  const int hx0 = hv0 % HX, hy0 = (hv0 / HX) % HY, hz0 = hv0 / (HX * HY);
  const uint32_t ld2 = (uint32_t)p.x_ld * 2u, sub16 = (uint32_t)sub * 16u;
  const char* __restrict__ wp = reinterpret_cast<const char*>(p.wp);
  const uint32_t wlane = (uint32_t)((g * Cout + co_base + j) * KPL) * 2u;
  const int nchunks = p.Cin / 16;
  const uint32_t x_csb = (uint32_t)p.x_cs * 2u, sc_csb = (uint32_t)p.sc_cs * 2u, y_csb = (uint32_t)p.y_cs * 2u, t_csb = (uint32_t)p.t_cs * 2u;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, spx = gridDim.x >> 3;
  // every byte offset of x is < 2^31 (checked by the launcher): a piece outside the volume gets offset 2^31 = out of range = zeros
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)0x80000000u, 0x00020000);
  // The epilogue's operands (t of the dgrad, the shortcut tensor / image) go through buffer descriptors too: a voxel outside the volume is an
  // out-of-range OFFSET (-> zeros) instead of a select on the loaded DATA - hipcc answers `cond ? load : 0` with a branch around the load and
  // an immediate vmcnt(0) per element, which would drain the DMA queue in front of the MFMA phase
  const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(EPI == EPI_DGRAD ? p.t : p.sc), 0, (int)0x80000000u, 0x00020000);
  constexpr uint32_t OOB = 0x80000000u;
  const bool has_norm = EPI == EPI_FWD && p.in_norm != nullptr;

  // Workgroup (xcd, slot) owns the CONTIGUOUS run of tiles [slot * run, (slot + 1) * run) of its XCD's range (x fastest, then y, z, sample):
  // the next tile follows from the current one by counter increments (a division-based decode per tile was ~150 scalar instructions), and
  // consecutive tiles of a workgroup share a halo face that is still in the XCD's L2
  struct TileInfo { int n, tile, z0, y0, x0; };
  auto decode = [&](int local) {
    const int tileId = xcd * p.tilesPerXcd + local;
    TileInfo t;
    t.n = tileId / p.tilesPerSample; t.tile = tileId - t.n * p.tilesPerSample;
    const int txi = t.tile % p.tilesX, tyi = (t.tile / p.tilesX) % p.tilesY, tzi = t.tile / (p.tilesX * p.tilesY);
    t.z0 = tzi * TZ; t.y0 = tyi * TY; t.x0 = txi * TX;
    return t;
  };
  auto advance = [&](const TileInfo& c) {
    TileInfo t = c;
    t.tile = c.tile + 1; t.x0 = c.x0 + TX;
    if (t.x0 >= p.tilesX * TX) {
      t.x0 = 0; t.y0 = c.y0 + TY;
      if (t.y0 >= p.tilesY * TY) {
        t.y0 = 0; t.z0 = c.z0 + TZ;
        if (t.tile >= p.tilesPerSample) { t.z0 = 0; t.tile = 0; t.n = c.n + 1; }
      }
    }
    return t;
  };
  const int run = (p.tilesPerXcd + spx - 1) / spx;
  const int local_end = min(min((slot + 1) * run, p.tilesPerXcd), p.totalTiles - xcd * p.tilesPerXcd);
  auto is_interior = [&](const TileInfo& t) {
    return t.z0 >= 1 && t.z0 + TZ + 1 <= D && t.y0 >= 1 && t.y0 + TY + 1 <= H && t.x0 >= 1 && t.x0 + TX + 1 <= W;
  };

  // DMA of one stage (tile t, input-channel chunk) into halo buffer `buf`, one 16-byte piece per lane and call; returns whether this thread's
  // piece u lies inside the volume.  The pieces of the NEXT stage are issued one per MFMA step of the current one (an LDS-DMA instruction
  // costs its wave 100-250 issue cycles: nine of them in a row were 2 K cycles per stage, scripts/dma_stamps.py)
  struct StageBase { uint32_t base_b; int zm, ym, xm; bool interior; int hx, hy, hz; };
  auto stage_base = [&](const TileInfo& t, int chunk) {
    StageBase b;
    b.base_b = (uint32_t)(((t.n * D + t.z0 - 1) * H + (t.y0 - 1)) * W + (t.x0 - 1)) * (uint32_t)p.x_ld * 2u + (uint32_t)chunk * x_csb;
    b.zm = t.z0 - 1; b.ym = t.y0 - 1; b.xm = t.x0 - 1;
    b.interior = is_interior(t);
    b.hx = hx0; b.hy = hy0; b.hz = hz0;
    return b;
  };
  auto issue_piece = [&](StageBase& b, int buf, int u) {      // pieces in order u = 0, 1, ...: b carries the walking coordinates
    bool ok = (u < NP - 1) || (NP - 1) * 256 + tid < NPIECE;
    const int hx = b.hx, hy = b.hy, hz = b.hz;
    // bitwise, not short-circuit: straight-line code instead of an exec-mask branch per comparison
    const bool in = ((unsigned)(b.zm + hz) < (unsigned)D) & ((unsigned)(b.ym + hy) < (unsigned)H) & ((unsigned)(b.xm + hx) < (unsigned)W);
    ok = ok & (b.interior | in);
    const uint32_t relb = (uint32_t)((hz * H + hy) * W + hx) * ld2 + sub16;
    {   // next piece: 128 halo voxels further
      int nx = hx + 2, ny = hy + 7;
      const bool wx = nx >= HX;
      nx = wx ? nx - HX : nx; ny = wx ? ny + 1 : ny;
      int nz = hz;
      if (HY < 8) { const bool w0 = ny >= 2 * HY; ny = w0 ? ny - HY : ny; nz = w0 ? nz + 1 : nz; }   // 4x4x16 tile: HY = 6, up to two wraps
      const bool wy = ny >= HY;
      b.hx = nx; b.hy = wy ? ny - HY : ny; b.hz = wy ? nz + 1 : nz;
    }
    const uint32_t off = ok ? b.base_b + relb : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(smem + buf * BUFB + u * 4096 + wave * 1024), 16, off, 0, 0, 0);
    return ok ? (1u << u) : 0u;
  };
  auto issue = [&](const TileInfo& t, int chunk, int buf) {
    StageBase b = stage_base(t, chunk);
    uint32_t vm = 0;
#pragma unroll
    for (int u = 0; u < NP; ++u) vm |= issue_piece(b, buf, u);
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
  float* ntab = reinterpret_cast<float*>(smem + 2 * BUFB + RED_BYTES);   // [2 stages][scale | shift][16]: pairs of channels are adjacent (v_pk_fma)
  auto norm_rec = [&](int n, int chunk) {
    return *reinterpret_cast<const f32x2_t*>(&p.in_norm[(size_t)n * p.Cin + chunk * 16 + (tid & 15)].scale);
  };

  if (slot * run >= local_end) return;
  // profiling (scripts/dma_stamps.py): cycle stamps of one steady-state stage of this workgroup, the last chunk of a tile
  long long* stamps = (p.stamps && tid == 0 && blockIdx.y == 0) ? p.stamps + (size_t)blockIdx.x * 16 : nullptr;
  int stamp_i = 0, it = 0;
  const int stamp_it = 2 * nchunks + nchunks - 1;
#define BPX_STAMP() do { if (stamps && it == stamp_it && stamp_i < 15) stamps[stamp_i++] = (long long)__builtin_readcyclecounter(); } while (0)
  int local = slot * run, chunk = 0, cbuf = 0;
  TileInfo cur = decode(local);
  uint32_t vm_cur = issue(cur, 0, 0), vm_next = 0;
  int sp = 0;                                        // parity of the stage: which half of the table holds its records
  if (has_norm) {
    if (tid < 16) { const f32x2_t r = norm_rec(cur.n, 0); ntab[tid] = r[0]; ntab[16 + tid] = r[1]; }
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
      nchunk = 0; nlocal = local + 1;
      hasnext = nlocal < local_end;
      if (hasnext) nxt = advance(cur);
    }
    const bool int_cur = is_interior(cur);
    BPX_STAMP();                                     // 0: top of the stage
    load_w(chunk);                                   // every stage (L1 / L2 hits), in front of the wait below, which covers them: the 56 registers are
                                                     // then free during the epilogue, which needs them for the shortcut operands
    f32x2_t nrec_next = f32x2_t{0.f, 0.f};
    if (has_norm && hasnext && tid < 16) nrec_next = norm_rec(nxt.n, nchunk);
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): this wave's pieces of the stage are in LDS (and the weights in registers)
    asm volatile("" ::: "memory");
    BPX_STAMP();                                     // 1: DMA of this stage (and the weights) landed
    if (has_norm) {
      if (hasnext && tid < 16) { ntab[(sp ^ 1) * 32 + tid] = nrec_next[0]; ntab[(sp ^ 1) * 32 + 16 + tid] = nrec_next[1]; }
      f32x2_t sc2[4], sh2[4];
      {
        const f32x4_t* q = reinterpret_cast<const f32x4_t*>(ntab + sp * 32 + sub * KPL);
        const f32x4_t a0 = q[0], a1 = q[1], b0 = q[4], b1 = q[5];
        sc2[0] = f32x2_t{a0[0], a0[1]}; sc2[1] = f32x2_t{a0[2], a0[3]}; sc2[2] = f32x2_t{a1[0], a1[1]}; sc2[3] = f32x2_t{a1[2], a1[3]};
        sh2[0] = f32x2_t{b0[0], b0[1]}; sh2[1] = f32x2_t{b0[2], b0[3]}; sh2[2] = f32x2_t{b1[0], b1[1]}; sh2[3] = f32x2_t{b1[2], b1[3]};
      }
      // in-place prologue: every thread transforms the pieces its own wave's DMA wrote (no barrier needed in between).  Three pieces per
      // round: the reads of a round are issued together and the math is branch-free (one piece at a time under an exec branch ran at one
      // LDS round trip + 35 dependent VALU instructions per piece: 4.5 K cycles per stage, scripts/dma_stamps.py); only the write-back is
      // predicated - zero padding applies to the ACTIVATED tensor, so out-of-volume pieces keep the zeros the DMA wrote
      unsigned char* hb = smem + cbuf * BUFB;
      constexpr int RB = 3;
#pragma unroll
      for (int u0 = 0; u0 < NP; u0 += RB) {
        u32x4_t v[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q)
          if (u0 + q < NP) v[q] = *reinterpret_cast<const u32x4_t*>(hb + (size_t)((u0 + q) * 256 + tid) * 16);
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          if (u0 + q >= NP) continue;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x2_t x2{lo16<T>(v[q][i]), hi16<T>(v[q][i])};
            const f32x2_t a2 = __builtin_elementwise_fma(sc2[i], x2, sh2[i]);
            float a = a2[0], b = a2[1];
            act_pair<ACTK>(a, b, p.act);
            v[q][i] = pk16<T>(a, b);
          }
        }
        if (int_cur) {             // every piece lies inside the volume (61 % of the 128^3 tiles): no exec-mask juggling around the write-back
#pragma unroll
          for (int q = 0; q < RB; ++q)
            if (u0 + q < NP) *reinterpret_cast<u32x4_t*>(hb + (size_t)((u0 + q) * 256 + tid) * 16) = v[q];
        } else {
#pragma unroll
          for (int q = 0; q < RB; ++q)
            if (u0 + q < NP && ((vm_cur >> (u0 + q)) & 1u)) *reinterpret_cast<u32x4_t*>(hb + (size_t)((u0 + q) * 256 + tid) * 16) = v[q];
        }
      }
    }
    BPX_STAMP();                                     // 2: in-place prologue done
    __syncthreads();   // the whole halo image of this stage is in LDS; every wave is done reading the other buffer
    BPX_STAMP();                                     // 3: barrier
    // Operands of the epilogue, requested at the top of the MFMA phase and consumed after it (hipcc waits at the first use, i.e. there; by then
    // the DMA pieces issued behind them have landed as well): the un-normalised activation t of the dgrad epilogue, the image of the
    // rank-1 shortcut, the first chunk of a 1x1x1 shortcut.  Without this the epilogue starts with a full HBM round trip that two workgroups
    // per CU do not cover (A/B: BPX_CONV_DBG=1 turns it off).  16-byte accesses: lane (j, g) fetches voxel row 2k + (g & 1), channels
    // (g >> 1) * 8 .. + 7 and v_permlane16_swap moves the halves to the MFMA layout (4 consecutive channels of rows 2k and 2k + 1).
    const bool last_chunk = chunk == nchunks - 1;
    const bool pre_epi = last_chunk;
    const int gh = g >> 1, go = g & 1;
    u32x4_t tq0[MS / 2];
    float img0[MS];
    u32x4_t bq0[MS];
    const bool sc_mm = EPI == EPI_FWD && p.sc != nullptr && p.sc_C >= 16;
    if (pre_epi) {
      const int n = cur.n, z0 = cur.z0, y0 = cur.y0, x0 = cur.x0;
      const bool full = z0 + TZ <= D && y0 + TY <= H && x0 + TX <= W;
      const int vox0 = ((n * D + z0) * H + y0) * W + x0 + evox_rel;
      const bool okzx = full || (z0 + wave < D && x0 + ex < W);
      const int yrem = full ? (1 << 20) : H - y0;
      if (EPI == EPI_DGRAD && p.t_norm != nullptr) {
        const uint32_t trow = (uint32_t)(W * p.t_ld) * 2u, tb = (uint32_t)(vox0 * p.t_ld + gh * 8) * 2u + (uint32_t)(co_base >> 4) * t_csb;
#pragma unroll
        for (int k = 0; k < MS / 2; ++k)
          tq0[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, (okzx && 2 * k + go < yrem) ? tb + (2 * k + go) * trow : OOB, 0, 0);
      }
      if (EPI == EPI_FWD && p.sc != nullptr && p.sc_C == 1) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
          img0[ms] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_t, (okzx && ms < yrem) ? (uint32_t)(vox0 + ms * W) * 4u : OOB, 0, 0));
      }
      if (sc_mm) {
        const uint32_t sb0 = (uint32_t)(vox0 * p.sc_ld) * 2u + (uint32_t)cg_off, srow = (uint32_t)(W * p.sc_ld) * 2u;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) bq0[ms] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, (okzx && ms < yrem) ? sb0 + ms * srow : OOB, 0, 0);
      }
    }
    BPX_STAMP();                                     // 4: epilogue operands requested
    // ---- 14 MFMA steps over the staged chunk; one DMA piece of the next stage behind the MFMAs of each of the first NP steps ----------
    {
      StageBase nb{};
      if (hasnext) nb = stage_base(nxt, nchunk);
      vm_next = 0;
      const unsigned char* hb = smem + cbuf * BUFB;
      static_assert(NP <= STEPS, "one DMA piece per MFMA step");
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int cls = s < 9 ? 0 : s < 12 ? 1 : s == 12 ? 2 : 3;
        const int imm = tap_off<HY, HX, VB>(bpx_tap_order_bf16(2 * s));
        u32x4_t af[MS];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) af[ms] = *reinterpret_cast<const u32x4_t*>(hb + lbase[cls] + ms * HSTR + imm);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0) once per step instead of one counted wait in front of every MFMA (112 -> 14 issue slots)
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = mfma_step<T>(wreg[s][ns], af[ms], acc[ms][ns]);
        if (s < NP && hasnext) vm_next |= issue_piece(nb, cbuf ^ 1, s);
        __builtin_amdgcn_sched_barrier(0);
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
        const char* __restrict__ wsc = reinterpret_cast<const char*>(p.wsc);
        const uint32_t sb0 = (uint32_t)(vox0 * p.sc_ld) * 2u + (uint32_t)cg_off, srow = (uint32_t)(W * p.sc_ld) * 2u;
        const int nch = p.sc_C / 16;
        // chunk 0 was requested before the MFMA phase (bq0); chunk c + 1 is requested before the MFMAs of chunk c and chunk c + 2 right behind
        // them, so that the HBM round trips of a 48-channel shortcut overlap instead of following each other (they were 8.6 K cycles per tile)
        auto sc_fetch = [&](int c, u32x4_t* dst) {
#pragma unroll
          for (int ms = 0; ms < MS; ++ms)
            dst[ms] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, (okzx && ms < yrem) ? sb0 + ms * srow + (uint32_t)c * sc_csb : OOB, 0, 0);
        };
        // the weights of chunk c + 1 are requested IN FRONT of its operands: the counted wait for chunk c's MFMAs then leaves both in flight
        auto sc_w = [&](int c, u32x4_t* wf) {
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) wf[ns] = *reinterpret_cast<const u32x4_t*>(wsc + (size_t)c * 4 * Cout * 16 + (wlane + ns * 256u));
        };
        auto sc_mfma = [&](const u32x4_t* wf, const u32x4_t* src) {
#pragma unroll
          for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = mfma_step<T>(wf[ns], src[ms], acc[ms][ns]);
        };
        u32x4_t bqb[MS], wfa[NS], wfb[NS];
        sc_w(0, wfa);
        for (int c = 0; c < nch; c += 2) {
          if (c + 1 < nch) { sc_w(c + 1, wfb); sc_fetch(c + 1, bqb); }
          sc_mfma(wfa, bq0);
          if (c + 2 < nch) { sc_w(c + 2, wfa); sc_fetch(c + 2, bq0); }
          if (c + 1 < nch) sc_mfma(wfb, bqb);
        }
      }

      BPX_STAMP();                                   // 6: fused shortcut
      char* __restrict__ yout = reinterpret_cast<char*>(p.y);
      const uint32_t yrow = (uint32_t)(W * p.y_ld) * 2u;                                       // bytes between m-subtiles
      // 16-byte stores: after v_permlane16_swap lane (j, g) holds the 8 channels (g >> 1) * 8 .. + 7 of voxel row 2k + (g & 1) - half the store
      // instructions of the 8-byte MFMA layout (the store tail is issue-bound: 8 dwordx2 per lane were ~4.5 K cycles per tile)
      const uint32_t yb0 = (uint32_t)(vox0 * p.y_ld + gh * 8) * 2u + (uint32_t)(co_base >> 4) * y_csb;
      auto store_rows = [&](int ns, const u32x2_t* pk) {
#pragma unroll
        for (int k = 0; k < MS / 2; ++k) {
          const auto r0 = __builtin_amdgcn_permlane16_swap(pk[2 * k][0], pk[2 * k + 1][0], false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(pk[2 * k][1], pk[2 * k + 1][1], false, false);
          if (okzx && 2 * k + go < yrem)
            *reinterpret_cast<u32x4_t*>(yout + (yb0 + (2 * k + go) * yrow + ns * y_csb)) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
        }
      };
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
        for (int ms = 0; ms < MS; ++ms) img[ms] = rank1 ? img0[ms] : 0.f;
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
            }
          }
          store_rows(ns, pk);
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
        const uint32_t trow = (uint32_t)(W * p.t_ld) * 2u, tb = (uint32_t)(vox0 * p.t_ld + gh * 8) * 2u + (uint32_t)(co_base >> 4) * t_csb;
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          u32x2_t tv[MS];
          if (has_t) {
#pragma unroll
            for (int k = 0; k < MS / 2; ++k) {
              u32x4_t q;
              if (ns == 0) q = tq0[k];
              else q = __builtin_amdgcn_raw_buffer_load_b128(rs_t, (okzx && 2 * k + go < yrem) ? tb + (2 * k + go) * trow + ns * t_csb : OOB, 0, 0);
              // loaded: channels (g >> 1) * 8 .. + 7 of row 2k + (g & 1); wanted: channels 4g .. 4g + 3 of rows 2k and 2k + 1
              const auto r0 = __builtin_amdgcn_permlane16_swap(q[0], q[2], false, false);
              const auto r1 = __builtin_amdgcn_permlane16_swap(q[1], q[3], false, false);
              tv[2 * k] = u32x2_t{r0[0], r1[0]};
              tv[2 * k + 1] = u32x2_t{r0[1], r1[1]};
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
          u32x2_t gk[MS];
#pragma unroll
          for (int ms = 0; ms < MS; ++ms) gk[ms] = u32x2_t{pk16<T>(acc[ms][ns][0], acc[ms][ns][1]), pk16<T>(acc[ms][ns][2], acc[ms][ns][3])};
          store_rows(ns, gk);
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
  int gx = std::max(8, (cu_count_dma() * (c.ty == 8 ? 2 : 3) / gy) & ~7);
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
  L(8, 1) L(4, 1)   // NS = 2 (32 output channels, 4x4x16 tile) compiles with scratch at 256 VGPRs (112 weight registers): those layers stay on conv3_lp_kernel
#undef L
  return 1;
}

}  // namespace

namespace bpxconv {
int g_conv_dma = 0;   // the DMA-pipelined kernel is an A/B variant (bpx_debug_set_conv_ws: 6 = on, 7 = off): measured equal to the lean kernel within +-4 %, DESIGN.md section 6

// byte offsets < 2^31 for the operand that goes through the buffer descriptor; 4x8x16 tiles, 16 output channels
bool conv3_dma_applies(const Conv3Params& p, const TileCfg& c) {
  const int64_t vox = (int64_t)p.N * p.D * p.H * p.W;
  const int64_t xbytes = (p.x_cs == 16 ? vox * p.x_ld : (int64_t)p.x_cs * (p.Cin / 16)) * 2;
  const int64_t tbytes = p.t ? (p.t_cs == 16 ? vox * p.t_ld : (int64_t)p.t_cs * (p.Cout / 16)) * 2 : 0;
  const int64_t sbytes = p.sc ? (p.sc_C == 1 ? vox * 4 : (p.sc_cs == 16 ? vox * p.sc_ld : (int64_t)p.sc_cs * (p.sc_C / 16)) * 2) : 0;
  return g_conv_dma != 0 && c.tz == 4 && c.tx == 16 && (c.ty == 8 || c.ty == 4) && c.ns == 1 && std::max(xbytes, std::max(tbytes, sbytes)) < (1ll << 31);
}
int launch_conv3_dma(int epi, const Conv3Params& p, const TileCfg& c, hipStream_t s) {
  return epi == EPI_FWD ? launch_dma<EPI_FWD>(p, c, s) : launch_dma<EPI_DGRAD>(p, c, s);
}
}  // namespace bpxconv
