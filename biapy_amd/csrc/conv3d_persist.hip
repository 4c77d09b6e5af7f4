// Conv3d 3x3x3 implicit GEMM - persistent 4-wave kernel with a cross-tile stage pipeline (bf16 production kernel).
//
// Built from the cycle anatomy of the one-tile-per-workgroup kernel (scripts/conv_stamps.py, 48->16 @128^3, cycles per
// workgroup): index math 3.9k | first chunk load+transform 11.4k | step loops 9.2k+7.1k+3.4k | epilogue 6.0k | total 42.7k.
// Half of a workgroup's life was per-TILE latency with the matrix pipe idle: the prologue's HBM round trip, integer
// division chains, bias / statistics traffic in the epilogue.  Here a workgroup is persistent (tiles b, b+G, b+2G, ...)
// and everything is a stage of ONE software pipeline over the flattened (tile, channel-chunk) sequence:
//
//   stage q   : MFMA step loop on LDS buffer q&1 ........................ (14 steps x MS x NS v_mfma_f32_16x16x32_bf16)
//               between steps: piece u of stage q+1 (raw, in registers since stage q-1) is normalised + activated (fp32),
//               converted (v_cvt_pk_bf16_f32) and written to buffer (q+1)&1; its registers are re-loaded with stage q+2
//   per tile  : epilogue operands (bias, shortcut image values, dgrad's t values) are fetched BEFORE the tile's last
//               step loop; statistics are accumulated in registers across tiles and flushed once per (workgroup, sample).
//   one s_barrier per stage; no HBM latency, address arithmetic or reduction on the critical path in steady state.
#include "conv3d_shared.h"

using namespace bpxconv;

namespace {

template <typename T, int TZ, int TY, int TX, int NS, int EPI, int ACTK>
__global__ void __launch_bounds__(256, 2) conv3_p_kernel(const Conv3Params p) {
  using Tr = ElemTraits<T>;
  constexpr int KPL = Tr::KPL, GPT = 16 / KPL, VB = 16 * (int)sizeof(T);
  constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX, TV = TZ * TY * TX;
  constexpr int QTOT = 27 * GPT, STEPS = (QTOT + 3) / 4, QPAD = STEPS * 4;
  constexpr int MT = TV / 16, MS = MT / 4;
  static_assert(MT % 4 == 0 && MS >= 1, "tile must give every wave at least one m-subtile");
  constexpr int BUFB = HV * VB;
  constexpr int NPM = (HV * GPT + 255) / 256;  // 16-byte pieces per thread, halo geometry
  constexpr int NPS = (TV * GPT + 255) / 256;  // ... shortcut (halo-free) geometry
  static_assert(NPM <= STEPS, "one staged piece per MFMA step");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB + (EPI == EPI_DGRAD ? NS * 16 * 16 : 0)];
  bpx_norm_rec* srec = reinterpret_cast<bpx_norm_rec*>(smem + 2 * BUFB);  // dgrad: records of t for this sample / channel block

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int nchunks = p.Cin / 16;
  const int nsc = (EPI == EPI_FWD && p.sc != nullptr && p.sc_C >= 16) ? p.sc_C / 16 : 0;
  const int S = nchunks + nsc;  // stages per tile
  const int ntl = (p.totalTiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nstage = ntl * S;
  const int co_base = blockIdx.y * 16 * NS;
  const int Cout = p.Cout;
  const int sub = tid % GPT;

  // ---------------------------------------------------------------- staging state (stage q+1 in pbuf, q+2 being loaded)
  u32x4_t pbuf[NPM];
  uint32_t goff[NPM];
  float psc[KPL], psh[KPL];       // norm records of the stage held in pbuf (re-loaded at the stage boundary)
  const bpx_norm_rec* nrec_next = nullptr;
  bool cur_norm = false, cur_main = true, nxt_norm = false, nxt_main = true;
  uint32_t cur_valid = 0, nxt_valid = 0;  // bit u: piece u lies inside the volume
  int ln = 0, lz0 = 0, ly0 = 0, lx0 = 0;  // tile of the stage being loaded
  const T* lsrc = nullptr;

  // geometry + source pointer + norm records of stage q (the one about to be LOADED)
  auto begin_load = [&](int q) {
    const int s = q % S;
    if (s == 0) {
      const int tt = (int)blockIdx.x + (q / S) * (int)gridDim.x;
      const int tile = tt % p.tilesPerSample;
      ln = tt / p.tilesPerSample;
      lx0 = (tile % p.tilesX) * TX; ly0 = ((tile / p.tilesX) % p.tilesY) * TY; lz0 = (tile / (p.tilesX * p.tilesY)) * TZ;
    }
    const bool main = s < nchunks;
    if (s == 0 || s == nchunks) {  // geometry changes: piece -> voxel map
      nxt_valid = 0;
#pragma unroll
      for (int u = 0; u < NPM; ++u) {
        const int idx = u * 256 + tid;
        goff[u] = 0xFFFFFFFFu;
        int gz, gy, gx;
        bool in;
        if (main) {
          const int hv = idx / GPT;
          gz = lz0 - 1 + hv / (HX * HY); gy = ly0 - 1 + (hv / HX) % HY; gx = lx0 - 1 + hv % HX;
          in = idx < HV * GPT;
        } else {
          const int t = idx / GPT;
          gz = lz0 + t / (TY * TX); gy = ly0 + (t / TX) % TY; gx = lx0 + t % TX;
          in = (u < NPS) && idx < TV * GPT;
        }
        if (in && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
          goff[u] = (uint32_t)((((uint32_t)gz * p.H + gy) * p.W + gx) * (uint32_t)(main ? p.x_ld : p.sc_ld));
          nxt_valid |= 1u << u;
        }
      }
    }
    lsrc = main ? reinterpret_cast<const T*>(p.x) + (size_t)ln * p.D * p.H * p.W * p.x_ld + s * 16 + sub * KPL
                : reinterpret_cast<const T*>(p.sc) + (size_t)ln * p.D * p.H * p.W * p.sc_ld + (s - nchunks) * 16 + sub * KPL;
    nxt_main = main;
    nxt_norm = main && p.in_norm != nullptr;
    nrec_next = nxt_norm ? p.in_norm + (size_t)ln * p.Cin + s * 16 + sub * KPL : nullptr;
  };
#define BPX_LOAD_PIECE(u)                                                                                     \
  do {                                                                                                        \
    pbuf[u] = u32x4_t{0u, 0u, 0u, 0u};                                                                        \
    if (goff[u] != 0xFFFFFFFFu) pbuf[u] = *reinterpret_cast<const u32x4_t*>(lsrc + goff[u]);                  \
  } while (0)
  // transform the piece held in pbuf[u] (stage in registers) and store it into the LDS buffer at byte offset wbuf
#define BPX_WRITE_PIECE(u, wbuf)                                                                              \
  do {                                                                                                        \
    const int idx_ = (u) * 256 + tid;                                                                         \
    const bool live_ = cur_main ? (idx_ < HV * GPT) : ((u) < NPS && idx_ < TV * GPT);                         \
    if (live_) {                                                                                              \
      u32x4_t v_ = pbuf[u];                                                                                   \
      if (cur_norm && ((cur_valid >> (u)) & 1u)) {                                                            \
        float f_[KPL];                                                                                        \
        unpack16<T>(v_, f_);                                                                                  \
        _Pragma("unroll") for (int e_ = 0; e_ < KPL; ++e_) f_[e_] = apply_act_rt<T, ACTK>(fmaf(psc[e_], f_[e_], psh[e_]), p.act); \
        v_ = pack16<T>(f_);                                                                                   \
      }                                                                                                       \
      *reinterpret_cast<u32x4_t*>(smem + (wbuf) + (size_t)idx_ * 16) = v_;                                    \
    }                                                                                                         \
  } while (0)
  auto promote = [&]() {  // the stage that was being loaded becomes the stage held in registers
    cur_norm = nxt_norm; cur_main = nxt_main; cur_valid = nxt_valid;
    if (cur_norm) {  // issued after the previous stage's last use of psc/psh; first needed one MFMA step + a barrier later
#pragma unroll
      for (int e = 0; e < KPL; ++e) { bpx_norm_rec r = nrec_next[e]; psc[e] = r.scale; psh[e] = r.shift; }
    }
  };

  // ---------------------------------------------------------------- MFMA-side constants
  f32x4_t acc[MS][NS];
  int hb[MS], tb[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int t = (wave * MS + ms) * 16 + j;
    hb[ms] = (((t / (TY * TX)) * HY + (t / TX) % TY) * HX + t % TX) * VB;
    tb[ms] = t * VB;
  }
  const int cg_off = (GPT == 2 ? (g & 1) : g) * 16;
  const bool hi_tap = (GPT == 2) && (g >> 1);
  // ds_read address = hb[ms] + (lane delta of the step's tap-pair class + buffer offset) + immediate: one v_add per read
  // instead of four pre-added copies of hb (register pressure - the persistent kernel keeps a second stage in registers)
  constexpr int NCLS = (GPT == 2) ? 4 : 1;
  int dcls[NCLS];
  dcls[0] = cg_off + ((GPT == 2 && hi_tap) ? VB : 0);
  if (GPT == 2) {
    dcls[1] = cg_off + (hi_tap ? HX * VB : 0);
    dcls[2] = cg_off + (hi_tap ? HY * HX * VB : 0);
    dcls[3] = cg_off;
  }
  const T* __restrict__ wp = reinterpret_cast<const T*>(p.wp);
  const T* __restrict__ wsc = reinterpret_cast<const T*>(p.wsc);
  // per-workgroup constants of the epilogue
  float add[NS][4], w1[NS][4];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co_base + ns * 16 + g * 4 + r;
      add[ns][r] = 0.f; w1[ns][r] = 0.f;
      if (EPI == EPI_FWD) {
        if (p.bias) add[ns][r] += p.bias[co];
        if (p.sc && p.bias_sc) add[ns][r] += p.bias_sc[co];
        if (p.sc && p.sc_C == 1) w1[ns][r] = reinterpret_cast<const float*>(p.wsc)[co];
      }
    }
  float s1[NS][4], s2[NS][4];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[ns][r] = s2[ns][r] = 0.f;
  int n_acc = -1;  // sample the statistics registers belong to
  auto flush_stats = [&](int n_from, int n_to) {
    // write the partial of sample n_from (if any) and zeros for samples (n_from, n_to): every (sample, workgroup, wave)
    // slot of the partial array is written exactly once, so it needs no memset and the sum is deterministic
    if (p.part == nullptr) return;
    for (int n = (n_from < 0 ? 0 : n_from); n < n_to; ++n) {
      const bool real = (n == n_from);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = real ? s1[ns][r] : 0.f, b = real ? s2[ns][r] : 0.f;
#pragma unroll
          for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
          if (j == 0) {
            float* pp = p.part + ((((size_t)n * gridDim.x + blockIdx.x) * 4 + wave) * 2) * Cout + co_base + ns * 16 + g * 4 + r;
            pp[0] = a;
            pp[Cout] = b;
          }
        }
    }
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 4; ++r) s1[ns][r] = s2[ns][r] = 0.f;
  };

  int stamp_i = 0;
#define BPX_STAMP() do { if (p.stamps && tid == 0 && blockIdx.y == 0 && stamp_i < 16) p.stamps[(size_t)blockIdx.x * 16 + stamp_i++] = (long long)__builtin_readcyclecounter(); } while (0)
  BPX_STAMP();  // 0: constants done
  // ---------------------------------------------------------------- pipeline prologue
  begin_load(0);
#pragma unroll
  for (int u = 0; u < NPM; ++u) BPX_LOAD_PIECE(u);
  promote();
#pragma unroll
  for (int u = 0; u < NPM; ++u) BPX_WRITE_PIECE(u, 0);
  if (nstage > 1) {
    begin_load(1);
#pragma unroll
    for (int u = 0; u < NPM; ++u) BPX_LOAD_PIECE(u);
    promote();
  }
  __syncthreads();

  BPX_STAMP();  // 1: prologue done
  int n = 0, tile = 0, z0 = 0, y0 = 0, x0 = 0;
  for (int q = 0; q < nstage; ++q) {
    const int s = q % S;
    const int bufo = (q & 1) * BUFB, nbuf = BUFB - bufo;
    const bool stage_next = q + 1 < nstage;     // pbuf holds stage q+1
    const bool load_next2 = q + 2 < nstage;
    if (s == 0) {
      const int tt = (int)blockIdx.x + (q / S) * (int)gridDim.x;
      tile = tt % p.tilesPerSample;
      n = tt / p.tilesPerSample;
      x0 = (tile % p.tilesX) * TX; y0 = ((tile / p.tilesX) % p.tilesY) * TY; z0 = (tile / (p.tilesX * p.tilesY)) * TZ;
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (n != n_acc) {
        flush_stats(n_acc, n);
        n_acc = n;
        if (EPI == EPI_DGRAD && p.t_norm) {  // visible to all waves after the stage's barrier, long before the tile's epilogue
          if (tid < NS * 16) srec[tid] = p.t_norm[(size_t)n * Cout + co_base + tid];
        }
      }
    }
    if (load_next2) begin_load(q + 2);   // index math of the stage after next: VALU filler, off the critical path
    BPX_STAMP();  // a: stage bookkeeping + index math

    // epilogue operands of this tile, fetched before its last step loop
    float img[MS];
    u32x2_t traw[MS][NS];
    if (s == S - 1) {
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        const int t = (wave * MS + ms) * 16 + j;
        const int z = z0 + t / (TY * TX), y = y0 + (t / TX) % TY, x = x0 + t % TX;
        const bool ok = z < p.D && y < p.H && x < p.W;
        const size_t vox = (((size_t)n * p.D + z) * p.H + y) * p.W + x;
        img[ms] = 0.f;
        if (EPI == EPI_FWD && p.sc && p.sc_C == 1 && ok) img[ms] = reinterpret_cast<const float*>(p.sc)[vox];
        if (EPI == EPI_DGRAD && p.t_norm) {
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) {
            traw[ms][ns] = u32x2_t{0u, 0u};
            if (ok) traw[ms][ns] = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const T*>(p.t) + vox * (size_t)p.t_ld + co_base + ns * 16 + g * 4);
          }
        }
      }
    }

    if (s < nchunks) {
      const T* wl = wp + ((size_t)s * QPAD * Cout + (size_t)g * Cout + co_base + j) * KPL;
      u32x4_t wq[3][NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        wq[0][ns] = *reinterpret_cast<const u32x4_t*>(wl + (size_t)ns * 16 * KPL);
        wq[1][ns] = *reinterpret_cast<const u32x4_t*>(wl + ((size_t)4 * Cout + ns * 16) * KPL);
      }
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        if (st + 2 < STEPS) {
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
            wq[(st + 2) % 3][ns] = *reinterpret_cast<const u32x4_t*>(wl + ((size_t)(st + 2) * 4 * Cout + ns * 16) * KPL);
        }
        const int tapA = (GPT == 2) ? bpx_tap_order_bf16(2 * st) : st;
        const int cls = (GPT == 2) ? (st < 9 ? 0 : st < 12 ? 1 : st == 12 ? 2 : 3) : 0;
        const int imm = tap_off<HY, HX, VB>(tapA);
        u32x4_t af[MS];
        const int dsel = dcls[cls] + bufo;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) af[ms] = *reinterpret_cast<const u32x4_t*>(smem + hb[ms] + dsel + imm);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = mfma_step<T>(wq[st % 3][ns], af[ms], acc[ms][ns]);
        if (st < NPM && stage_next) {
          BPX_WRITE_PIECE(st < NPM ? st : 0, nbuf);
          if (load_next2) BPX_LOAD_PIECE(st < NPM ? st : 0);
        }
      }
    } else {
      const T* wl = wsc + ((size_t)(s - nchunks) * 4 * Cout + (size_t)g * Cout + co_base + j) * KPL;
      u32x4_t wf[NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) wf[ns] = *reinterpret_cast<const u32x4_t*>(wl + (size_t)ns * 16 * KPL);
      u32x4_t af[MS];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) af[ms] = *reinterpret_cast<const u32x4_t*>(smem + bufo + tb[ms] + cg_off);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = mfma_step<T>(wf[ns], af[ms], acc[ms][ns]);
      if (stage_next) {
#pragma unroll
        for (int u = 0; u < NPM; ++u) {
          BPX_WRITE_PIECE(u, nbuf);
          if (load_next2) BPX_LOAD_PIECE(u);
        }
      }
    }
    if (load_next2) promote();
    BPX_STAMP();  // b: step loop (+ staging) done

    if (s == S - 1) {
      // ------------------------------------------------ epilogue of the tile (no loads, no reductions)
      T* __restrict__ yout = reinterpret_cast<T*>(p.y);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        const int t = (wave * MS + ms) * 16 + j;
        const int z = z0 + t / (TY * TX), y = y0 + (t / TX) % TY, x = x0 + t % TX;
        if (z < p.D && y < p.H && x < p.W) {
          const size_t vox = (((size_t)n * p.D + z) * p.H + y) * p.W + x;
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) {
            float v[4];
            if (EPI == EPI_FWD) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v[r] = acc[ms][ns][r] + add[ns][r] + img[ms] * w1[ns][r];
                s1[ns][r] += v[r];
                s2[ns][r] += v[r] * v[r];
              }
            } else if (p.t_norm) {
              const float tv4[4] = {bf16lo(traw[ms][ns][0]), bf16hi(traw[ms][ns][0]), bf16lo(traw[ms][ns][1]), bf16hi(traw[ms][ns][1])};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const bpx_norm_rec rc = srec[ns * 16 + g * 4 + r];
                const float u = fmaf(rc.scale, tv4[r], rc.shift);
                v[r] = acc[ms][ns][r] * apply_act_bwd_rt<T, ACTK>(u, p.t_act);
                s1[ns][r] += v[r];
                s2[ns][r] += v[r] * ((tv4[r] - rc.mean) * rc.rstd);
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = acc[ms][ns][r];
            }
            *reinterpret_cast<u32x2_t*>(yout + vox * (size_t)p.y_ld + co_base + ns * 16 + g * 4) =
                u32x2_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          }
        }
      }
    }
    BPX_STAMP();  // c: epilogue (if last stage of the tile)
    __syncthreads();
    BPX_STAMP();  // d: barrier
  }
  flush_stats(n_acc, p.N);
#undef BPX_LOAD_PIECE
#undef BPX_WRITE_PIECE
}

template <int EPI>
int launch_p(const Conv3Params& p0, const TileCfg& c, hipStream_t s) {
  using T = uint16_t;
  Conv3Params p = p0;
  p.tilesY = cdiv(p.H, c.ty);
  p.tilesX = cdiv(p.W, c.tx);
  p.tilesPerSample = cdiv(p.D, c.tz) * p.tilesY * p.tilesX;
  p.totalTiles = p.N * p.tilesPerSample;
  p.stamps = g_conv_stamps;
  const int gy = p.Cout / (16 * c.ns);
  dim3 grid((unsigned)conv3_persist_groups(p.totalTiles, gy), (unsigned)gy);
  const bool elu = (EPI == EPI_FWD ? p.act : p.t_act) == BPX_ACT_ELU;
#define L(TZ, TY, TX, NS)                                                          \
  if (c.tz == TZ && c.ty == TY && c.tx == TX && c.ns == NS) {                      \
    if (elu) conv3_p_kernel<T, TZ, TY, TX, NS, EPI, 1><<<grid, 256, 0, s>>>(p);    \
    else conv3_p_kernel<T, TZ, TY, TX, NS, EPI, 0><<<grid, 256, 0, s>>>(p);        \
    return 0;                                                                      \
  }
  L(4, 8, 16, 1) L(4, 4, 16, 1) L(4, 4, 16, 2) L(4, 4, 16, 3) L(4, 4, 16, 4) L(4, 4, 8, 1) L(4, 4, 8, 2) L(4, 4, 8, 3) L(4, 4, 8, 4)
#undef L
  return 1;
}

}  // namespace

namespace bpxconv {
int launch_conv3_persist(int epi, const Conv3Params& p, const TileCfg& c, hipStream_t s) {
  return epi == EPI_FWD ? launch_p<EPI_FWD>(p, c, s) : launch_p<EPI_DGRAD>(p, c, s);
}
}  // namespace bpxconv
