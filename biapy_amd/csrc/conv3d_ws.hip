// Conv3d 3x3x3 implicit GEMM, wave-specialised and persistent (bf16 storage; the production kernel).
//
// Why this shape.  rocprofv3 on the plain 4-wave kernel (profiles/r01_pmc_conv3_fwd_48to16_128.txt) showed the matrix pipe
// 18 % and the VALU 38 % busy: every wave alternated between a VALU-only phase (normalise + ELU + bf16 convert of the
// halo it stages) and an MFMA-only phase, with barriers and LDS/HBM latency in between, and only two waves per SIMD to
// hide any of it.  Here the two kinds of work live in DIFFERENT waves of one 512-thread workgroup:
//
//   waves 0-3  "MFMA"  : ds_read_b128 operand fetch -> v_mfma_f32_16x16x32_bf16 -> epilogue (bias / shortcut / ELU' /
//                        statistics / stores).  Their VMEM queue only ever holds weight fragments (L1/L2 hits), so the
//                        in-order vmcnt retirement never makes a weight wait behind an HBM prefetch.
//   waves 4-7  "STAGE" : global_load (one stage ahead, kept in registers) -> fp32 normalise + activate ->
//                        v_cvt_pk_bf16_f32 -> ds_write_b128 into the OTHER LDS halo buffer.
//
// One MFMA wave and one STAGE wave share each SIMD, so the matrix pipe and the VALU run concurrently by construction.
// The workgroup is persistent: it walks tiles blockIdx.x, +gridDim.x, ... and the stage pipeline (tile, channel chunk)
// never drains at tile boundaries - the STAGE waves are already fetching the next tile's halo while the MFMA waves run
// the epilogue.  One s_barrier per stage.  The fused 1x1x1 shortcut chunks are ordinary stages with a halo-free geometry.
#include "conv3d_shared.h"

using namespace bpxconv;

namespace {

template <typename T, int TZ, int TY, int TX, int NS, int EPI, int ACTK>
__global__ void __launch_bounds__(512, ((NS <= 2 && TZ * TY * TX <= 256) ? 4 : 2)) conv3_ws_kernel(const Conv3Params p) {
  using Tr = ElemTraits<T>;
  constexpr int KPL = Tr::KPL, GPT = 16 / KPL, VB = 16 * (int)sizeof(T);
  constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX, TV = TZ * TY * TX;
  constexpr int QTOT = 27 * GPT, STEPS = (QTOT + 3) / 4, QPAD = STEPS * 4;
  constexpr int MT = TV / 16, MS = MT / 4;
  static_assert(MT % 4 == 0 && MS >= 1, "tile must give every MFMA wave at least one m-subtile");
  constexpr int BUFB = HV * VB;
  constexpr int NPM = (HV * GPT + 255) / 256;  // 16-byte pieces per STAGE thread, halo geometry
  constexpr int NPS = (TV * GPT + 255) / 256;  // ... shortcut (halo-free) geometry
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunks = p.Cin / 16;
  const int nsc = (EPI == EPI_FWD && p.sc != nullptr && p.sc_C >= 16) ? p.sc_C / 16 : 0;
  const int S = nchunks + nsc;                                            // stages per tile
  const int ntl = (p.totalTiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nstage = ntl * S;
  const int co_base = blockIdx.y * 16 * NS;
  const int Cout = p.Cout;

  if (wave >= 4) {
    // =============================================== STAGE waves ===============================================
    const int st = tid - 256, sub = st % GPT;
    u32x4_t pbuf[NPM];
    uint32_t goff[NPM];
    float psc[KPL], psh[KPL];
    bool pnorm = false;         // the stage held in pbuf gets the normalise+activate prologue
    bool pmain = true;          // ... and has the halo geometry
    uint32_t pvalid = 0;        // bit u: piece u of that stage is inside the volume
    int n = 0, z0 = 0, y0 = 0, x0 = 0;

    // issue the global loads of stage q into pbuf (+ its normalisation records)
    auto load_stage = [&](int q) {
      const int s = q % S;
      if (s == 0) {
        const int tt = (int)blockIdx.x + (q / S) * (int)gridDim.x;
        const int tile = tt % p.tilesPerSample;
        n = tt / p.tilesPerSample;
        x0 = (tile % p.tilesX) * TX; y0 = ((tile / p.tilesX) % p.tilesY) * TY; z0 = (tile / (p.tilesX * p.tilesY)) * TZ;
      }
      const bool main = s < nchunks;
      if (s == 0 || s == nchunks) {  // new geometry: recompute the piece -> voxel map
        pvalid = 0;
#pragma unroll
        for (int u = 0; u < NPM; ++u) {
          const int idx = u * 256 + st;
          goff[u] = 0xFFFFFFFFu;
          int gz, gy, gx;
          bool in;
          if (main) {
            const int hv = idx / GPT;
            gz = z0 - 1 + hv / (HX * HY); gy = y0 - 1 + (hv / HX) % HY; gx = x0 - 1 + hv % HX;
            in = idx < HV * GPT;
          } else {
            const int t = idx / GPT;
            gz = z0 + t / (TY * TX); gy = y0 + (t / TX) % TY; gx = x0 + t % TX;
            in = (u < NPS) && idx < TV * GPT;
          }
          if (in && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
            goff[u] = (uint32_t)((((uint32_t)gz * p.H + gy) * p.W + gx) * (uint32_t)(main ? p.x_ld : p.sc_ld));
            pvalid |= 1u << u;
          }
        }
      }
      const T* src = main ? reinterpret_cast<const T*>(p.x) + (size_t)n * p.D * p.H * p.W * p.x_ld + s * 16 + sub * KPL
                          : reinterpret_cast<const T*>(p.sc) + (size_t)n * p.D * p.H * p.W * p.sc_ld + (s - nchunks) * 16 + sub * KPL;
#pragma unroll
      for (int u = 0; u < NPM; ++u) {
        pbuf[u] = u32x4_t{0u, 0u, 0u, 0u};
        if (goff[u] != 0xFFFFFFFFu) pbuf[u] = *reinterpret_cast<const u32x4_t*>(src + goff[u]);
      }
      pmain = main;
      pnorm = main && p.in_norm != nullptr;
      if (pnorm) {
        const bpx_norm_rec* nr = p.in_norm + (size_t)n * p.Cin + s * 16 + sub * KPL;
#pragma unroll
        for (int e = 0; e < KPL; ++e) { bpx_norm_rec r = nr[e]; psc[e] = r.scale; psh[e] = r.shift; }
      }
    };
    // normalise + activate (fp32) what pbuf holds and store it to the LDS buffer at byte offset wbuf (piece-linear image)
    auto transform_stage = [&](int wbuf) {
#pragma unroll
      for (int u = 0; u < NPM; ++u) {
        const int idx = u * 256 + st;
        const bool live = pmain ? (idx < HV * GPT) : (u < NPS && idx < TV * GPT);
        if (live) {
          u32x4_t v = pbuf[u];
          if (pnorm && ((pvalid >> u) & 1u)) {
            float f[KPL];
            unpack16<T>(v, f);
#pragma unroll
            for (int e = 0; e < KPL; ++e) f[e] = apply_act_rt<T, ACTK>(fmaf(psc[e], f[e], psh[e]), p.act);
            v = pack16<T>(f);
          }
          *reinterpret_cast<u32x4_t*>(smem + wbuf + (size_t)idx * 16) = v;
        }
      }
    };

    load_stage(0);
    transform_stage(0);
    if (nstage > 1) load_stage(1);
    __syncthreads();
    for (int q = 0; q < nstage; ++q) {
      if (q + 1 < nstage && !(p.dbg & 2)) {
        transform_stage(((q + 1) & 1) * BUFB);
        if (q + 2 < nstage) load_stage(q + 2);
      }
      __syncthreads();
    }
    return;
  }

  // ================================================= MFMA waves =================================================
  const int j = lane & 15, g = lane >> 4;
  f32x4_t acc[MS][NS];
  int hb[MS], tb[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int t = (wave * MS + ms) * 16 + j;
    const int tz = t / (TY * TX), ty = (t / TX) % TY, tx = t % TX;
    hb[ms] = ((tz * HY + ty) * HX + tx) * VB;
    tb[ms] = t * VB;
  }
  const int cg_off = (GPT == 2 ? (g & 1) : g) * 16;
  const bool hi_tap = (GPT == 2) && (g >> 1);
  constexpr int NCLS = (GPT == 2) ? 4 : 1;
  int lbase[NCLS][MS];   // ds_read bases for LDS buffer 0 (see conv3d_igemm.hip for the paired-tap K order)
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    if (GPT == 2) {
      lbase[0][ms] = hb[ms] + cg_off + (hi_tap ? VB : 0);
      lbase[1][ms] = hb[ms] + cg_off + (hi_tap ? HX * VB : 0);
      lbase[2][ms] = hb[ms] + cg_off + (hi_tap ? HY * HX * VB : 0);
      lbase[NCLS - 1][ms] = hb[ms] + cg_off;
    } else {
      lbase[0][ms] = hb[ms] + cg_off;
    }
  }
  const T* __restrict__ wp = reinterpret_cast<const T*>(p.wp);
  const T* __restrict__ wsc = reinterpret_cast<const T*>(p.wsc);
  int n = 0, tile = 0, z0 = 0, y0 = 0, x0 = 0;

  __syncthreads();
  for (int q = 0; q < nstage; ++q) {
    const int s = q % S;
    const int bufo = (q & 1) * BUFB;
    if (s == 0) {
      const int tt = (int)blockIdx.x + (q / S) * (int)gridDim.x;
      tile = tt % p.tilesPerSample;
      n = tt / p.tilesPerSample;
      x0 = (tile % p.tilesX) * TX; y0 = ((tile / p.tilesX) % p.tilesY) * TY; z0 = (tile / (p.tilesX * p.tilesY)) * TZ;
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    if (p.dbg & 1) {
    } else if (s < nchunks) {
      // Software pipeline inside the wave: the LDS reads of step st+1 and the weight fragment of step st+2 are issued
      // BEFORE the MFMAs of step st, so the matrix pipe never waits for an operand fetch it could have overlapped.
      const T* wl = wp + ((size_t)s * QPAD * Cout + (size_t)g * Cout + co_base + j) * KPL;
      u32x4_t wq[3][NS];
      u32x4_t af[2][MS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        wq[0][ns] = *reinterpret_cast<const u32x4_t*>(wl + (size_t)ns * 16 * KPL);
        wq[1][ns] = *reinterpret_cast<const u32x4_t*>(wl + ((size_t)4 * Cout + ns * 16) * KPL);
      }
      {
        const int tap0 = (GPT == 2) ? bpx_tap_order_bf16(0) : 0;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) af[0][ms] = *reinterpret_cast<const u32x4_t*>(smem + bufo + lbase[0][ms] + tap_off<HY, HX, VB>(tap0));
      }
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        if (st + 1 < STEPS) {
          const int tapN = (GPT == 2) ? bpx_tap_order_bf16(2 * (st + 1)) : st + 1;
          const int clsN = (GPT == 2) ? (st + 1 < 9 ? 0 : st + 1 < 12 ? 1 : st + 1 == 12 ? 2 : 3) : 0;
#pragma unroll
          for (int ms = 0; ms < MS; ++ms)
            af[(st + 1) & 1][ms] = *reinterpret_cast<const u32x4_t*>(smem + bufo + lbase[clsN][ms] + tap_off<HY, HX, VB>(tapN));
        }
        if (st + 2 < STEPS) {
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
            wq[(st + 2) % 3][ns] = *reinterpret_cast<const u32x4_t*>(wl + ((size_t)(st + 2) * 4 * Cout + ns * 16) * KPL);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = mfma_step<T>(wq[st % 3][ns], af[st & 1][ms], acc[ms][ns]);
        __builtin_amdgcn_sched_barrier(0);
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
    }

    if (s == S - 1) {
      // ---------------------------------------------- epilogue of the tile ----------------------------------------------
      T* __restrict__ yout = reinterpret_cast<T*>(p.y);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const int co = co_base + ns * 16 + g * 4;
        float add[4] = {0.f, 0.f, 0.f, 0.f}, w1[4] = {0.f, 0.f, 0.f, 0.f};
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        bpx_norm_rec rec[4];
        if (EPI == EPI_FWD) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (p.bias) add[r] += p.bias[co + r];
            if (p.sc && p.bias_sc) add[r] += p.bias_sc[co + r];
            if (p.sc && p.sc_C == 1) w1[r] = reinterpret_cast<const float*>(p.wsc)[co + r];
          }
        } else if (p.t_norm) {
#pragma unroll
          for (int r = 0; r < 4; ++r) rec[r] = p.t_norm[(size_t)n * Cout + co + r];
        }
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
          const int t = (wave * MS + ms) * 16 + j;
          const int z = z0 + t / (TY * TX), y = y0 + (t / TX) % TY, x = x0 + t % TX;
          if (z < p.D && y < p.H && x < p.W) {
            const size_t vox = (((size_t)n * p.D + z) * p.H + y) * p.W + x;
            float v[4];
            if (EPI == EPI_FWD) {
              const float img = (p.sc && p.sc_C == 1) ? reinterpret_cast<const float*>(p.sc)[vox] : 0.f;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v[r] = acc[ms][ns][r] + add[r] + img * w1[r];
                s1[r] += v[r];
                s2[r] += v[r] * v[r];
              }
            } else if (p.t_norm) {
              const u32x2_t traw = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const T*>(p.t) + vox * (size_t)p.t_ld + co);
              const float tv4[4] = {bf16lo(traw[0]), bf16hi(traw[0]), bf16lo(traw[1]), bf16hi(traw[1])};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float u = fmaf(rec[r].scale, tv4[r], rec[r].shift);
                v[r] = acc[ms][ns][r] * apply_act_bwd_rt<T, ACTK>(u, p.t_act);
                s1[r] += v[r];
                s2[r] += v[r] * ((tv4[r] - rec[r].mean) * rec[r].rstd);
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = acc[ms][ns][r];
            }
            *reinterpret_cast<u32x2_t*>(yout + vox * (size_t)p.y_ld + co) = u32x2_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          }
        }
        if (p.part != nullptr) {  // one partial per (tile, MFMA wave): no workgroup barrier needed
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float a = s1[r], b = s2[r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
            if (j == 0) {
              float* pp = p.part + ((((size_t)n * p.tilesPerSample + tile) * 4 + wave) * 2) * Cout + co + r;
              pp[0] = a;
              pp[Cout] = b;
            }
          }
        }
      }
    }
    __syncthreads();
  }
}

template <int EPI>
int launch_ws(const Conv3Params& p0, const TileCfg& c, hipStream_t s) {
  using T = uint16_t;
  Conv3Params p = p0;
  const int tilesZ = cdiv(p.D, c.tz);
  p.tilesY = cdiv(p.H, c.ty);
  p.tilesX = cdiv(p.W, c.tx);
  p.tilesPerSample = tilesZ * p.tilesY * p.tilesX;
  p.totalTiles = p.N * p.tilesPerSample;
  { const char* e = getenv("BPX_CONV_DBG"); p.dbg = e ? atoi(e) : 0; }
  const int gy = p.Cout / (16 * c.ns);
  const int gx = std::max(1, std::min(p.totalTiles, (2 * 256 + gy - 1) / gy));  // ~2 persistent workgroups per CU
  dim3 grid((unsigned)gx, (unsigned)gy);
  const bool elu = (EPI == EPI_FWD ? p.act : p.t_act) == BPX_ACT_ELU;
#define L(TZ, TY, TX, NS)                                                          \
  if (c.tz == TZ && c.ty == TY && c.tx == TX && c.ns == NS) {                      \
    if (elu) conv3_ws_kernel<T, TZ, TY, TX, NS, EPI, 1><<<grid, 512, 0, s>>>(p);   \
    else conv3_ws_kernel<T, TZ, TY, TX, NS, EPI, 0><<<grid, 512, 0, s>>>(p);       \
    return 0;                                                                      \
  }
  L(4, 8, 16, 1) L(4, 4, 16, 1) L(4, 4, 16, 2) L(4, 4, 16, 3) L(4, 4, 16, 4) L(4, 4, 8, 1) L(4, 4, 8, 2) L(4, 4, 8, 3) L(4, 4, 8, 4)
#undef L
  return 1;
}

}  // namespace

namespace bpxconv {
int launch_conv3_ws(int epi, const Conv3Params& p, const TileCfg& c, hipStream_t s) {
  return epi == EPI_FWD ? launch_ws<EPI_FWD>(p, c, s) : launch_ws<EPI_DGRAD>(p, c, s);
}
}  // namespace bpxconv
