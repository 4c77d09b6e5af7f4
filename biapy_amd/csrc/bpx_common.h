// Shared device/host helpers for libbiapy_amd (gfx950 only - no portability layer by design).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/biapy_amd.h"

// ---- error plumbing -------------------------------------------------------------------------
void bpx_set_error(const char* fmt, ...);
#define BPX_FAIL(...)            \
  do {                           \
    bpx_set_error(__VA_ARGS__);  \
    return 1;                    \
  } while (0)
#define BPX_CHECK(cond, ...) \
  do {                       \
    if (!(cond)) BPX_FAIL(__VA_ARGS__); \
  } while (0)
#define BPX_LAUNCH_CHECK(name)                                                         \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) BPX_FAIL("%s: launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---- vector types ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;  // 8 bf16 = one 16x16x32 MFMA operand
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) ------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi);
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// storage-type traits: T = float (exact mode) or uint16_t (bf16 bits)
template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int KPL = 4;   // elements per 16-byte lane operand
  static constexpr int DT = BPX_F32;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemTraits<uint16_t> {
  static constexpr int KPL = 8;
  static constexpr int DT = BPX_BF16;
  __device__ static __forceinline__ float ld(const uint16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(uint16_t* p, float v) { *p = f32_to_bf16(v); }
};
// fp16 storage: the same 16 bits per element with an 11-bit mantissa instead of 8 - the forward mode whose Dice matches the fp32 reference to
// < 1e-4.  Inference, and the forward half of the mixed training mode (BPX_MIX16: fp16 activations, bf16 gradient tensors - gradients in fp16
// would need loss scaling, so the backward kernels take fp16 only as their ACTIVATION operand).  fp16 ends at 65504 where bf16 has the fp32
// range: producers of RAW (pre-normalisation) tensors store through pk16s / the saturating pack below.  A distinct element type: _Float16.
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
template <> struct ElemTraits<f16_t> {
  static constexpr int KPL = 8;
  static constexpr int DT = BPX_F16;
  __device__ static __forceinline__ float ld(const f16_t* p) { return (float)*p; }
  __device__ static __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }
};

// 16 bytes of T -> up to 8 floats and back
template <typename T> __device__ __forceinline__ void unpack16(const u32x4_t& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4_t& v, float* f) {
  f[0] = __uint_as_float(v[0]); f[1] = __uint_as_float(v[1]); f[2] = __uint_as_float(v[2]); f[3] = __uint_as_float(v[3]);
}
template <> __device__ __forceinline__ void unpack16<uint16_t>(const u32x4_t& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = bf16lo(v[i]); f[2 * i + 1] = bf16hi(v[i]); }
}
template <> __device__ __forceinline__ void unpack16<f16_t>(const u32x4_t& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t w = v[i];   // a copy first: __builtin_bit_cast applied to the vector ELEMENT v[i] reads element 0 for every i (hipcc 7.2)
    const f16x2_t h = __builtin_bit_cast(f16x2_t, w);
    f[2 * i] = (float)h[0]; f[2 * i + 1] = (float)h[1];          // v_cvt_f32_f16 (+ SDWA for the high half)
  }
}
template <typename T> __device__ __forceinline__ u32x4_t pack16(const float* f);
template <> __device__ __forceinline__ u32x4_t pack16<float>(const float* f) {
  u32x4_t v; v[0] = __float_as_uint(f[0]); v[1] = __float_as_uint(f[1]); v[2] = __float_as_uint(f[2]); v[3] = __float_as_uint(f[3]);
  return v;
}
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// hardware RNE conversion: v_cvt_pk_bf16_f32 (one instruction per two values)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, hw_bf16x2_t));
}
template <> __device__ __forceinline__ u32x4_t pack16<uint16_t>(const float* f) {
  u32x4_t v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = cvt_pk_bf16(f[2 * i], f[2 * i + 1]);
  return v;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) { return cvt_pk_bf16(lo, hi); }
// v_cvt_pk_f16_f32 (RNE)
__device__ __forceinline__ uint32_t cvt_pk_f16(float lo, float hi) { return __builtin_bit_cast(uint32_t, f16x2_t{(f16_t)lo, (f16_t)hi}); }
template <> __device__ __forceinline__ u32x4_t pack16<f16_t>(const float* f) {
  u32x4_t v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = cvt_pk_f16(f[2 * i], f[2 * i + 1]);
  return v;
}
// the two halves of a packed pair and the packing itself, by 16-bit storage type (uint16_t = bf16 bits, f16_t = fp16)
template <typename T> __device__ __forceinline__ float lo16(uint32_t w);
template <typename T> __device__ __forceinline__ float hi16(uint32_t w);
template <typename T> __device__ __forceinline__ uint32_t pk16(float lo, float hi);
template <> __device__ __forceinline__ float lo16<uint16_t>(uint32_t w) { return bf16lo(w); }
template <> __device__ __forceinline__ float hi16<uint16_t>(uint32_t w) { return bf16hi(w); }
template <> __device__ __forceinline__ uint32_t pk16<uint16_t>(float lo, float hi) { return cvt_pk_bf16(lo, hi); }
template <> __device__ __forceinline__ float lo16<f16_t>(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
template <> __device__ __forceinline__ float hi16<f16_t>(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
template <> __device__ __forceinline__ uint32_t pk16<f16_t>(float lo, float hi) { return cvt_pk_f16(lo, hi); }
// SATURATING pack for the stores of raw conv / transposed-conv / pooling outputs: an fp32 result beyond +-65504 becomes +-65504
// instead of +-inf (an inf in a stored activation turns the next InstanceNorm's output into NaN for the whole channel and train_engine stops on
// the non-finite loss; bf16 storage cannot fail that way).  v_cvt_pk_f16_f32 + v_pk_min_f16 + v_pk_max_f16: two more VALU instructions per
// pair, only in epilogues.  The statistics partials are taken from the UNCLAMPED fp32 values, so a genuinely diverged (NaN) result still
// poisons the norm record and surfaces as a non-finite loss.  bf16: identical to pk16.
template <typename T> __device__ __forceinline__ uint32_t pk16s(float lo, float hi) { return pk16<T>(lo, hi); }
#ifndef BPX_F16_SATURATE
#define BPX_F16_SATURATE 1
#endif
template <> __device__ __forceinline__ uint32_t pk16s<f16_t>(float lo, float hi) {
  if (!BPX_F16_SATURATE) return cvt_pk_f16(lo, hi);
  const f16x2_t mx{(f16_t)65504.f, (f16_t)65504.f};
  f16x2_t h{(f16_t)lo, (f16_t)hi};
  h = __builtin_elementwise_max(__builtin_elementwise_min(h, mx), -mx);
  return __builtin_bit_cast(uint32_t, h);
}

// K order of the 27 taps for bf16 storage: one MFMA step = two taps x 16 channels.  Taps are paired so that
// the LDS address difference between the two taps of a step is one of three constants (+1 voxel in x for the
// (dx0,dx1) pairs, +1 row for the dx2 pairs, +1 plane for (8,17)); tap 26 is alone (-1 = zero weights).
__host__ __device__ constexpr int bpx_tap_order_bf16(int i) {
  constexpr int t[28] = {0, 1, 3, 4, 6, 7, 9, 10, 12, 13, 15, 16, 18, 19, 21, 22, 24, 25, 2, 5, 11, 14, 20, 23, 8, 17, 26, -1};
  return t[i];
}

// ---- activations ----------------------------------------------------------------------------
// act(u) and act'(u).  ELU(alpha=1): u>0 ? u : exp(u)-1 (PyTorch uses expm1; |diff| <= 1 ulp of 1.0
// which is far below bf16 and within the stated fp32 tolerance).
template <int ACT> __device__ __forceinline__ float act_fwd(float u) {
  if (ACT == BPX_ACT_ELU) return u > 0.f ? u : (__expf(u) - 1.f);
  if (ACT == BPX_ACT_RELU) return u > 0.f ? u : 0.f;
  if (ACT == BPX_ACT_SILU) return u / (1.f + __expf(-u));
  return u;
}
template <int ACT> __device__ __forceinline__ float act_bwd(float u) {
  if (ACT == BPX_ACT_ELU) return u > 0.f ? 1.f : __expf(u);
  if (ACT == BPX_ACT_RELU) return u > 0.f ? 1.f : 0.f;
  if (ACT == BPX_ACT_SILU) { float s = 1.f / (1.f + __expf(-u)); return s * (1.f + u * (1.f - s)); }
  return 1.f;
}
// erf(x) by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7: below fp32 rounding of the sums it enters; one exp, five fmas) - GELU
__device__ __forceinline__ float bpx_erf(float x) {
  const float ax = fabsf(x), t = 1.f / fmaf(0.3275911f, ax, 1.f);
  const float p = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float r = 1.f - p * __expf(-ax * ax);
  return copysignf(r, x);
}
// ONE run-time switch over every block activation of the reference (blocks.py:1973-1998), used by every kernel's run-time-activation
// instance (conv prologues, dgrad epilogues, wgrad staging, the materialised norm + act pair, gate MLPs).  PRECISE: fp32 storage mode (libm
// exp / expm1 instead of the fast forms).
// The round-4 codes (leaky_relu ... softplus) OUT OF LINE: inlined at the 36-72 sites of an unrolled staging loop the nine-way switch blew the
// run-time-activation instance of the lean conv kernel up to 17 K instructions (I-cache misses on the common relu / silu path: the RCAN trunk's
// SiLU forward went 8.8 -> 10.2 ms).  A real call costs the rare path a few cycles and the common path nothing.  (One copy per translation unit.)
static __device__ __attribute__((noinline)) float bpx_act_ext(float u, int act) {
  switch (act) {
    case BPX_ACT_LEAKY_RELU: return u > 0.f ? u : 0.01f * u;
    case BPX_ACT_GELU: return 0.5f * u * (1.f + bpx_erf(u * 0.70710678118654752f));
    case BPX_ACT_TANH: { const float e = expf(-2.f * fabsf(u)); return copysignf((1.f - e) / (1.f + e), u); }
    case BPX_ACT_SIGMOID: return 1.f / (1.f + expf(-u));
    case BPX_ACT_SOFTPLUS: return u > 20.f ? u : log1pf(expf(u));
    default: return u;
  }
}
static __device__ __attribute__((noinline)) float bpx_act_ext_bwd(float u, int act) {
  switch (act) {
    case BPX_ACT_LEAKY_RELU: return u > 0.f ? 1.f : 0.01f;
    case BPX_ACT_GELU: return 0.5f * (1.f + bpx_erf(u * 0.70710678118654752f)) + u * 0.39894228040143268f * expf(-0.5f * u * u);
    case BPX_ACT_TANH: { const float e = expf(-2.f * fabsf(u)); const float th = (1.f - e) / (1.f + e); return 1.f - th * th; }
    case BPX_ACT_SIGMOID: { const float s = 1.f / (1.f + expf(-u)); return s * (1.f - s); }
    case BPX_ACT_SOFTPLUS: return u > 20.f ? 1.f : 1.f / (1.f + expf(-u));
    default: return 1.f;
  }
}
// ONE run-time form for every block activation of the reference (blocks.py:1973-1998), used by every kernel's run-time-activation instance
// (conv prologues, dgrad epilogues, wgrad staging, the materialised norm + act pair, gate MLPs).  PRECISE: fp32 storage mode (libm exp / expm1
// instead of the fast forms).  Codes 0-3 inline (select chain), codes 4-8 through the out-of-line functions above.
// EXT = false: codes 0-3 only, no call in the kernel at all (measured: even the never-taken call costs the lean conv kernel's SiLU instance 8 % -
// RCAN trunk forward 8.2 -> 8.9 ms); the tuned kernels (lean conv, shift-dy wgrad, fused backward) are compiled that way and their launchers hand
// the codes 4-8 to the plain kernels' ACTK = 2 instances, which are compiled with EXT = true.
template <bool PRECISE, bool EXT = true> __device__ __forceinline__ float bpx_act_rt(float u, int act) {
  if (EXT && act > BPX_ACT_SILU) return bpx_act_ext(u, act);
  return act == BPX_ACT_ELU ? (u > 0.f ? u : (PRECISE ? expm1f(u) : (__expf(u) - 1.f)))
         : act == BPX_ACT_RELU ? fmaxf(u, 0.f) : act == BPX_ACT_SILU ? u / (1.f + (PRECISE ? expf(-u) : __expf(-u))) : u;
}
template <bool PRECISE, int N, bool EXT = true> __device__ __forceinline__ void bpx_act_vec(float* f, int act) {
#pragma unroll
  for (int e = 0; e < N; ++e) f[e] = bpx_act_rt<PRECISE, EXT>(f[e], act);
}
template <bool PRECISE, bool EXT = true> __device__ __forceinline__ float bpx_act_bwd_rt(float u, int act) {
  if (EXT && act > BPX_ACT_SILU) return bpx_act_ext_bwd(u, act);
  if (act == BPX_ACT_ELU) return u > 0.f ? 1.f : (PRECISE ? expf(u) : __expf(u));
  if (act == BPX_ACT_RELU) return u > 0.f ? 1.f : 0.f;
  if (act == BPX_ACT_SILU) { const float s = 1.f / (1.f + (PRECISE ? expf(-u) : __expf(-u))); return s * (1.f + u * (1.f - s)); }
  return 1.f;
}

// ---- helpers shared by the bf16 "lean" kernels (conv3d_lean.hip, wgrad.hip) -------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (= the 16 voxels of an MFMA column group); every lane receives the total
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  return v;
}

// normalised pair -> activated pair.  ELU(u) = med3(u, exp(u) - 1, 0): exp(u) - 1 >= u everywhere, so the median picks u
// for u > 0 and exp(u) - 1 for u <= 0 - one VALU op instead of compare + select; the multiplies/adds pack (v_pk_*_f32).
template <int ACTK> __device__ __forceinline__ void act_pair(float& a, float& b, int act) {
  if (ACTK == 1) {
    f32x2_t u{a, b};
    f32x2_t w = u * f32x2_t{1.44269504088896341f, 1.44269504088896341f};
    f32x2_t e = f32x2_t{__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])} + f32x2_t{-1.f, -1.f};
    a = __builtin_amdgcn_fmed3f(u[0], e[0], 0.f);
    b = __builtin_amdgcn_fmed3f(u[1], e[1], 0.f);
  } else {
    a = bpx_act_rt<false, false>(a, act);     // act_pair serves the tuned kernels only: codes 0-3
    b = bpx_act_rt<false, false>(b, act);
  }
}

// ---- MFMA step: 16 bytes of A and B per lane -> one (bf16) or four (f32) MFMAs ----------------
// D[i][j] += sum_k A[i][k] * B[k][j]; lane l supplies A[i=l&15][kgroup l>>4], B[kgroup l>>4][j=l&15];
// result lane l holds D[i=(l>>4)*4+r][j=l&15], r=0..3 (cdna_hip_programming.md section 3).
template <typename T> __device__ __forceinline__ f32x4_t mfma_step(const u32x4_t& a, const u32x4_t& b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t mfma_step<uint16_t>(const u32x4_t& a, const u32x4_t& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mfma_step<f16_t>(const u32x4_t& a, const u32x4_t& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mfma_step<float>(const u32x4_t& a, const u32x4_t& b, f32x4_t c) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
  return c;
}

__host__ __device__ __forceinline__ int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

static inline size_t dtype_size(int dt) {
  switch (dt) {
    case BPX_F32: return 4;
    case BPX_BF16: case BPX_F16: return 2;
    case BPX_U8: return 1;
  }
  return 0;
}

// wgrad.hip: dw[ci*si + co*sj + tap*st] = sum over the `groups` partial slabs [groups][taps][Cin][Cout] (fixed order, no atomics) and
// db[c] += sum of the rows [groups][ndb] (ndb = 0: Cout); may_defer: queued with the other weight-gradient reductions while the deferred mode is on
namespace bpxred {
bool defer_active();   // between bpx_wgrad_defer_begin and _flush on this host thread
int reduce_rows(const char* fn, const float* rows, int groups, int64_t stride, int n, float* dst, hipStream_t s);   // dst[i] += sum_g rows[g * stride + i] (deferrable)
int reduce_partials(const char* fn, const float* part, float* dw, int groups, int taps, int Cin, int Cout, int64_t si, int64_t sj, int64_t st,
                    const float* dbpart, float* db, int ndb, bool may_defer, hipStream_t s);
// the same with a second destination db2 of the bias sums (bwd_fused.hip: the shortcut bias of a residual block)
int reduce_partials2(const char* fn, const float* part, float* dw, int groups, int taps, int Cin, int Cout, int64_t si, int64_t sj, int64_t st,
                     const float* dbpart, float* db, float* db2, int ndb, bool may_defer, hipStream_t s);
}
