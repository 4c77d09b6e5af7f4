// Streaming kernels either side of the network: the input normalisation of biapy/data/norm.py
// (percentile clip :395-473, zero-mean / unit-variance :586-645) and the binarisation after the merge
// (biapy/engine/semantic_seg.py:418-431, threshold_otsu = 256-bin histogram + arg-max of the between-class variance).
// All of them are HBM-bound scans over volumes that already live on the device (a 1024^3 prediction is 4.3 GB - sending it
// to the host to take a percentile or a histogram costs more than the whole network).
//   * exact order statistics (what np.percentile interpolates between) by a 4-pass radix select on the order-preserving
//     integer image of the floats - no sort, 4 reads of the data, no host round trip between the passes;
//   * a histogram that reproduces np.histogram's float32 bin arithmetic operation by operation (index estimate, then the
//     two edge corrections against the SAME float32 edge table NumPy builds), so the counts are bit-identical;
//   * min / max and two-pass moments as per-block partials (reduced by the caller: a few KB).
#include <algorithm>

#include "bpx_common.h"

namespace {

__device__ __forceinline__ uint32_t fkey(float f) {  // monotone float -> uint32
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

struct SelectState { uint32_t prefix; uint32_t pad; unsigned long long k; unsigned long long hist[256]; };

__global__ void __launch_bounds__(256) select_init_kernel(SelectState* st, unsigned long long k) {
  if (threadIdx.x == 0) { st->prefix = 0; st->k = k; }
  st->hist[threadIdx.x] = 0;
}

// histogram of byte (key >> shift) & 255 over the elements whose higher bytes equal the prefix found so far
__global__ void __launch_bounds__(256) select_hist_kernel(const float* __restrict__ x, int64_t n, int shift, SelectState* st) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t prefix = st->prefix;
  const uint32_t himask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4_t v = reinterpret_cast<const f32x4_t*>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t key = fkey(v[e]);
      if ((key & himask) == (prefix & himask)) atomicAdd(&h[(key >> shift) & 255u], 1u);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
    const uint32_t key = fkey(x[n4 * 4 + threadIdx.x]);
    if ((key & himask) == (prefix & himask)) atomicAdd(&h[(key >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// picks the bucket that holds rank k, extends the prefix, re-bases k, clears the histogram; last pass writes the value
__global__ void __launch_bounds__(256) select_pick_kernel(SelectState* st, int shift, float* out) {
  __shared__ unsigned long long c[256];
  c[threadIdx.x] = st->hist[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long k = st->k, cum = 0;
    int b = 0;
    for (; b < 255; ++b) {
      if (k < cum + c[b]) break;
      cum += c[b];
    }
    st->prefix |= (uint32_t)b << shift;
    st->k = k - cum;
    if (shift == 0) *out = fkey_inv(st->prefix);
  }
  st->hist[threadIdx.x] = 0;
}

__global__ void __launch_bounds__(256) minmax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
  float mn = INFINITY, mx = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { mn = fminf(mn, __shfl_xor(mn, m, 64)); mx = fmaxf(mx, __shfl_xor(mx, m, 64)); }
  __shared__ float r[4][2];
  if ((threadIdx.x & 63) == 0) { r[threadIdx.x >> 6][0] = mn; r[threadIdx.x >> 6][1] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = fminf(fminf(r[0][0], r[1][0]), fminf(r[2][0], r[3][0]));
    part[2 * blockIdx.x + 1] = fmaxf(fmaxf(r[0][1], r[1][1]), fmaxf(r[2][1], r[3][1]));
  }
}

// part[b] = sum over the block's elements of (x - center)^power, power 1 or 2, in double
__global__ void __launch_bounds__(256) moment_kernel(const float* __restrict__ x, int64_t n, double center, int power, double* __restrict__ part) {
  double s = 0.0;
  const int64_t n4 = (((uintptr_t)x & 15) == 0) ? n / 4 : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4_t v = reinterpret_cast<const f32x4_t*>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double d = (double)v[e] - center;
      s += power == 2 ? d * d : d;
    }
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double d = (double)x[i] - center;
    s += power == 2 ? d * d : d;
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
  __shared__ double r[4];
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (r[0] + r[1]) + (r[2] + r[3]);
}

// np.histogram(a, bins=nbins, range=(first, last)) for a float32 array `a`, NumPy >= 2 (float32 bin arithmetic):
//   f = ((a - first) / (last - first)) * nbins ; i = int(f) ; i -= (i == nbins) ; i -= (a < edges[i]) ;
//   i += (a >= edges[i+1]) & (i != nbins-1)            (numpy/lib/_histograms_impl.py, uniform-bin fast path)
__global__ void __launch_bounds__(256) hist_uniform_kernel(const float* __restrict__ x, int64_t n, float first, float denom, int nbins,
                                                           const float* __restrict__ edges, unsigned long long* __restrict__ counts) {
  extern __shared__ unsigned char sm[];
  float* e = reinterpret_cast<float*>(sm);                      // [nbins + 1]
  uint32_t* h = reinterpret_cast<uint32_t*>(sm + (size_t)(nbins + 1) * 4);   // [copies][nbins]: one copy per wave when it fits
  const int copies = nbins <= 1024 ? 4 : 1;
  for (int i = threadIdx.x; i <= nbins; i += blockDim.x) e[i] = edges[i];
  for (int i = threadIdx.x; i < copies * nbins; i += blockDim.x) h[i] = 0;
  __syncthreads();
  uint32_t* hw = h + (copies == 4 ? (threadIdx.x >> 6) * nbins : 0);
  const float fn = (float)nbins, last = e[nbins];
  auto count = [&](float a) {
    if (!(a >= first && a <= last)) return;                   // out of range (and NaN): not counted
    int idx = (int)(((a - first) / denom) * fn);
    if (idx == nbins) --idx;
    if (a < e[idx]) --idx;
    else if (a >= e[idx + 1] && idx != nbins - 1) ++idx;
    atomicAdd(&hw[idx], 1u);
  };
  const int64_t n4 = (((uintptr_t)x & 15) == 0) ? n / 4 : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4_t v = reinterpret_cast<const f32x4_t*>(x)[i];
    count(v[0]); count(v[1]); count(v[2]); count(v[3]);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) count(x[i]);
  __syncthreads();
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
    uint32_t c = h[i];
    if (copies == 4) c += h[nbins + i] + h[2 * nbins + i] + h[3 * nbins + i];
    if (c) atomicAdd(&counts[i], (unsigned long long)c);
  }
}

__global__ void __launch_bounds__(256) threshold_kernel(const float* __restrict__ x, int64_t n, float thr, uint8_t* __restrict__ out) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4_t v = reinterpret_cast<const f32x4_t*>(x)[i];
    reinterpret_cast<uint32_t*>(out)[i] = (v[0] > thr ? 1u : 0u) | (v[1] > thr ? 0x100u : 0u) | (v[2] > thr ? 0x10000u : 0u) | (v[3] > thr ? 0x1000000u : 0u);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) out[n4 * 4 + threadIdx.x] = x[n4 * 4 + threadIdx.x] > thr ? 1 : 0;
}

// out = (clip(x, lo, hi) - sub) / div in float32 operations (np.clip followed by (data - mean) / std on a float32 array)
__global__ void __launch_bounds__(256) clip_affine_kernel(const float* __restrict__ x, int64_t n, float lo, float hi, float sub, float div,
                                                          float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (fminf(fmaxf(x[i], lo), hi) - sub) / div;
}

int blocks_for(int64_t n) { return (int)std::min<int64_t>(std::max<int64_t>(1, (n + 1023) / 1024), 4096); }

}  // namespace

extern "C" int bpx_scan_blocks(int64_t n) { return blocks_for(n); }

extern "C" int64_t bpx_select_workspace(void) { return (int64_t)sizeof(SelectState); }

extern "C" int bpx_select_kth_f32(const float* x_d, int64_t n, int64_t k, float* out_d, void* ws_d, bpx_stream_t stream) {
  const char* fn = "bpx_select_kth_f32";
  BPX_CHECK(x_d && out_d && ws_d, "%s: null pointer", fn);
  BPX_CHECK(n > 0 && k >= 0 && k < n, "%s: rank %lld outside [0, %lld)", fn, (long long)k, (long long)n);
  BPX_CHECK(((uintptr_t)x_d & 15) == 0 && ((uintptr_t)ws_d & 7) == 0, "%s: x must be 16-byte aligned, the workspace 8-byte aligned", fn);
  hipStream_t s = (hipStream_t)stream;
  SelectState* st = reinterpret_cast<SelectState*>(ws_d);
  select_init_kernel<<<1, 256, 0, s>>>(st, (unsigned long long)k);
  for (int shift = 24; shift >= 0; shift -= 8) {
    select_hist_kernel<<<blocks_for(n / 4), 256, 0, s>>>(x_d, n, shift, st);
    select_pick_kernel<<<1, 256, 0, s>>>(st, shift, out_d);
  }
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

extern "C" int bpx_minmax_f32(const float* x_d, int64_t n, float* partials_d, bpx_stream_t stream) {
  const char* fn = "bpx_minmax_f32";
  BPX_CHECK(x_d && partials_d && n > 0, "%s: bad arguments", fn);
  minmax_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream>>>(x_d, n, partials_d);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

extern "C" int bpx_moment_f32(const float* x_d, int64_t n, double center, int power, double* partials_d, bpx_stream_t stream) {
  const char* fn = "bpx_moment_f32";
  BPX_CHECK(x_d && partials_d && n > 0, "%s: bad arguments", fn);
  BPX_CHECK(power == 1 || power == 2, "%s: power must be 1 or 2", fn);
  moment_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream>>>(x_d, n, center, power, partials_d);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

extern "C" int bpx_histogram_f32(const float* x_d, int64_t n, float first_edge, float last_edge, int nbins, const float* edges_d,
                                 unsigned long long* counts_d, bpx_stream_t stream) {
  const char* fn = "bpx_histogram_f32";
  BPX_CHECK(x_d && edges_d && counts_d && n > 0, "%s: bad arguments", fn);
  BPX_CHECK(nbins >= 1 && nbins <= 4096, "%s: nbins must be in [1, 4096]", fn);
  BPX_CHECK(last_edge > first_edge, "%s: empty range", fn);
  const size_t shm = (size_t)(nbins + 1) * 4 + (size_t)nbins * 4 * (nbins <= 1024 ? 4 : 1);
  hist_uniform_kernel<<<blocks_for(n), 256, shm, (hipStream_t)stream>>>(x_d, n, first_edge, last_edge - first_edge, nbins, edges_d, counts_d);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

extern "C" int bpx_threshold_u8(const float* x_d, int64_t n, float thr, uint8_t* out_d, bpx_stream_t stream) {
  const char* fn = "bpx_threshold_u8";
  BPX_CHECK(x_d && out_d && n > 0, "%s: bad arguments", fn);
  BPX_CHECK((((uintptr_t)x_d) & 15) == 0 && (((uintptr_t)out_d) & 3) == 0, "%s: x must be 16-byte, out 4-byte aligned", fn);
  threshold_kernel<<<blocks_for(n / 4), 256, 0, (hipStream_t)stream>>>(x_d, n, thr, out_d);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

extern "C" int bpx_clip_affine_f32(const float* x_d, int64_t n, float lo, float hi, float sub, float div, float* out_d, bpx_stream_t stream) {
  const char* fn = "bpx_clip_affine_f32";
  BPX_CHECK(x_d && out_d && n > 0, "%s: bad arguments", fn);
  clip_affine_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream>>>(x_d, n, lo, hi, sub, div, out_d);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

// ---- class head of the sliding-window harness (base_workflow.py:2135-2141): the last k channels of the blended prediction become ONE
// channel holding np.argmax over them (first maximum wins), the leading C - k channels pass through:
//   out[v][0 .. C-k) = in[v][0 .. C-k),  out[v][C-k] = (float) argmax_q in[v][C-k+q]
__global__ void __launch_bounds__(256) class_argmax_kernel(const float* __restrict__ in, int64_t vox, int C, int k, float* __restrict__ out) {
  const int Co = C - k + 1;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < vox; v += (int64_t)gridDim.x * 256) {
    const float* p = in + v * C;
    float* o = out + v * Co;
    for (int c = 0; c < C - k; ++c) o[c] = p[c];
    int best = 0;
    float bv = p[C - k];
    for (int q = 1; q < k; ++q) {
      const float x = p[C - k + q];
      if (x > bv) { bv = x; best = q; }      // strict >: the first maximum, as np.argmax (a NaN never wins here; np.argmax returns the first NaN)
    }
    o[C - k] = (float)best;
  }
}

extern "C" int bpx_class_argmax(const float* in_d, int64_t voxels, int C, int k, float* out_d, bpx_stream_t stream) {
  const char* fn = "bpx_class_argmax";
  BPX_CHECK(in_d && out_d && voxels > 0, "%s: bad arguments", fn);
  BPX_CHECK(k >= 1 && k <= C, "%s: %d class channels of %d", fn, k, C);
  class_argmax_kernel<<<blocks_for(voxels), 256, 0, (hipStream_t)stream>>>(in_d, voxels, C, k, out_d);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

// ---- test-time augmentation: signed axis permutations of (Z,Y,X,C) volumes ---------------------------
// biapy/data/post_processing/tta.py:64-196 (AxisTransform: output axis a comes from input axis perm[a], reversed when
// sign[a] == -1) and post_processing.py:1386-1540 (ensemble_predictions: predict every orientation, undo it, reduce).
// orient    : out = t.apply(in)                                   (gather, out has the permuted extents)
// accumulate: acc (op)= t.inverse.apply(pred_t), op = first ? assign : add / min / max - called once per orientation in the
//             reference's order, so the float32 sum is the same sequential sum np.mean(stack, axis=0) forms.
namespace {
struct Orient { int n_in[3]; int perm[3]; int sign[3]; int C; };

__global__ void __launch_bounds__(256) tta_orient_kernel(const float* __restrict__ in, float* __restrict__ out, Orient t, int64_t total) {
  const int no[3] = {t.n_in[t.perm[0]], t.n_in[t.perm[1]], t.n_in[t.perm[2]]};
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(q % t.C);
    int64_t v = q / t.C;
    int o[3];
    o[2] = (int)(v % no[2]); v /= no[2];
    o[1] = (int)(v % no[1]); o[0] = (int)(v / no[1]);
    int i[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) i[t.perm[a]] = t.sign[a] > 0 ? o[a] : no[a] - 1 - o[a];
    out[q] = in[(((int64_t)i[0] * t.n_in[1] + i[1]) * t.n_in[2] + i[2]) * t.C + c];
  }
}

// acc is in the un-oriented frame (extents n_in), pred in the oriented frame; mode 0 mean-sum, 1 min, 2 max
__global__ void __launch_bounds__(256) tta_accumulate_kernel(const float* __restrict__ pred, float* __restrict__ acc, Orient t, int mode,
                                                             int first, float scale, int64_t total) {
  const int no[3] = {t.n_in[t.perm[0]], t.n_in[t.perm[1]], t.n_in[t.perm[2]]};
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(q % t.C);
    int64_t v = q / t.C;
    int i[3];
    i[2] = (int)(v % t.n_in[2]); v /= t.n_in[2];
    i[1] = (int)(v % t.n_in[1]); i[0] = (int)(v / t.n_in[1]);
    int o[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) o[a] = t.sign[a] > 0 ? i[t.perm[a]] : no[a] - 1 - i[t.perm[a]];
    const float p = pred[(((int64_t)o[0] * no[1] + o[1]) * no[2] + o[2]) * t.C + c];
    float r = first ? p : (mode == 0 ? acc[q] + p : mode == 1 ? fminf(acc[q], p) : fmaxf(acc[q], p));
    if (scale != 1.f) r = r / scale;     // last orientation of "mean": the sum divided by the count, as np.mean does
    acc[q] = r;
  }
}

int fill_orient(const char* fn, Orient& t, int Z, int Y, int X, int C, const int* perm, const int* sign) {
  BPX_CHECK(Z > 0 && Y > 0 && X > 0 && C > 0, "%s: empty volume", fn);
  bool seen[3] = {false, false, false};
  for (int a = 0; a < 3; ++a) {
    BPX_CHECK(perm[a] >= 0 && perm[a] < 3 && !seen[perm[a]], "%s: perm must be a permutation of (0,1,2)", fn);
    BPX_CHECK(sign[a] == 1 || sign[a] == -1, "%s: sign entries must be +1 or -1", fn);
    seen[perm[a]] = true;
    t.perm[a] = perm[a]; t.sign[a] = sign[a];
  }
  t.n_in[0] = Z; t.n_in[1] = Y; t.n_in[2] = X; t.C = C;
  return 0;
}
}  // namespace

extern "C" int bpx_tta_orient(const float* in_d, int Z, int Y, int X, int C, const int* perm, const int* sign, float* out_d, bpx_stream_t stream) {
  const char* fn = "bpx_tta_orient";
  BPX_CHECK(in_d && out_d && perm && sign, "%s: null pointer", fn);
  Orient t;
  if (fill_orient(fn, t, Z, Y, X, C, perm, sign)) return 1;
  const int64_t total = (int64_t)Z * Y * X * C;
  tta_orient_kernel<<<blocks_for(total), 256, 0, (hipStream_t)stream>>>(in_d, out_d, t, total);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

extern "C" int bpx_tta_accumulate(const float* pred_d, int Z, int Y, int X, int C, const int* perm, const int* sign, int mode, int first,
                                  int count_if_last, float* acc_d, bpx_stream_t stream) {
  const char* fn = "bpx_tta_accumulate";
  BPX_CHECK(pred_d && acc_d && perm && sign, "%s: null pointer", fn);
  BPX_CHECK(mode >= 0 && mode <= 2, "%s: mode must be 0 (mean), 1 (min) or 2 (max)", fn);
  Orient t;
  if (fill_orient(fn, t, Z, Y, X, C, perm, sign)) return 1;
  const int64_t total = (int64_t)Z * Y * X * C;
  tta_accumulate_kernel<<<blocks_for(total), 256, 0, (hipStream_t)stream>>>(pred_d, acc_d, t, mode, first, (mode == 0 && count_if_last > 0) ? (float)count_if_last : 1.f, total);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}
