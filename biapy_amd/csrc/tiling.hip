// Overlap tiling on device: patch gather (crop) and deterministic blend (merge).
//
// Both kernels are HBM-bound scans: every input element is read once, every output
// element written once, x is the fastest thread index so a wave touches 64 consecutive voxels of one
// (z,y) row.  The blend is a GATHER in the reference's patch order so that the fp32 sums are
// bit-identical to NumPy's scatter `merged[sl] += data[c] * w` (data_3D_manipulation.py:838-844)
// and independent of scheduling - no atomics.
#include <hip/hip_fp16.h>

#include "bpx_common.h"

struct AxisG {
  int n, step, last, patch, limit;
  __device__ __forceinline__ int start(int i) const {
    int s = i * step;
    return s - ((s + patch < limit) ? 0 : last);
  }
};
static inline AxisG to_axis(const bpx_axis_grid& g) { return AxisG{g.n, g.step, g.last, g.patch, g.limit}; }

// Division of a 31-bit index by a launch-uniform divisor as multiply + shift (the row kernels decompose a flat thread index
// five times; a runtime 64-bit division is ~100 VALU instructions, this is 3): q = (n * m) >> p with p = 31 + ceil(log2 d),
// m = floor(2^p / d) + 1 - exact for 0 <= n < 2^31 (the launchers check the range).
struct FastDiv {
  uint32_t d, m, p;
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return (uint32_t)(((uint64_t)n * m) >> p); }
  __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const { q = div(n); r = n - q * d; }
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  if (d == 0) d = 1;                                   // degenerate step (see cover_range): the quotient is not used
  f.d = d;
  uint32_t k = 0;
  while ((1ull << k) < d) ++k;
  f.p = 31 + k;
  f.m = (uint32_t)(((1ull << f.p) / d) + 1);
  return f;
}

// test / A-B hook: 1 = force the element-per-thread kernels of round 1 (bpx_debug_set_tiling_scalar)
static int g_tiling_scalar = 0;
extern "C" int bpx_debug_set_tiling_scalar(int on) { g_tiling_scalar = on; return 0; }

// ------------------------------------------------------------------------------------------------
// crop
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pad_src(int q, int n, int mode, bool& inside) {
  // q: coordinate in the un-padded volume, possibly outside [0,n)
  if (q >= 0 && q < n) return q;
  if (mode == BPX_PAD_ZEROS) { inside = false; return 0; }
  return q < 0 ? -q : 2 * (n - 1) - q;  // np.pad(..., "reflect"): mirror without repeating the edge
}

template <typename E>
__global__ void __launch_bounds__(256) crop3d_kernel(const E* __restrict__ vol, E* __restrict__ out, int Z, int Y, int X, int C,
                                                     int pz, int py, int px, int mode, AxisG gz, AxisG gy, AxisG gx,
                                                     int64_t c_begin, int64_t total) {
  const int Pz = gz.patch, Py = gy.patch, Px = gx.patch;
  const int64_t row = (int64_t)Px * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / row;
    int xc = (int)(i - r * row);
    int lx = xc / C, ch = xc - lx * C;
    int ly = (int)(r % Py); r /= Py;
    int lz = (int)(r % Pz); r /= Pz;
    int64_t c = c_begin + r;
    int ix = (int)(c % gx.n); c /= gx.n;
    int iy = (int)(c % gy.n);
    int iz = (int)(c / gy.n);
    bool inside = true;
    int sz = pad_src(gz.start(iz) + lz - pz, Z, mode, inside);
    int sy = pad_src(gy.start(iy) + ly - py, Y, mode, inside);
    int sx = pad_src(gx.start(ix) + lx - px, X, mode, inside);
    E v = 0;
    if (inside) v = vol[(((int64_t)sz * Y + sy) * X + sx) * C + ch];
    out[i] = v;
  }
}


// Vectorised form (the production path): one thread moves 16 bytes of one patch row.  An output row (Px*C elements) is a shifted
// copy of a run of the source row in the flattened (x, c) index, so whenever the 16 bytes lie inside the volume they are ONE
// (possibly misaligned) 16-byte load; only vectors that touch the padded border go element by element.  The z / y source rows,
// the patch index decomposition and the pad rule are evaluated once per 16 bytes instead of once per element, with 32-bit math.
template <int ES> struct CropElem;
template <> struct CropElem<1> { typedef uint8_t type; };
template <> struct CropElem<2> { typedef uint16_t type; };
template <> struct CropElem<4> { typedef uint32_t type; };

template <int ES>
__global__ void __launch_bounds__(256) crop3d_row_kernel(const unsigned char* __restrict__ vol, unsigned char* __restrict__ out, int Z, int Y,
                                                         int X, int C, int pz, int py, int px, int mode, AxisG gz, AxisG gy, AxisG gx,
                                                         uint32_t c_begin_lo, uint32_t total_vec, FastDiv dq, FastDiv dPy, FastDiv dPz, FastDiv dnx,
                                                         FastDiv dny, FastDiv dC) {
  typedef typename CropElem<ES>::type E;
  constexpr int VEC = 16 / ES;
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= total_vec) return;
  uint32_t rowid, q, r1, ly, r2, lz, c1, ix, iz, iy;
  dq.divmod(t, rowid, q);
  dPy.divmod(rowid, r1, ly);
  dPz.divmod(r1, r2, lz);
  dnx.divmod(c_begin_lo + r2, c1, ix);
  dny.divmod(c1, iz, iy);
  bool inside = true;
  const int sz = pad_src(gz.start((int)iz) + (int)lz - pz, Z, mode, inside);
  const int sy = pad_src(gy.start((int)iy) + (int)ly - py, Y, mode, inside);
  const int e0 = (int)q * VEC;
  const int x0 = (int)dC.div((uint32_t)e0), ch0 = e0 - x0 * C;
  const int xs = gx.start((int)ix) - px;                           // source x of patch column 0
  const int xl = (int)dC.div((uint32_t)(e0 + VEC - 1));            // patch column of the vector's last element
  const unsigned char* srow = vol + ((size_t)sz * Y + sy) * (size_t)X * C * ES;
  u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
  if (inside) {
    if (xs + x0 >= 0 && xs + xl < X) {
      __builtin_memcpy(&v, srow + ((size_t)(xs + x0) * C + ch0) * ES, 16);   // misaligned 16-byte load
    } else {
      E tmp[VEC];
      int x = x0, ch = ch0;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        bool in2 = true;
        const int sx = pad_src(xs + x, X, mode, in2);
        tmp[k] = in2 ? reinterpret_cast<const E*>(srow)[(size_t)sx * C + ch] : (E)0;
        if (++ch == C) { ch = 0; ++x; }
      }
      __builtin_memcpy(&v, tmp, 16);
    }
  }
  *reinterpret_cast<u32x4_t*>(out + (size_t)t * 16) = v;
}

extern "C" int bpx_crop3d_gather(const void* vol_d, int elem_size, int Z, int Y, int X, int C, int pad_z, int pad_y, int pad_x,
                                 int pad_mode, const bpx_axis_grid* g, int64_t c_begin, int64_t c_count, void* out_d,
                                 bpx_stream_t stream) {
  BPX_CHECK(vol_d && out_d && g, "bpx_crop3d_gather: null pointer");
  BPX_CHECK(elem_size == 1 || elem_size == 2 || elem_size == 4, "bpx_crop3d_gather: elem_size %d unsupported", elem_size);
  BPX_CHECK(pad_mode == BPX_PAD_REFLECT || pad_mode == BPX_PAD_ZEROS, "bpx_crop3d_gather: bad pad_mode %d", pad_mode);
  BPX_CHECK(pad_z < Z && pad_y < Y && pad_x < X, "bpx_crop3d_gather: padding must be smaller than the volume");
  int64_t n_all = (int64_t)g[0].n * g[1].n * g[2].n;
  BPX_CHECK(c_begin >= 0 && c_count >= 0 && c_begin + c_count <= n_all, "bpx_crop3d_gather: patch range out of grid");
  int64_t total = c_count * g[0].patch * g[1].patch * g[2].patch * C;
  if (total == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  AxisG gz = to_axis(g[0]), gy = to_axis(g[1]), gx = to_axis(g[2]);
  const int64_t row_bytes = (int64_t)g[2].patch * C * elem_size;
  if (row_bytes % 16 == 0 && ((uintptr_t)out_d & 15) == 0 && !g_tiling_scalar && n_all < (1ll << 30)) {
    // 31-bit thread indices per launch (FastDiv): a larger request is cut into launches of whole patches
    const int64_t vec_per_patch = (int64_t)g[0].patch * g[1].patch * g[2].patch * C * elem_size / 16;
    const int qpr = (int)(row_bytes / 16);
    BPX_CHECK(vec_per_patch < (1ll << 30), "bpx_crop3d_gather: patch too large");
    const int64_t per_launch = std::max<int64_t>(1, ((1ll << 30) - 1) / vec_per_patch);
    const FastDiv dq = make_fastdiv((uint32_t)qpr), dPy = make_fastdiv((uint32_t)g[1].patch), dPz = make_fastdiv((uint32_t)g[0].patch),
                  dnx = make_fastdiv((uint32_t)g[2].n), dny = make_fastdiv((uint32_t)g[1].n), dC = make_fastdiv((uint32_t)C);
    for (int64_t c0 = 0; c0 < c_count; c0 += per_launch) {
      const int64_t cn = std::min(per_launch, c_count - c0);
      const uint32_t total_vec = (uint32_t)(cn * vec_per_patch);
      const unsigned nb = (unsigned)cdiv64(total_vec, 256);
      const unsigned char* v8 = (const unsigned char*)vol_d;
      unsigned char* o8 = (unsigned char*)out_d + (size_t)c0 * vec_per_patch * 16;
      const uint32_t cb = (uint32_t)(c_begin + c0);
      if (elem_size == 4) crop3d_row_kernel<4><<<nb, 256, 0, s>>>(v8, o8, Z, Y, X, C, pad_z, pad_y, pad_x, pad_mode, gz, gy, gx, cb, total_vec, dq, dPy, dPz, dnx, dny, dC);
      else if (elem_size == 2) crop3d_row_kernel<2><<<nb, 256, 0, s>>>(v8, o8, Z, Y, X, C, pad_z, pad_y, pad_x, pad_mode, gz, gy, gx, cb, total_vec, dq, dPy, dPz, dnx, dny, dC);
      else crop3d_row_kernel<1><<<nb, 256, 0, s>>>(v8, o8, Z, Y, X, C, pad_z, pad_y, pad_x, pad_mode, gz, gy, gx, cb, total_vec, dq, dPy, dPz, dnx, dny, dC);
    }
    BPX_LAUNCH_CHECK("bpx_crop3d_gather");
    return 0;
  }
  int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 256 * 16);
  if (elem_size == 4)
    crop3d_kernel<uint32_t><<<blocks, 256, 0, s>>>((const uint32_t*)vol_d, (uint32_t*)out_d, Z, Y, X, C, pad_z, pad_y, pad_x, pad_mode, gz, gy, gx, c_begin, total);
  else if (elem_size == 2)
    crop3d_kernel<uint16_t><<<blocks, 256, 0, s>>>((const uint16_t*)vol_d, (uint16_t*)out_d, Z, Y, X, C, pad_z, pad_y, pad_x, pad_mode, gz, gy, gx, c_begin, total);
  else
    crop3d_kernel<uint8_t><<<blocks, 256, 0, s>>>((const uint8_t*)vol_d, (uint8_t*)out_d, Z, Y, X, C, pad_z, pad_y, pad_x, pad_mode, gz, gy, gx, c_begin, total);
  BPX_LAUNCH_CHECK("bpx_crop3d_gather");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// by-chunks tiler (biapy/data/generators/chunked_test_pair_data_generator.py:440-565, base_workflow.py:2603-2610):
// a patch is the padded read region of one chunk, clipped to the volume and completed by np.pad(..., "reflect"); its
// prediction is written back without the padding.  The reflection is about the edges of the CLIPPED region, so the host
// expresses it as three source-index tables per patch (chunked.ChunkGrid.index_tables) and the gather is table driven.
// ------------------------------------------------------------------------------------------------
template <typename E>
__global__ void __launch_bounds__(256) gather3d_tables_kernel(const E* __restrict__ vol, int Y, int X, int C, const int* __restrict__ tables,
                                                              int Pz, int Py, int Px, E* __restrict__ out, int64_t total) {
  const int64_t per = (int64_t)Pz * Py * Px * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / per;
    int64_t r = i - b * per;
    const int c = (int)(r % C); r /= C;
    const int x = (int)(r % Px); r /= Px;
    const int y = (int)(r % Py);
    const int z = (int)(r / Py);
    const int* t = tables + b * (int64_t)(Pz + Py + Px);
    const int sz = t[z], sy = t[Pz + y], sx = t[Pz + Py + x];     // a negative index = outside the volume: zero (space-to-batch padding)
    out[i] = (sz | sy | sx) < 0 ? (E)0 : vol[(((size_t)sz * Y + sy) * X + sx) * C + c];
  }
}

// inverse of the gather: vol[tz[z], ty[y], tx[x], c] = in[b, z, y, x, c] for non-negative table entries (batch-to-space of the
// dilated convolutions: every destination voxel belongs to exactly one (b, z, y, x))
template <typename E>
__global__ void __launch_bounds__(256) scatter3d_tables_kernel(const E* __restrict__ in, const int* __restrict__ tables, int Pz, int Py, int Px,
                                                               E* __restrict__ vol, int Y, int X, int C, int64_t total) {
  const int64_t per = (int64_t)Pz * Py * Px * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / per;
    int64_t r = i - b * per;
    const int c = (int)(r % C); r /= C;
    const int x = (int)(r % Px); r /= Px;
    const int y = (int)(r % Py);
    const int z = (int)(r / Py);
    const int* t = tables + b * (int64_t)(Pz + Py + Px);
    const int sz = t[z], sy = t[Pz + y], sx = t[Pz + Py + x];
    if ((sz | sy | sx) >= 0) vol[(((size_t)sz * Y + sy) * X + sx) * C + c] = in[i];
  }
}

// 16-byte forms of the two table kernels (channels-last tensors whose voxel is a whole number of 16-byte vectors: the space-to-batch
// transport of the dilated convolutions moves 16 / 32-channel bf16 voxels).  One thread moves one vector; the flat index is decomposed
// with 32-bit multiply-shift divisions.  ASPP of cfg 4 (80^3 x 32 channels, rate 18): 87 -> see profiles/r02_breakdown_resunetpp_events.txt
template <bool SCATTER>
__global__ void __launch_bounds__(256) tables3d_vec_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, int Y, int X, int cv,
                                                           const int* __restrict__ tables, int Pz, int Py, int Px, uint32_t total, FastDiv dcv,
                                                           FastDiv dPx, FastDiv dPy, FastDiv dPz) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= total) return;
  uint32_t vox, c, r1, x, r2, y, b, z;
  dcv.divmod(t, vox, c);
  dPx.divmod(vox, r1, x);
  dPy.divmod(r1, r2, y);
  dPz.divmod(r2, b, z);
  const int* tt = tables + (size_t)b * (size_t)(Pz + Py + Px);
  const int sz = tt[z], sy = tt[Pz + y], sx = tt[Pz + Py + x];
  const bool in = (sz | sy | sx) >= 0;
  const size_t vi = (((size_t)sz * Y + sy) * X + sx) * (size_t)cv + c;      // vector index inside the volume (used when in)
  if (SCATTER) {
    if (in) dst[vi] = src[t];
  } else {
    dst[t] = in ? src[vi] : u32x4_t{0u, 0u, 0u, 0u};
  }
}

static bool tables_vec_ok(const void* a, const void* b, int elem_size, int C, int64_t total_elems) {
  return ((int64_t)C * elem_size) % 16 == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0 && total_elems * elem_size / 16 < (1ll << 31) && !g_tiling_scalar;
}

template <bool SCATTER>
static void launch_tables_vec(const void* src, void* dst, int elem_size, int Y, int X, int C, const int* tables_d, int Pz, int Py, int Px,
                              int64_t total_elems, hipStream_t s) {
  const int cv = C * elem_size / 16;
  const uint32_t total = (uint32_t)(total_elems * elem_size / 16);
  tables3d_vec_kernel<SCATTER><<<(unsigned)cdiv64(total, 256), 256, 0, s>>>((const u32x4_t*)src, (u32x4_t*)dst, Y, X, cv, tables_d, Pz, Py, Px, total,
                                                                              make_fastdiv((uint32_t)cv), make_fastdiv((uint32_t)Px),
                                                                              make_fastdiv((uint32_t)Py), make_fastdiv((uint32_t)Pz));
}

// regions: per patch {src z0,y0,x0 (first voxel kept, i.e. the padding stripped), dst z0,y0,x0, length z,y,x}
__global__ void __launch_bounds__(256) scatter3d_regions_kernel(const float* __restrict__ pred, int Py, int Px, int C, int64_t patch_elems,
                                                                const int* __restrict__ regions, float* __restrict__ out, int Y, int X) {
  const int b = blockIdx.y;
  const int* r = regions + b * 9;
  const int lz = r[6], ly = r[7], lx = r[8];
  const int64_t n = (int64_t)lz * ly * lx * C;
  const float* src = pred + (size_t)b * patch_elems;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t q = i;
    const int c = (int)(q % C); q /= C;
    const int x = (int)(q % lx); q /= lx;
    const int y = (int)(q % ly);
    const int z = (int)(q / ly);
    out[(((size_t)(r[3] + z) * Y + r[4] + y) * X + r[5] + x) * C + c] = src[(((size_t)(r[0] + z) * Py + r[1] + y) * Px + r[2] + x) * C + c];
  }
}

extern "C" int bpx_gather3d_tables(const void* vol_d, int elem_size, int Z, int Y, int X, int C, const int* tables_d, int n, int Pz, int Py, int Px,
                                   void* out_d, bpx_stream_t stream) {
  const char* fn = "bpx_gather3d_tables";
  BPX_CHECK(vol_d && out_d && tables_d, "%s: null pointer", fn);
  BPX_CHECK(elem_size == 1 || elem_size == 2 || elem_size == 4, "%s: elem_size %d unsupported", fn, elem_size);
  BPX_CHECK(Z > 0 && Y > 0 && X > 0 && C > 0 && n >= 0 && Pz > 0 && Py > 0 && Px > 0, "%s: bad extents", fn);
  const int64_t total = (int64_t)n * Pz * Py * Px * C;
  if (total == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (tables_vec_ok(vol_d, out_d, elem_size, C, total)) {
    launch_tables_vec<false>(vol_d, out_d, elem_size, Y, X, C, tables_d, Pz, Py, Px, total, s);
    BPX_LAUNCH_CHECK(fn);
    return 0;
  }
  const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 256 * 16);
  if (elem_size == 4) gather3d_tables_kernel<uint32_t><<<blocks, 256, 0, s>>>((const uint32_t*)vol_d, Y, X, C, tables_d, Pz, Py, Px, (uint32_t*)out_d, total);
  else if (elem_size == 2) gather3d_tables_kernel<uint16_t><<<blocks, 256, 0, s>>>((const uint16_t*)vol_d, Y, X, C, tables_d, Pz, Py, Px, (uint16_t*)out_d, total);
  else gather3d_tables_kernel<uint8_t><<<blocks, 256, 0, s>>>((const uint8_t*)vol_d, Y, X, C, tables_d, Pz, Py, Px, (uint8_t*)out_d, total);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

extern "C" int bpx_scatter3d_tables(const void* in_d, int elem_size, const int* tables_d, int n, int Pz, int Py, int Px, void* vol_d, int Z, int Y,
                                    int X, int C, bpx_stream_t stream) {
  const char* fn = "bpx_scatter3d_tables";
  BPX_CHECK(vol_d && in_d && tables_d, "%s: null pointer", fn);
  BPX_CHECK(elem_size == 1 || elem_size == 2 || elem_size == 4, "%s: elem_size %d unsupported", fn, elem_size);
  BPX_CHECK(Z > 0 && Y > 0 && X > 0 && C > 0 && n >= 0 && Pz > 0 && Py > 0 && Px > 0, "%s: bad extents", fn);
  const int64_t total = (int64_t)n * Pz * Py * Px * C;
  if (total == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (tables_vec_ok(in_d, vol_d, elem_size, C, total)) {
    launch_tables_vec<true>(in_d, vol_d, elem_size, Y, X, C, tables_d, Pz, Py, Px, total, s);
    BPX_LAUNCH_CHECK(fn);
    return 0;
  }
  const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 256 * 16);
  if (elem_size == 4) scatter3d_tables_kernel<uint32_t><<<blocks, 256, 0, s>>>((const uint32_t*)in_d, tables_d, Pz, Py, Px, (uint32_t*)vol_d, Y, X, C, total);
  else if (elem_size == 2) scatter3d_tables_kernel<uint16_t><<<blocks, 256, 0, s>>>((const uint16_t*)in_d, tables_d, Pz, Py, Px, (uint16_t*)vol_d, Y, X, C, total);
  else scatter3d_tables_kernel<uint8_t><<<blocks, 256, 0, s>>>((const uint8_t*)in_d, tables_d, Pz, Py, Px, (uint8_t*)vol_d, Y, X, C, total);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

extern "C" int bpx_scatter3d_regions(const float* pred_d, int n, int Pz, int Py, int Px, int C, const int* regions_d, float* out_d, int Z, int Y,
                                     int X, bpx_stream_t stream) {
  const char* fn = "bpx_scatter3d_regions";
  BPX_CHECK(pred_d && out_d && regions_d, "%s: null pointer", fn);
  BPX_CHECK(n >= 0 && Pz > 0 && Py > 0 && Px > 0 && C > 0 && Z > 0 && Y > 0 && X > 0, "%s: bad extents", fn);
  if (n == 0) return 0;
  dim3 grid((unsigned)std::min<int64_t>(cdiv64((int64_t)Pz * Py * Px * C, 256), 1024), (unsigned)n);
  scatter3d_regions_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(pred_d, Py, Px, C, (int64_t)Pz * Py * Px * C, regions_d, out_d, Y, X);
  BPX_LAUNCH_CHECK(fn);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// merge
// ------------------------------------------------------------------------------------------------
template <typename E> __device__ __forceinline__ float load_as_f32(const E* p);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_as_f32<uint8_t>(const uint8_t* p) { return (float)*p; }
template <> __device__ __forceinline__ float load_as_f32<__half>(const __half* p) { return __half2float(*p); }

template <typename E> __device__ __forceinline__ void store_from_f32(E* p, float v);
template <> __device__ __forceinline__ void store_from_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_from_f32<uint8_t>(uint8_t* p, float v) { *p = (uint8_t)(int)v; }  // astype: truncate
template <> __device__ __forceinline__ void store_from_f32<__half>(__half* p, float v) { *p = __float2half_rn(v); }

__device__ __forceinline__ void cover_range(const AxisG& g, int q, int& lo, int& hi) {
  // patches i with start(i) <= q < start(i)+patch lie in [lo,hi]; start(i) in [i*step-last, i*step]
  if (g.step <= 0) { lo = 0; hi = g.n - 1; return; }   // volume == patch along this axis with an overlap: the reference places
  //                                                      ceil(dim/step) patches and then shrinks the step to 0 - all start at 0
  int t = q - g.patch;
  lo = t < 0 ? 0 : t / g.step + 1;
  hi = (q + g.last) / g.step;
  if (hi > g.n - 1) hi = g.n - 1;
}

template <typename EI, typename EO>
__global__ void __launch_bounds__(256) merge3d_kernel(const EI* __restrict__ patches, int Pzf, int Pyf, int Pxf, int C, int pz, int py,
                                                      int px, AxisG gz, AxisG gy, AxisG gx, const float* __restrict__ wz,
                                                      const float* __restrict__ wy, const float* __restrict__ wx, int Y, int X,
                                                      int z_lo, int z_hi, int zrow_lo, int zrow_hi, float* acc, float* wacc,
                                                      int flags, EO* __restrict__ out) {
  const int64_t row = (int64_t)X * C;
  const int64_t total = (int64_t)(z_hi - z_lo) * Y * row;
  const int64_t pstride_y = (int64_t)Pxf * C, pstride_z = pstride_y * Pyf, pstride_c = pstride_z * Pzf;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / row;
    int xc = (int)(i - r * row);
    int x = xc / C, ch = xc - x * C;
    int y = (int)(r % Y);
    int z = (int)(r / Y) + z_lo;
    float num = 0.f, ws = 0.f;
    if (flags & 2) { num = acc[i]; ws = wacc[i / C]; }
    int zl, zh, yl, yh, xl, xh;
    cover_range(gz, z, zl, zh);
    cover_range(gy, y, yl, yh);
    cover_range(gx, x, xl, xh);
    if (zl < zrow_lo) zl = zrow_lo;
    if (zh > zrow_hi - 1) zh = zrow_hi - 1;
    for (int iz = zl; iz <= zh; ++iz) {
      int lz = z - gz.start(iz);
      if (lz < 0 || lz >= gz.patch) continue;
      float wzv = wz[lz];
      for (int iy = yl; iy <= yh; ++iy) {
        int ly = y - gy.start(iy);
        if (ly < 0 || ly >= gy.patch) continue;
        float wzy = __fmul_rn(wzv, wy[ly]);
        const EI* pbase = patches + ((int64_t)(iz - zrow_lo) * gy.n + iy) * gx.n * pstride_c + (int64_t)(lz + pz) * pstride_z +
                          (int64_t)(ly + py) * pstride_y + ch;
        for (int ix = xl; ix <= xh; ++ix) {
          int lx = x - gx.start(ix);
          if (lx < 0 || lx >= gx.patch) continue;
          float w = __fmul_rn(wzy, wx[lx]);
          float v = load_as_f32<EI>(pbase + (int64_t)ix * pstride_c + (int64_t)(lx + px) * C);
          num = __fadd_rn(num, __fmul_rn(v, w));
          ws = __fadd_rn(ws, w);
        }
      }
    }
    if (flags & 1) {
      acc[i] = num;
      if (ch == 0) wacc[i / C] = ws;
    } else {
      store_from_f32<EO>(out + i, __fdiv_rn(num, __fadd_rn(ws, 1e-18f)));
    }
  }
}


// Vectorised form (the production path).  A thread owns VEC consecutive elements of one output row in the flattened (x, c)
// index.  For a covering patch the same VEC elements are consecutive in the patch row as well (the patch row is the output
// row shifted by start(ix)), so a vector whose first and last element lie inside the patch is ONE (possibly misaligned) wide
// load; vectors straddling a patch edge go element by element.  The patch cover of the row (z and y ranges, the row weight
// fl(wz*wy), the row base pointers) is evaluated once per vector, not per element; indices inside a patch are 32-bit.  The
// accumulation order per element - patches in z-major order, num = fl(num + fl(v*w)), ws = fl(ws + w) - is the reference's.
constexpr int MERGE_WXMAX = 2048;
template <typename E> struct MergeVec;
template <> struct MergeVec<float> { static constexpr int VEC = 4; };
template <> struct MergeVec<__half> { static constexpr int VEC = 8; };
template <> struct MergeVec<uint8_t> { static constexpr int VEC = 8; };

__device__ __forceinline__ void cover_range_fast(const AxisG& g, const FastDiv& ds, int q, int& lo, int& hi) {
  if (g.step <= 0) { lo = 0; hi = g.n - 1; return; }   // see cover_range
  const int t = q - g.patch;
  lo = t < 0 ? 0 : (int)ds.div((uint32_t)t) + 1;
  hi = (int)ds.div((uint32_t)(q + g.last));
  if (hi > g.n - 1) hi = g.n - 1;
}

struct MergeDivs { FastDiv q, Y, C, sz, sy, sx; };

// One covering patch of a vector: its VEC values (zero where the element lies outside the patch), the row weight fl(wz*wy),
// the patch start along x and the per-element validity mask.  Up to 3 x 3 (y, x) patches of one z row are LOADED FIRST (nine
// independent wide loads in flight per thread - the kernel is a latency-bound gather otherwise) and then accumulated in the
// reference's order; geometries with more than three covering patches along y or x (overlap > 2/3) take the sequential path.
template <typename EI, int VEC> struct MergeSlot {
  EI v[VEC];
  float wzy;
  int sx;
  unsigned mask;
};

template <typename EI, typename EO>
__global__ void __launch_bounds__(256) merge3d_row_kernel(const EI* __restrict__ patches, int Pzf, int Pyf, int Pxf, int C, int pz, int py,
                                                          int px, AxisG gz, AxisG gy, AxisG gx, const float* __restrict__ wz,
                                                          const float* __restrict__ wy, const float* __restrict__ wx, int Y, int X,
                                                          int z_lo, int zrow_lo, int zrow_hi, float* acc, float* wacc, int flags,
                                                          EO* __restrict__ out, uint32_t total_vec, uint32_t t_base, MergeDivs dv) {
  constexpr int VEC = MergeVec<EI>::VEC;
  // The x taper lives in LDS.  Measured alternatives on 512 x 128^3 -> 512^3 (this form: 1.73 ms): reading the taper through the
  // vector L1 instead (four strided dword loads per patch: 2.73 ms - the texture path is the bottleneck then), and a
  // workgroup-per-output-row variant with the row arithmetic on the scalar unit (2.16 ms), and four workgroups per CU instead of three
  // (__launch_bounds__(256, 4), with the slot records slimmed to values + mask: 88 B/lane of scratch and 2.72 ms - the 141 VGPRs are the
  // nine loads in flight per thread, which is what the kernel lives on), and patches placed 64 ... 65,728 elements further apart than their
  // 8 MiB (a power of two: would the up to 27 simultaneous gathers meet on the same HBM channels?): 1.77 ms at every padding - they do not.
  __shared__ float swx[MERGE_WXMAX];                              // (the launcher checks gx.patch <= MERGE_WXMAX)
  for (int i = threadIdx.x; i < gx.patch; i += 256) swx[i] = wx[i];
  __syncthreads();
  const uint32_t tl = blockIdx.x * 256u + threadIdx.x;
  if (tl >= total_vec) return;
  uint32_t row, q, zr, yu;
  dv.q.divmod(tl, row, q);                                        // row = (z - z_lo') * Y + y inside this launch
  dv.Y.divmod(row, zr, yu);
  const int y = (int)yu;
  const int z = (int)zr + z_lo;
  const int e0 = (int)q * VEC;
  const int x0 = (int)dv.C.div((uint32_t)e0), ch0 = e0 - x0 * C;
  int xk[VEC];
  {
    int x = x0, ch = ch0;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      xk[k] = x;
      if (++ch == C) { ch = 0; ++x; }
    }
  }
  const int64_t grow = (int64_t)(t_base / dv.q.d) + row;           // row index inside the [z_hi-z_lo][Y][X][C] arrays of the CALL
  const int64_t i0 = grow * (int64_t)X * C + e0;                   // flat index of element 0
  const int64_t v0 = grow * (int64_t)X;                            // voxel index of x = 0 of this row
  float num[VEC], ws[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) { num[k] = 0.f; ws[k] = 0.f; }
  if (flags & 2) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) { num[k] = acc[i0 + k]; ws[k] = wacc[v0 + xk[k]]; }
  }
  int zl, zh, yl, yh, xl, xh, dummy;
  cover_range_fast(gz, dv.sz, z, zl, zh);
  cover_range_fast(gy, dv.sy, y, yl, yh);
  cover_range_fast(gx, dv.sx, xk[0], xl, dummy);
  cover_range_fast(gx, dv.sx, xk[VEC - 1], dummy, xh);
  if (zl < zrow_lo) zl = zrow_lo;
  if (zh > zrow_hi - 1) zh = zrow_hi - 1;
  const int pstride_y = Pxf * C, pstride_z = pstride_y * Pyf;
  const int64_t pstride_c = (int64_t)pstride_z * Pzf;
  const bool batched = (yh - yl) < 3 && (xh - xl) < 3;

  auto accumulate = [&](const EI* v, float wzy, int sx, unsigned mask) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      if (mask & (1u << k)) {
        const float w = __fmul_rn(wzy, swx[xk[k] - sx]);
        num[k] = __fadd_rn(num[k], __fmul_rn(load_as_f32<EI>(&v[k]), w));
        ws[k] = __fadd_rn(ws[k], w);
      }
    }
  };
  // loads the VEC values of patch (row pointer prow, column ix) that cover this vector; returns the validity mask (0 = none)
  auto fetch = [&](const EI* prow, int ix, EI* v, int& sx_out) -> unsigned {
    const int sx = gx.start(ix);
    sx_out = sx;
    const int l0 = xk[0] - sx, l1 = xk[VEC - 1] - sx;
    if (l1 < 0 || l0 >= gx.patch) return 0u;
    const EI* pp = prow + (int64_t)ix * pstride_c + ((px - sx) * C + e0);   // element k of the vector is pp[k] when it is inside
    if (l0 >= 0 && l1 < gx.patch) {
      __builtin_memcpy(v, pp, sizeof(EI) * VEC);                    // one wide (possibly misaligned) load
      return (1u << VEC) - 1u;
    }
    unsigned mask = 0u;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int lx = xk[k] - sx;
      v[k] = EI(0);
      if (lx >= 0 && lx < gx.patch) { v[k] = pp[k]; mask |= 1u << k; }
    }
    return mask;
  };

  for (int iz = zl; iz <= zh; ++iz) {
    const int lz = z - gz.start(iz);
    if (lz < 0 || lz >= gz.patch) continue;
    const float wzv = wz[lz];
    const EI* pz_base = patches + (int64_t)(iz - zrow_lo) * gy.n * gx.n * pstride_c + (lz + pz) * pstride_z;
    if (batched) {
      MergeSlot<EI, VEC> slot[9];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int iy = yl + j;
        const int ly = y - gy.start(iy);
        const bool oky = iy <= yh && ly >= 0 && ly < gy.patch;
        const float wzy = oky ? __fmul_rn(wzv, wy[ly]) : 0.f;
        const EI* prow = pz_base + (int64_t)iy * gx.n * pstride_c + (ly + py) * pstride_y;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          MergeSlot<EI, VEC>& sl = slot[j * 3 + i];
          sl.mask = 0u;
          sl.wzy = wzy;
          sl.sx = 0;
          if (oky && xl + i <= xh) sl.mask = fetch(prow, xl + i, sl.v, sl.sx);
        }
      }
#pragma unroll
      for (int b = 0; b < 9; ++b)
        if (slot[b].mask) accumulate(slot[b].v, slot[b].wzy, slot[b].sx, slot[b].mask);
    } else {
      for (int iy = yl; iy <= yh; ++iy) {
        const int ly = y - gy.start(iy);
        if (ly < 0 || ly >= gy.patch) continue;
        const float wzy = __fmul_rn(wzv, wy[ly]);
        const EI* prow = pz_base + (int64_t)iy * gx.n * pstride_c + (ly + py) * pstride_y;
        for (int ix = xl; ix <= xh; ++ix) {
          EI v[VEC];
          int sx;
          const unsigned mask = fetch(prow, ix, v, sx);
          if (mask) accumulate(v, wzy, sx, mask);
        }
      }
    }
  }
  if (flags & 1) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[i0 + k] = num[k];
    {
      int x = x0, ch = ch0;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if (ch == 0) wacc[v0 + x] = ws[k];
        if (++ch == C) { ch = 0; ++x; }
      }
    }
  } else {
    EO o[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) store_from_f32<EO>(&o[k], __fdiv_rn(num[k], __fadd_rn(ws[k], 1e-18f)));
    __builtin_memcpy(__builtin_assume_aligned(out + i0, sizeof(EO) * VEC >= 16 ? 16 : sizeof(EO) * VEC), o, sizeof(o));
  }
}

extern "C" int bpx_merge3d_blend(const void* patches_d, int dtype, int Pz, int Py, int Px, int C, int pad_z, int pad_y, int pad_x,
                                 const bpx_axis_grid* g, const float* wz_d, const float* wy_d, const float* wx_d, int Z, int Y, int X,
                                 int z_lo, int z_hi, int zrow_lo, int zrow_hi, float* acc_d, float* wacc_d, int flags,
                                 void* out_d, int out_dtype, bpx_stream_t stream) {
  BPX_CHECK(patches_d && g && wz_d && wy_d && wx_d, "bpx_merge3d_blend: null pointer");
  BPX_CHECK(0 <= z_lo && z_lo <= z_hi && z_hi <= Z, "bpx_merge3d_blend: bad z range [%d,%d) for Z=%d", z_lo, z_hi, Z);
  BPX_CHECK(0 <= zrow_lo && zrow_lo <= zrow_hi && zrow_hi <= g[0].n, "bpx_merge3d_blend: bad patch-row range");
  BPX_CHECK(g[0].patch == Pz - 2 * pad_z && g[1].patch == Py - 2 * pad_y && g[2].patch == Px - 2 * pad_x,
            "bpx_merge3d_blend: grid patch extent must equal the padding-stripped patch");
  BPX_CHECK(g[0].limit == Z && g[1].limit == Y && g[2].limit == X, "bpx_merge3d_blend: grid limit must equal the volume extent");
  BPX_CHECK(!(flags & 3) || (acc_d && wacc_d), "bpx_merge3d_blend: acc/wacc required for partial/seeded blends");
  BPX_CHECK((flags & 1) || out_d, "bpx_merge3d_blend: out_d is null");
  int64_t total = (int64_t)(z_hi - z_lo) * Y * X * C;
  if (total == 0) return 0;
  int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 256 * 16);
  hipStream_t s = (hipStream_t)stream;
  AxisG gz = to_axis(g[0]), gy = to_axis(g[1]), gx = to_axis(g[2]);
  {
    // vector path: rows of whole vectors, 16-byte aligned output, patch strides that fit 32 bits
    const int vec = dtype == BPX_F32 ? 4 : 8;
    const size_t oes = dtype_size(out_dtype);
    const bool row_ok = ((int64_t)X * C) % vec == 0 && (int64_t)Px * Py * Pz * C < (1ll << 31) && g[2].patch <= MERGE_WXMAX && Z < (1 << 29) &&
                        Y < (1 << 29) && X < (1 << 29);
    const bool al_ok = (flags & 1) ? (((uintptr_t)acc_d & 15) == 0) : (((uintptr_t)out_d & 15) == 0);
    const int combo = dtype == BPX_F32 && out_dtype == BPX_F32 ? 0 : dtype == BPX_U8 && out_dtype == BPX_U8 ? 1 : dtype == BPX_F16 && out_dtype == BPX_F16 ? 2
                      : dtype == BPX_F16 && out_dtype == BPX_F32 ? 3 : dtype == BPX_U8 && out_dtype == BPX_F32 ? 4 : -1;
    const int qpr = row_ok ? (int)(((int64_t)X * C) / vec) : 0;
    if (row_ok && al_ok && combo >= 0 && !g_tiling_scalar && (int64_t)Y * qpr < (1ll << 30)) {
      (void)oes;
      // 31-bit thread indices per launch (FastDiv): a taller slab is cut into launches of whole z slices
      const int64_t vec_per_slice = (int64_t)Y * qpr;
      const int slices_per_launch = (int)std::max<int64_t>(1, ((1ll << 30) - 1) / vec_per_slice);
      MergeDivs dv{make_fastdiv((uint32_t)qpr), make_fastdiv((uint32_t)Y), make_fastdiv((uint32_t)C), make_fastdiv((uint32_t)g[0].step),
                   make_fastdiv((uint32_t)g[1].step), make_fastdiv((uint32_t)g[2].step)};
      for (int za = z_lo; za < z_hi; za += slices_per_launch) {
        const int zn = std::min(slices_per_launch, z_hi - za);
        const uint32_t total_vec = (uint32_t)(zn * vec_per_slice);
        const unsigned nb = (unsigned)cdiv64(total_vec, 256);
        // element / voxel offsets of this launch inside the call's arrays are carried by t_base (in vectors of the call)
        const int64_t tb64 = (int64_t)(za - z_lo) * vec_per_slice;
        BPX_CHECK(tb64 < (1ll << 32), "bpx_merge3d_blend: slab too large");
        const uint32_t tb = (uint32_t)tb64;
#define MERGE_ROW(EI, EO)                                                                                                                   \
  merge3d_row_kernel<EI, EO><<<nb, 256, 0, s>>>((const EI*)patches_d, Pz, Py, Px, C, pad_z, pad_y, pad_x, gz, gy, gx, wz_d, wy_d, wx_d, Y, X, za, \
                                                zrow_lo, zrow_hi, acc_d, wacc_d, flags, (EO*)out_d, total_vec, tb, dv)
        if (combo == 0) MERGE_ROW(float, float);
        else if (combo == 1) MERGE_ROW(uint8_t, uint8_t);
        else if (combo == 2) MERGE_ROW(__half, __half);
        else if (combo == 3) MERGE_ROW(__half, float);
        else MERGE_ROW(uint8_t, float);
#undef MERGE_ROW
      }
      BPX_LAUNCH_CHECK("bpx_merge3d_blend");
      return 0;
    }
  }
#define MERGE_LAUNCH(EI, EO)                                                                                                   \
  merge3d_kernel<EI, EO><<<blocks, 256, 0, s>>>((const EI*)patches_d, Pz, Py, Px, C, pad_z, pad_y, pad_x, gz, gy, gx, wz_d, wy_d, \
                                                wx_d, Y, X, z_lo, z_hi, zrow_lo, zrow_hi, acc_d, wacc_d, flags, (EO*)out_d)
  if (dtype == BPX_F32 && out_dtype == BPX_F32) MERGE_LAUNCH(float, float);
  else if (dtype == BPX_U8 && out_dtype == BPX_U8) MERGE_LAUNCH(uint8_t, uint8_t);
  else if (dtype == BPX_F16 && out_dtype == BPX_F16) MERGE_LAUNCH(__half, __half);
  else if (dtype == BPX_F16 && out_dtype == BPX_F32) MERGE_LAUNCH(__half, float);
  else if (dtype == BPX_U8 && out_dtype == BPX_F32) MERGE_LAUNCH(uint8_t, float);
  else BPX_FAIL("bpx_merge3d_blend: unsupported dtype combination in=%d out=%d", dtype, out_dtype);
#undef MERGE_LAUNCH
  BPX_LAUNCH_CHECK("bpx_merge3d_blend");
  return 0;
}
