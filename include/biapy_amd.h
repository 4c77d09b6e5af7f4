/*
 * biapy_amd.h - C-ABI of libbiapy_amd.so: the MI355X (gfx950) kernels behind BiaPy's 3D patch
 * U-Net hot path.  Plain pointers and sizes only; every pointer named *_d is a DEVICE pointer,
 * the caller owns all memory, nothing is retained across calls, all launches go to `stream`
 * (a hipStream_t; NULL = the default stream).  Every entry returns 0 on success, non-zero on
 * error with bpx_last_error() (thread-local) describing it.
 *
 * The reference (BiaPyX/BiaPy v3.7.0) is 100 % Python on PyTorch/NumPy, so there is no native FFI
 * to bind to; each entry cites the reference operator/function it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes stubs a BiaPy maintainer would add.
 *
 * Tensor layout: activations are NDHWC ("(Z,Y,X,C) patch layout"), i.e. [N][D][H][W][ld] with the
 * channel stride `ld` >= C so that a tensor can be a channel slice of a wider buffer (this is how
 * torch.cat([up, skip], 1) - biapy/models/blocks.py:1653 - is eliminated).
 */
#ifndef BIAPY_AMD_H
#define BIAPY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bpx_stream_t; /* hipStream_t */

enum bpx_dtype { BPX_F32 = 0, BPX_BF16 = 1, BPX_F16 = 2, BPX_U8 = 3,
                 /* BPX_MIX16: the mixed 16-bit TRAINING mode, accepted by the backward entry points only.  Activation tensors (the forward
                  * pass's raw outputs: the `t` / `x` operands below) are fp16 - written by the BPX_F16 forward kernels, whose logits
                  * agree with the fp32 reference to Dice delta < 1e-4 - while every GRADIENT tensor (dy, g, dx, addend) and the MFMA
                  * operands of the backward kernels are bf16 (fp32 exponent range: no loss scaling).  Each backward entry says which of
                  * its operands are activations. */
                 BPX_MIX16 = 4 };
/* Block activations (biapy/models/blocks.py:1973-1998, get_activation): codes 0-3 since round 1; round 4 adds leaky_relu (slope 0.01, nn.LeakyReLU's
 * default), gelu (nn.GELU(): the exact erf form), tanh, sigmoid and softplus (beta 1, threshold 20).  "softmax" as a block activation needs a
 * reduction over channels and is not a per-element prologue: not offered.  ELU (the reference default) has compile-time instances of every
 * kernel; the other codes take the run-time-switch instances. */
enum bpx_act { BPX_ACT_NONE = 0, BPX_ACT_ELU = 1, BPX_ACT_RELU = 2, BPX_ACT_SILU = 3, BPX_ACT_LEAKY_RELU = 4, BPX_ACT_GELU = 5, BPX_ACT_TANH = 6,
               BPX_ACT_SIGMOID = 7, BPX_ACT_SOFTPLUS = 8, BPX_ACT_LAST = 8 };
enum bpx_pad_mode { BPX_PAD_REFLECT = 0, BPX_PAD_ZEROS = 1 };

int bpx_version(void);
const char* bpx_last_error(void);
/* MFMA / LDS-transpose lane-layout self test (prints nothing; writes 64*16 floats). Used by tests. */
int bpx_selftest_layouts(float* out_d /* 1024 floats */, bpx_stream_t stream);
/* Test / A-B hook of the weight-gradient kernels (default 1).  bit 0: ds_read_b64_tr_b16 operands (0 = scalar LDS gathers);
 * bits 1-2 choose the bf16 3x3x3 schedule: 2 = never a shift-dy kernel, 4 = always; bits 3-5: input-channel chunks per
 * workgroup of the windowed shift-dy kernel (1, 2, 3 forced; 4 = the plain shift-dy kernel); bit 6: transposed-conv wgrad
 * with the former 2048-workgroup target; bit 7 + bits 8..: k = 1 launches target (value)% of 1024 workgroups; without bit 7,
 * bits 8..: windowed kernel workgroups in percent of the co-resident capacity. */
int bpx_debug_set_wgrad_tr(int use_tr);
int bpx_debug_set_conv_stamps(void* stamps_d); /* profiling hook: [workgroup][16] int64 cycle stamps of the plain conv kernel, NULL = off */
int bpx_debug_set_conv_ws(int on);     /* test / A-B hook of the bf16 3x3x3 conv schedule: 0 = automatic, 4 = always the double-buffered kernel, 5 = always the lean persistent one */
int bpx_debug_set_conv_occ(int wg_per_cu); /* test / A-B hook: persistent workgroups per CU of the lean bf16 conv kernel (0 = built-in table) */
int bpx_debug_set_conv_zm(int mode);       /* test / A-B hook of the z-marching forward kernel of the 16-output-channel layers (conv3d_zmarch.hip; same bits as the lean kernel): -1 = environment BPX_CONV_ZM (default 1), 0 = off, 1 = where a run has several z-steps, 2 = wherever the shape admits it; + 4 = without the role-split form (conv3_zs_kernel) of the one-chunk layers; bits 8.. (tests) = cap on the number of workgroups */
int bpx_debug_conv_zm_launches(void);      /* tests: how many launches took the z-marching kernel so far */
int bpx_debug_conv_zm_occupancy(int nch);  /* tests: resident workgroups per CU of the z-marching kernel (1 or 3 input chunks; 0 = its role-split form of 512 threads) as the runtime computes it; -1 = query failed */
int bpx_debug_set_c1_persist(int wgs); /* test / A-B hook: persistent workgroups of bpx_conv3d_c1_fwd (default 2048; 0 = one workgroup per tile); + 2^30 = the pointer-addressed instance instead of the buffer-addressed one (same bits) */
int bpx_debug_set_convt_k1(int on);  /* test / A-B hook of the transposed-conv forward with one K step into a chunk-planar buffer: -1 = environment BPX_CONVT_K1 (default 1), 1 = the branch-free buffer-addressed kernel where it applies (W % 16 == 0, operands within 32-bit offsets), 0 = the general kernel; same bits */
int bpx_debug_set_pw_stream(int on); /* test / A-B hook: 1 (default) = the streaming kernel for bpx_conv1x1_fwd_split with the IN-backward affine at the large levels, 0 = the tile kernel */
int bpx_debug_set_wgrad_k1(int on); /* test / A-B hook of the streaming weight-gradient kernels at the large levels: 1 (default) = both on, 0 = the tile kernels, 3 = streaming k = 1 (raw-input shortcut) only, 5 = streaming transposed-conv only, 7 = both and the 32 -> 32 transposed-conv instance too (measured slower than its tile kernel) */
int bpx_debug_set_wgrad_cap(int percent); /* test / A-B hook: size cap of a conv layer's weight-gradient partial slabs in percent of the default (~26 / 64 MB) */
int bpx_debug_set_tile_order(int bits); /* test / A-B hook: bit 0 = XCD-contiguous y-strip tile walk of the windowed shift-dy wgrad kernel (default 1; 0 = tile = group + k * groups as until round 3) */
int bpx_debug_set_tiling_scalar(int on); /* test / A-B hook: 1 = crop / merge through the element-per-thread kernels instead of the 16-byte row kernels */

/* ------------------------------------------------------------------------------------------------
 * Tiling.  Patch placement along one axis, exactly the reference's integer rule
 * (biapy/data/data_3D_manipulation.py:541-547, :596-598 / :789-795, :826-835):
 *   start(i) = i*step - ((i*step + patch < limit) ? 0 : last)
 * ---------------------------------------------------------------------------------------------- */
typedef struct bpx_axis_grid {
  int32_t n, step, last, patch, limit;
} bpx_axis_grid;

/* crop_3D_data_with_overlap's copy loop + np.pad (data_3D_manipulation.py:505-515, :591-623) as an
 * on-device gather: out[c] = padded(vol)[z0:z0+Pz, y0:y0+Py, x0:x0+Px] for the patches
 * c in [c_begin, c_begin+c_count) of the z-major patch order.  elem_size = bytes per element
 * (1, 2 or 4); data is moved bit-exactly.  grid[] is the CROP grid (padded coordinates). */
int bpx_crop3d_gather(const void* vol_d, int elem_size, int Z, int Y, int X, int C,
                      int pad_z, int pad_y, int pad_x, int pad_mode,
                      const bpx_axis_grid* grid_zyx /* host, 3 entries */,
                      int64_t c_begin, int64_t c_count, void* out_d, bpx_stream_t stream);

/* merge_3D_data_with_overlap (data_3D_manipulation.py:754-856) as a deterministic gather: every
 * output voxel sums fl32(patch*w) over the patches covering it in the reference's patch order
 * (z-major), w = fl32(fl32(wz*wy)*wx), then divides by fl32(sum(w)+1e-18f) and casts to out_dtype
 * (truncating for u8, as NumPy's astype).  grid[] is the MERGE grid (original coordinates,
 * padding-stripped patch).  Patches are [n][Pz][Py][Px][C] including the padding that is skipped.
 *
 * Sharding (one Z-slab of the volume per GPU): only output slices z in [z_lo, z_hi) are produced,
 * from the patches whose z-row index is in [zrow_lo, zrow_hi) - `patches_d` holds just those rows.
 * acc_d / wacc_d (float, [z_hi-z_lo][Y][X][C] / [..][1], may be NULL) seed the sums with a
 * neighbour's partial sums; with write_partial != 0 the un-normalised sums are written back to
 * acc_d / wacc_d instead of the normalised volume (used for the slab-boundary exchange). */
int bpx_merge3d_blend(const void* patches_d, int dtype, int Pz, int Py, int Px, int C,
                      int pad_z, int pad_y, int pad_x,
                      const bpx_axis_grid* grid_zyx /* host, 3 entries */,
                      const float* wz_d, const float* wy_d, const float* wx_d,
                      int Z, int Y, int X, int z_lo, int z_hi, int zrow_lo, int zrow_hi,
                      float* acc_d, float* wacc_d, int write_partial,
                      void* out_d, int out_dtype, bpx_stream_t stream);

/* By-chunks tiler (chunked_test_pair_data_generator.py:440-565; base_workflow.py:2603-2610).
 * gather: out[b,z,y,x,c] = vol[tz[z], ty[y], tx[x], c] with the three int32 source-index tables of patch b stored back to
 * back in tables_d[b*(Pz+Py+Px) ...] (they encode the clipped read region + np.pad "reflect"); data moves bit-exactly.
 * scatter: the prediction of patch b without its padding goes to its place in the output volume; regions_d[b*9 ...] =
 * {first kept voxel of the patch z,y,x ; destination z,y,x ; extent z,y,x}.  Regions of different patches are disjoint. */
int bpx_gather3d_tables(const void* vol_d, int elem_size, int Z, int Y, int X, int C, const int* tables_d, int n, int Pz, int Py, int Px,
                        void* out_d, bpx_stream_t stream);
/* Inverse of bpx_gather3d_tables: vol[tz[z], ty[y], tx[x], c] = in[b,z,y,x,c] for the non-negative table entries (negative entries
 * of either call mean "outside the volume": the gather returns 0 there, the scatter skips them).  Together they are the
 * space-to-batch / batch-to-space pair that turns a dilated 3x3x3 convolution (ASPP, heads.py:77-104) into ordinary ones. */
int bpx_scatter3d_tables(const void* in_d, int elem_size, const int* tables_d, int n, int Pz, int Py, int Px, void* vol_d, int Z, int Y,
                         int X, int C, bpx_stream_t stream);
int bpx_scatter3d_regions(const float* pred_d, int n, int Pz, int Py, int Px, int C, const int* regions_d, float* out_d, int Z, int Y,
                          int X, bpx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Network kernels.  T = BPX_BF16 (bf16 storage, fp32 accumulate, v_mfma_f32_16x16x32_bf16) or
 * BPX_F32 (fp32 storage, exact-fp32 v_mfma_f32_16x16x4_f32; the parity/debug mode).
 * ---------------------------------------------------------------------------------------------- */

/* Per-(n,c) normalisation record produced by bpx_norm_finalize and consumed by conv prologues. */
typedef struct bpx_norm_rec {
  float mean, rstd, scale, shift; /* y = act(scale*x + shift), scale = gamma*rstd, shift = beta - mean*scale */
} bpx_norm_rec;

typedef struct bpx_tensor {
  void* ptr;   /* device; element (n,z,y,x,c) at ((((n*D+z)*H+y)*W+x)*ld + c) */
  int32_t ld;  /* channel stride in elements */
  int32_t C;   /* channels used */
  int64_t cs;  /* 0: the C channels of a voxel are contiguous (above).  != 0: CHUNK-PLANAR - the 16-channel chunk c/16 of every voxel
                * lives in its own plane, element (voxel v, channel c) at v*ld + (c/16)*cs + c%16 (cs in elements, a multiple of 8;
                * ptr 16-byte aligned at channel 0 of a chunk).  The engine keeps the torch.cat buffers of the decoder this way (planes
                * written by different producers are whole cache lines).  Accepted only where an entry point says so. */
} bpx_tensor;

/* Weight packing: PyTorch fp32 weights -> MFMA operand order (DESIGN.md "packed weights").
 *   mode 0 BPX_PK_K3     Conv3d (Cout,Cin,3,3,3)       -> [Cin/16][q<QPAD][Cout][KPL]      (bpx_conv3d_fwd)
 *   mode 1 BPX_PK_K3_T   same weight, dgrad operator   -> [Cout/16][q<QPAD][Cin][KPL], taps mirrored (bpx_conv3d_dgrad)
 *   mode 2 BPX_PK_K1     Conv3d (Cout,Cin,1,1,1)       -> [Cin/16][4][Cout][KPL]           (fused shortcut of bpx_conv3d_fwd)
 *   mode 3 BPX_PK_DENSE  Conv3d k=1                    -> [ceil4(Cin/KPL)][Cout][KPL]      (bpx_conv1x1_fwd)
 *   mode 4 BPX_PK_DENSE_T same weight, dgrad operator  -> [ceil4(Cout/KPL)][Cin][KPL]      (bpx_conv1x1_fwd as dgrad)
 *   mode 5 BPX_PK_CT     ConvTranspose3d (Cin,Cout,2,2,2) -> [ceil4(Cin/KPL)][8*Cout][KPL] (bpx_convT3d_k2s2_fwd)
 *   mode 6 BPX_PK_CT_T   same weight, dgrad operator   -> [8*Cout/KPL][Cin][KPL]           (bpx_convT3d_k2s2_dgrad)
 *   mode 7 / 8 BPX_PK_CT4 / _T   the same for the anisotropic kernel (1,2,2) (Z_DOWN = 1): 4 sub-positions instead of 8
 * KPL = 8 (bf16) / 4 (f32) elements per 16-byte lane operand; QPAD = 56 (bf16) / 108 (f32).
 * dtype BPX_MIX16: the forward operators (modes 0, 2, 3, 5, 7) are packed as fp16, the transposed / backward ones (1, 4, 6, 8) as bf16. */
enum bpx_pack_mode { BPX_PK_K3 = 0, BPX_PK_K3_T = 1, BPX_PK_K1 = 2, BPX_PK_DENSE = 3, BPX_PK_DENSE_T = 4, BPX_PK_CT = 5, BPX_PK_CT_T = 6,
                     BPX_PK_CT4 = 7, BPX_PK_CT4_T = 8 };
int64_t bpx_packed_weight_elems(int mode, int Cin, int Cout, int dtype);
int bpx_pack_weight(int mode, const float* w_d, int Cin, int Cout, int dtype, void* packed_d, bpx_stream_t stream);
/* The same for up to 64 weights in ONE launch (a training step re-packs ~60 small tensors after every optimizer step;
 * one launch instead of 60).  `jobs` is a HOST array; it is copied into the kernel arguments, nothing is retained. */
typedef struct bpx_pack_job {
  const float* w_d;   /* fp32 weights, PyTorch layout, device */
  void* packed_d;     /* destination, device, bpx_packed_weight_elems(mode, Cin, Cout, dtype) elements */
  int32_t mode, Cin, Cout, reserved;
} bpx_pack_job;
int bpx_pack_weights_batched(int dtype, int count, const bpx_pack_job* jobs, bpx_stream_t stream);

/* torch.optim.Adam / AdamW step (capturable form: `step` is a device float per tensor, lr may be a device scalar) over `count` fp32 tensors in
 * ceil(count / 64) + 1 launches of 4096-element blocks; the arithmetic, its order and its TYPES (double hyper-parameters, double x float products rounded to float at the assignment) are those of torch's fused kernel (train_engine.py:173-177
 * `optimizer.step()` of the reference loop).  `tensors` is a HOST array copied into the kernel arguments.  lr_d (device) overrides lr when not
 * NULL; decoupled = 1: AdamW (p -= lr * wd * p), 0: Adam (g += wd * p).  amsgrad / maximize are not supported (the caller keeps torch's step). */
typedef struct bpx_adam_tensor {
  float* p; const float* g; float* m; float* v; /* parameter, gradient, exp_avg, exp_avg_sq: `numel` contiguous floats each, device */
  float* step;                                  /* device scalar, incremented by one */
  int64_t numel;
} bpx_adam_tensor;
int bpx_adam_step(int count, const bpx_adam_tensor* tensors, const float* lr_d, double lr, double beta1, double beta2, double eps,
                  double weight_decay, int decoupled, bpx_stream_t stream);

/* Conv3d k=3 "same" + bias (biapy/models/blocks.py:154-157), implicit GEMM on MFMA with an
 * LDS-staged input halo.  Fusions:
 *   prologue : x <- act(scale*x+shift) with in_norm_d[n*Cin+c]  (the InstanceNorm+ELU that
 *              precedes this conv, blocks.py:1308-1314 / :158-161); NULL = raw input.
 *   shortcut : + Conv3d k=1 of a second raw tensor `sc` (blocks.py:1372, :1458) when sc.ptr != NULL;
 *              sc.C == 1 is handled as a rank-1 update.  w_sc packed with k=1.
 *   epilogue : per-(n,c) sum / sum-of-squares partials of the fp32 result into stats_part_d
 *              ([n][tiles][2][Cout] floats) for the next InstanceNorm (NULL = skip).
 * x.C must be 1 or a multiple of 16; Cout a multiple of 16. */
int bpx_conv3d_fwd(int dtype, int N, int D, int H, int W, bpx_tensor x, const bpx_norm_rec* in_norm_d, int act,
                   const void* w_packed_d, const float* bias_d, bpx_tensor sc, const void* w_sc_packed_d,
                   const float* bias_sc_d, bpx_tensor y, float* stats_part_d, bpx_stream_t stream);
int bpx_conv3d_stats_tiles(int dtype, int N, int D, int H, int W, int Cout); /* partial-sum slots per sample the call above writes */
/* The same convolution with the MaxPool3d (pool_sz,2,2) that follows an encoder block (resunet.py:256-257) fused into the
 * epilogue: `pooled` (extents D/pool_sz, H/2, W/2) and its statistics partials ([n][tiles][2][Cout], same tile count) are
 * written from registers, so the output slice is not read again.  Only where bpx_conv3d_fwd_pool_supported() returns 1
 * (the lean 16-bit kernel of the >= 32^3 levels); elsewhere use bpx_conv3d_fwd + bpx_maxpool3d_fwd. */
int bpx_conv3d_fwd_pool(int dtype, int N, int D, int H, int W, bpx_tensor x, const bpx_norm_rec* in_norm_d, int act,
                        const void* w_packed_d, const float* bias_d, bpx_tensor sc, const void* w_sc_packed_d,
                        const float* bias_sc_d, bpx_tensor y, float* stats_part_d, int pool_sz, bpx_tensor pooled,
                        float* pool_stats_part_d, bpx_stream_t stream);
int bpx_conv3d_fwd_pool_supported(int dtype, int N, int D, int H, int W, int x_ld, int y_ld, int Cout);
/* Conv3d k = 3 from x.C to 16 * s^3 channels followed by a 3-D pixel shuffle by s, in one pass (the up-scaling stage of the RCAN
 * super-resolution network, cfg 5).  biapy/models/rcan.py:317-319 builds `conv(filters, filters * scale**2) + nn.PixelShuffle(scale)`, which is
 * 2-D only (on 5-D tensors it raises); the 3-D form is DEFINED here as its natural extension:
 *   out[n, c, s z + a, s y + b, s x + e] = conv[n, c s^3 + (a s + b) s + e, z, y, x]        (a, b, e in [0, s))
 * The kernel takes the conv's output channels in the order [sub-position (a, b, e)][c] (w_packed_d = BPX_PK_K3 of the re-ordered weight, bias_d
 * re-ordered alike) and stores block (a, b, e) of voxel (z, y, x) to voxel (s z + a, s y + b, s x + e) of y = (N, sD, sH, sW, 16): the
 * 16 s^3-channel tensor never exists in memory.  dtype BF16 / F16; volumes the lean kernel takes (D*H*W >= 32^3, W > 8); s = 2, 3, 4. */
int bpx_conv3d_fwd_shuffle(int dtype, int N, int D, int H, int W, bpx_tensor x, const bpx_norm_rec* in_norm_d, int act, const void* w_packed_d,
                           const float* bias_d, int s, bpx_tensor y, bpx_stream_t stream);

/* dgrad of the conv above w.r.t. its (normalised+activated) input, fused with the backward of that
 * activation:  g = convT(dy, W) * act'(scale*t+shift),  t = the conv's raw input tensor.
 * Writes g and the per-(n,c) partials of sum(g) and sum(g*xhat) (xhat = (t-mean)*rstd) that both
 * InstanceNorm's input gradient and its dgamma/dbeta need.  t_norm_d == NULL: plain dgrad. */
int bpx_conv3d_dgrad(int dtype, int N, int D, int H, int W, bpx_tensor dy, const void* w_packed_T_d,
                     bpx_tensor t, const bpx_norm_rec* t_norm_d, int act, bpx_tensor g,
                     float* red_part_d, bpx_stream_t stream);

/* wgrad: dW[co][ci][tap] = sum_v act(norm(x))[v+tap][ci] * dy[v][co] written (overwritten) in the PyTorch
 * layout (Cout,Cin,k,k,k); db[co] += sum_v dy[v][co] (db_d must be zeroed by the caller, may be NULL).
 * k = 3 or 1.  Deterministic: per-workgroup partial sums of dW AND of the bias column sums go to the caller-provided workspace
 * and are summed in a fixed order (no atomics; db is updated by the reduction, i.e. at the flush in the deferred mode).
 * ws_bytes >= bpx_conv3d_wgrad_workspace(...).  x may be chunk-planar (bpx_tensor.cs). */
int64_t bpx_conv3d_wgrad_workspace(int N, int D, int H, int W, int Cin, int Cout, int k);
int bpx_conv3d_wgrad(int dtype, int N, int D, int H, int W, bpx_tensor x, const bpx_norm_rec* in_norm_d, int act,
                     bpx_tensor dy, int k, float* dw_d, float* db_d, void* ws_d, int64_t ws_bytes, bpx_stream_t stream);
/* The same with a second bias-gradient destination: db2_d[co] += the same column sums (needs db_d).  A residual block adds the biases of its
 * second convolution and of its 1x1x1 shortcut to the same tensor (biapy/models/blocks.py:1456-1459), so the two bias gradients are equal:
 * the reduction writes both instead of the caller copying one onto the other after the flush. */
int bpx_conv3d_wgrad_db2(int dtype, int N, int D, int H, int W, bpx_tensor x, const bpx_norm_rec* in_norm_d, int act,
                         bpx_tensor dy, int k, float* dw_d, float* db_d, float* db2_d, void* ws_d, int64_t ws_bytes, bpx_stream_t stream);

/* Backward of ONE 3x3x3 convolution in one pass over its operands: bpx_conv3d_dgrad (g, its sum(g) / sum(g*xhat) partials) AND
 * bpx_conv3d_wgrad_db2 (dW, db, db2) of the same (dy, t) pair - the two Conv3d gradients autograd computes for blocks.py:154-157 under
 * train_engine.py:173.  Both kernels stage the same haloed dy tile and the same raw input tile t (t is the conv's raw input; the conv saw
 * act(scale*t+shift)); fused, dy and t are read once (16 -> 16 channels: 3 tensor passes instead of 5; dy 16 -> t 48: 7 instead of 11).
 * Results are those of the two separate calls: g bit for bit that of bpx_conv3d_dgrad; its partials, dW and db fixed-order sums
 * of per-workgroup partials (bit-reproducible; the grouping of the voxels differs from bpx_conv3d_wgrad's, so the fp32 sums differ in the last bits).
 * Supported (bpx_conv3d_bwd_fused_supported): dtype BF16 or MIX16 (t fp16); (dy.C, t.C) = (16, 16 | 48) - the level-0 layers of cfg 2 - or
 * (32, 16 | 32) - level-1 layers (32 -> 96 measured no faster than the two kernels and is not taken); W > 8, >= 32^3 voxels per sample, t_norm_d given.  red_part_d: [N][bpx_conv3d_bwd_fused_stats_tiles(N, D, H, W, t.C, dy.C)][2][t.C] floats - one row per (sample, persistent workgroup column), every row written (zeros where a workgroup had no tile of the sample); (dy 16, t 48): one row per 4x4x16 tile.  ws_bytes >= bpx_conv3d_bwd_fused_workspace(...); inside a
 * bpx_wgrad_defer_begin / _flush window the reduction is queued like bpx_conv3d_wgrad's (own workspace per call).  t may be chunk-planar. */
int bpx_conv3d_bwd_fused_supported(int dtype, int N, int D, int H, int W, int Ct, int Cdy);
int bpx_conv3d_bwd_fused_stats_tiles(int N, int D, int H, int W, int Ct, int Cdy);
int64_t bpx_conv3d_bwd_fused_workspace(int N, int D, int H, int W, int Ct, int Cdy);
int bpx_conv3d_bwd_fused(int dtype, int N, int D, int H, int W, bpx_tensor dy, const void* w_packed_T_d, bpx_tensor t,
                         const bpx_norm_rec* t_norm_d, int act, bpx_tensor g, float* red_part_d, float* dw_d, float* db_d, float* db2_d,
                         void* ws_d, int64_t ws_bytes, bpx_stream_t stream);
int bpx_debug_set_bwd_fused(int bits); /* test / A-B hook: bit 0 clear = bpx_conv3d_bwd_fused_supported answers 0 everywhere (the engine then takes the two separate kernels); bit 1 set = only the dy.C == 16 instances */
int bpx_debug_set_bwd_rs(int mask);     /* test / A-B hook: bit 0 = the (dy 16, t 48) shape of bpx_conv3d_bwd_fused runs the role-split kernel (8 waves per CU: 4 dgrad + 4 wgrad, two tiles in flight), bit 1 = the (dy 16, t 16) shape does; call before _stats_tiles / _workspace */
int bpx_debug_set_conv_kg(int on);      /* test / A-B hook: the two-K-group form of the small-tile conv kernel (<= 16^3 layers with >= 4 input chunks): 1 on, 0 off, -1 = environment BPX_CONV_KG (default on) */

/* Deferred reduction of the weight-gradient partials.  Between bpx_wgrad_defer_begin() and bpx_wgrad_defer_flush() (same
 * host thread) bpx_conv3d_wgrad and the bf16 bpx_convT3d_k2s2_wgrad write only their partial slabs and queue the reduction;
 * the flush finishes up to 32 of them per launch (dw_d is written then).  Every deferred call needs its OWN workspace, alive
 * and untouched until the flush has run on the stream; dw_d AND db_d are complete only after the flush (a caller that copies a
 * bias gradient elsewhere does so after it).  bpx_conv3d_c1_wgrad and bpx_conv1x1_c1_wgrad queue their reductions the same way. */
int bpx_wgrad_defer_begin(void);
int bpx_wgrad_defer_flush(bpx_stream_t stream);

/* Per-(n,c) coefficients of InstanceNorm's input gradient, dx = a*g + b*t + c0 (see bpx_norm_bwd_finalize). */
typedef struct bpx_nbwd_coef { float a, b, c0, pad; } bpx_nbwd_coef;

/* Conv3d k=1 as a GEMM over voxels: y = x*W + bias [+ a*g + b*t + c0] [+ addend].  Used for the dgrad of
 * the residual shortcut (blocks.py:1372) fused with the InstanceNorm-backward affine of the main path, and
 * for stand-alone 1x1x1 convolutions.  `voxels` = voxels per sample; g/t/coef_d and addend may be null. */
int bpx_conv1x1_fwd(int dtype, int N, int64_t voxels, bpx_tensor x, const void* w_packed_d, const float* bias_d,
                    bpx_tensor g, bpx_tensor t, const bpx_nbwd_coef* coef_d, bpx_tensor addend, bpx_tensor y,
                    bpx_stream_t stream);
/* Same, with the output columns split over two dense tensors: [0, y_lo.C) -> y_lo, the rest -> y_hi (both counts multiples
 * of 4).  The gradient of torch.cat([up, bridge], 1) (blocks.py:1653) leaves as its two parts, each consumed by a kernel
 * that would otherwise read a channel slice with a 3/2x or 3x line over-fetch. */
int bpx_conv1x1_fwd_split(int dtype, int N, int64_t voxels, bpx_tensor x, const void* w_packed_d, const float* bias_d,
                          bpx_tensor g, bpx_tensor t, const bpx_nbwd_coef* coef_d, bpx_tensor addend, bpx_tensor y_lo,
                          bpx_tensor y_hi, bpx_stream_t stream);
/* The same with the residual block's SHORTCUT WEIGHT GRADIENT riding along (round 6): dw_d[co][ci] (Cout = x.C, Cin = t.C, 1, 1, 1) = sum_v t[v][ci] * x[v][co]
 * - the k = 1 bpx_conv3d_wgrad of the raw block input t against dOut = x (biapy/models/blocks.py ResConvBlock `shortcut(x)`), whose two operands this
 * kernel streams anyway.  Only where the streaming kernel applies: _workspace answers the bytes of ws_d, or 0 = call bpx_conv1x1_fwd_split and
 * bpx_conv3d_wgrad instead (BPX_PWS_WG=0 forces that).  Between bpx_wgrad_defer_begin and _flush dw_d is complete at the flush and ws_d must stay untouched until then. */
int64_t bpx_conv1x1_fwd_split_wgrad_workspace(int dtype, int N, int64_t voxels, int K);
int bpx_conv1x1_fwd_split_wgrad(int dtype, int N, int64_t voxels, bpx_tensor x, const void* w_packed_d, bpx_tensor g, bpx_tensor t,
                                const bpx_nbwd_coef* coef_d, bpx_tensor y_lo, bpx_tensor y_hi, float* dw_d, void* ws_d, int64_t ws_bytes,
                                bpx_stream_t stream);

/* ConvTranspose3d k = s = (sz,2,2), sz = z_down of the level = 1 or 2 (blocks.py:1607):
 * y[n,sz*z+a,2y+b,2x+c,:] = x[n,z,y,x,:]*W[:,:,a,b,c] + b.
 * (D,H,W) are the INPUT extents; y has extents (sz*D,2H,2W) and may be a channel slice of the
 * concat buffer.  stats_part_d: ([n][tiles][2][Cout]) partials for the following norm.
 * Weights packed with BPX_PK_CT / _T (sz = 2) or BPX_PK_CT4 / _T (sz = 1). */
int bpx_convT3d_k2s2_fwd(int dtype, int N, int D, int H, int W, int sz, bpx_tensor x, const void* w_packed_d,
                         const float* bias_d, bpx_tensor y, float* stats_part_d, bpx_stream_t stream);
int bpx_convT3d_stats_tiles(int D, int H, int W, int sz);
int bpx_convT3d_k2s2_dgrad(int dtype, int N, int D, int H, int W, int sz, bpx_tensor dy, const void* w_packed_T_d,
                           bpx_tensor dx, bpx_stream_t stream);
int64_t bpx_convT3d_k2s2_wgrad_workspace(int N, int D, int H, int W, int sz, int Cin, int Cout);
int bpx_convT3d_k2s2_wgrad(int dtype, int N, int D, int H, int W, int sz, bpx_tensor x, bpx_tensor dy,
                           float* dw_d /* (Cin,Cout,sz,2,2), overwritten */, float* db_d /* accumulated */,
                           void* ws_d, int64_t ws_bytes, bpx_stream_t stream);

/* InstanceNorm3d(affine, eps) == GroupNorm with G = C (blocks.py:2122-2125).  Reduces the partials
 * written by a producer kernel to bpx_norm_rec[n*out_ld + out_off + c]; groups < C gives GroupNorm(groups).
 * out_ld/out_off let two producers (up-conv and skip) fill one record array for the concatenated tensor.
 * The partial array is CONSUMED: with >= 1024 tiles a first pass compacts it in place (no scratch buffer). */
int bpx_norm_finalize(float* stats_part_d, int N, int tiles, int C, int64_t count_per_channel,
                      const float* gamma_d, const float* beta_d, float eps, int groups,
                      bpx_norm_rec* out_d, int out_ld, int out_off, bpx_stream_t stream);
/* GroupNorm(groups) with ANY channels-per-group, also over the concatenation of two producers' outputs (torch.cat([up, skip], 1) in front of
 * the decoder's first norm, blocks.py:1653: with 8 groups over 48 / 96 / 192 / 384 channels one group straddles the boundary).
 *   bpx_norm_channel_sums     : a producer's partials [N][tiles][2][C] (CONSUMED like in bpx_norm_finalize) -> per-channel totals
 *                               sums_d[(n*out_ld + out_off + c)*2 + {0, 1}] (double): each producer fills its columns of one (N, out_ld, 2) array
 *   bpx_groupnorm_finalize    : the records of all C channels from those totals (group statistics = fixed-order sums of the group's channels)
 *   bpx_groupnorm_bwd_finalize: bpx_norm_bwd_finalize for the same layout: sums_d = per-channel totals of {S1 = sum g, S2 = sum g*xhat}
 * The reference's "gn" (nn.GroupNorm(out_channels, num_groups=8), blocks.py:2122-2125) raises a TypeError; what it means - GroupNorm(8, C) -
 * is what these implement, checked against torch.nn.GroupNorm. */
int bpx_norm_channel_sums(float* stats_part_d, int N, int tiles, int C, double* sums_d, int out_ld, int out_off, bpx_stream_t stream);
int bpx_groupnorm_finalize(const double* sums_d, int N, int C, int64_t count_per_channel, const float* gamma_d, const float* beta_d, float eps,
                           int groups, bpx_norm_rec* out_d, bpx_stream_t stream);
int bpx_groupnorm_bwd_finalize(const double* sums_d, int N, int C, int64_t count_per_channel, const bpx_norm_rec* rec_d, const float* gamma_d,
                               float* dgamma_d, float* dbeta_d, int groups, bpx_nbwd_coef* coef_d, bpx_stream_t stream);
/* Stand-alone statistics of a tensor (used for tensors no conv kernel produced, and in tests). */
int bpx_tensor_stats(int dtype, int N, int64_t voxels, bpx_tensor x, float* stats_part_d, bpx_stream_t stream);
int bpx_tensor_stats_tiles(int64_t voxels);

/* Backward of InstanceNorm (groups == C) / GroupNorm(groups) (blocks.py:2117-2125) given the per-channel partials from
 * bpx_conv3d_dgrad / bpx_norm_act_bwd (S1 = sum g, S2 = sum g*xhat with the group's statistics in rec_d):
 *   coef[n*C+c] = {a, b, c0} with dx = a*g + b*t + c0 ;  dgamma[c] += sum_n S2 ; dbeta[c] += sum_n S1
 * (sums over the samples in a fixed order by one thread per channel: deterministic).  Channels per group: 1, 2, 4, 8, 16, 32, 64;
 * count_per_channel = voxels per sample. */
int bpx_norm_bwd_finalize(float* red_part_d /* consumed, see bpx_norm_finalize */, int N, int tiles, int C, int64_t count_per_channel,
                          const bpx_norm_rec* rec_d, const float* gamma_d, float* dgamma_d, float* dbeta_d, int groups,
                          bpx_nbwd_coef* coef_d, bpx_stream_t stream);
/* The same inside a bpx_wgrad_defer_begin / _flush window, one block per (channel block, sample): coef_d is complete when the call's kernel is; the
 * sums over the samples that dgamma / dbeta need are queued with the window's weight-gradient reductions and arrive at the flush (sample order,
 * fixed: bit-reproducible).  The first partial row of every sample is overwritten with that sample's totals: red_part_d must stay untouched
 * until the flush.  Outside a window (or for N = 1) it is bpx_norm_bwd_finalize. */
int bpx_norm_bwd_finalize_deferred(float* red_part_d /* consumed, see bpx_norm_finalize */, int N, int tiles, int C, int64_t count_per_channel,
                          const bpx_norm_rec* rec_d, const float* gamma_d, float* dgamma_d, float* dbeta_d, int groups,
                          bpx_nbwd_coef* coef_d, bpx_stream_t stream);
/* dx = a*g + b*t + c0 (+ addend): applies the coefficients above elementwise. dx may alias g. */
int bpx_norm_bwd_apply(int dtype, int N, int64_t voxels, bpx_tensor g, bpx_tensor t, const bpx_nbwd_coef* coef_d,
                       bpx_tensor addend, bpx_tensor dx, bpx_stream_t stream);

/* Materialised InstanceNorm + activation and its backward, for consumers without a fused prologue (plain U-Net:
 * biapy/models/blocks.py:154-166 Conv -> Norm -> Act feeding MaxPool / ConvTranspose / the head, unet.py:382-394).
 *   fwd: y = act(scale*x + shift), rec_d = [N][C] records of bpx_norm_finalize.
 *   bwd: g = dy * act'(scale*x + shift) (+ addend); red_part_d = [N][bpx_norm_act_tiles()][2][C] partial sums of
 *        S1 = sum g, S2 = sum g*xhat of the product term only, the layout bpx_norm_bwd_finalize reads.  g may alias dy.
 * dtype fwd: BF16, F16, F32; bwd: BF16, F32, or MIX16 (x = the forward pass's fp16 tensor; dy, addend, g bf16 - the mixed training mode). */
int bpx_norm_act_tiles(int dtype, int64_t voxels, int C);
int bpx_norm_act_fwd(int dtype, int N, int64_t voxels, bpx_tensor x, const bpx_norm_rec* rec_d, int act, bpx_tensor y,
                     bpx_stream_t stream);
int bpx_norm_act_bwd(int dtype, int N, int64_t voxels, bpx_tensor dy, bpx_tensor x, const bpx_norm_rec* rec_d, int act,
                     bpx_tensor addend, bpx_tensor g, float* red_part_d, bpx_stream_t stream);

/* Dropout of a residual block (biapy/models/blocks.py:163: `nn.Dropout(p)` after Conv -> Norm -> Act, element-wise) - only for p > 0; p = 0 (BiaPy's
 * default DROPOUT_VALUES) keeps the fused prologue of the block's second convolution.
 *   fwd: y = act(scale*x + shift) * keep / (1 - p)        (the tensor the second convolution then reads WITHOUT a prologue)
 *   bwd: g = dy * keep / (1 - p) * act'(scale*x + shift), red_part_d = [N][bpx_norm_act_dropout_tiles()][2][C] partial sums for bpx_norm_bwd_finalize
 * keep = Philox4x32-10(key = seed, counter = (element / 4, site, *counter_d)) >= p * 2^32 over the linear index of the dense NDHWC tensor: forward and
 * backward regenerate the same mask, nothing is stored; *counter_d is a DEVICE uint64 the caller bumps once per forward pass (a replayed graph
 * draws a new mask each time).  Not the reference's random stream (torch's CPU / CUDA generators cannot be reproduced): parity is held with the
 * mask made explicit - mask_mode 1 reads the keep flags (uint8 per element) from mask_io_d instead of drawing them, 2 also writes the drawn
 * flags there (forward only), 0 ignores it.  dtype fwd: BF16, F16, F32; bwd: BF16, MIX16 (x fp16), F32.  Dense tensors (ld == C). */
int bpx_norm_act_dropout_tiles(int dtype, int64_t voxels, int C);
int bpx_norm_act_dropout_fwd(int dtype, int N, int64_t voxels, bpx_tensor x, const bpx_norm_rec* rec_d, int act, float p, uint64_t seed,
                             const uint64_t* counter_d, int site, uint8_t* mask_io_d, int mask_mode, bpx_tensor y, bpx_stream_t stream);
int bpx_norm_act_dropout_bwd(int dtype, int N, int64_t voxels, bpx_tensor dy, bpx_tensor x, const bpx_norm_rec* rec_d, int act, float p,
                             uint64_t seed, const uint64_t* counter_d, int site, uint8_t* mask_io_d, int mask_mode, bpx_tensor g,
                             float* red_part_d, bpx_stream_t stream);

/* Channel attention of the RCAN trunk (biapy/models/rcan.py: ChannelAttention.forward `x * module(x)`, RCAB_rcan.forward
 * `x + module(x)`): y = [x +] scale[n,c] * h [+ offset[n,c]] with per-(sample, channel) fp32 factors (x.ptr / offset_d may be
 * null; y may alias h), and the reduction its backward needs: part_d[n][bpx_norm_act_tiles()][C] partial sums of a*b over
 * voxels (d scale[n,c] = sum_v dy*h). */
int bpx_channel_affine(int dtype, int N, int64_t voxels, bpx_tensor x, bpx_tensor h, const float* scale_d, const float* offset_d,
                       bpx_tensor y, bpx_stream_t stream);
int bpx_dot_stats(int dtype, int N, int64_t voxels, bpx_tensor a, bpx_tensor b, float* part_d, bpx_stream_t stream);
/* The gate itself, on the pooled (N, C) vector (rcan.py ChannelAttention.module: Conv 1x1 -> SiLU -> Conv 1x1 -> Sigmoid;
 * blocks.py:1119-1191 SqExBlock.excitation: Linear -> ReLU -> Linear -> Sigmoid, no biases):
 *   m[n,c] = sum_t part_d[n][t][0][c] / voxels   (part_d: [N][tiles][2][C] statistics partials of the producing kernel, NOT consumed)
 *   u1 = W1 m + b1 (R values; w1_d [R][C]),  a1 = act(u1),  s = sigmoid(W2 a1 + b2) (w2_d [C][R]);  b1_d / b2_d may be null
 *   saved_d [N][C + 2R] = m, u1, a1 (read by the backward).  C <= 256, R <= 64.
 * Backward: dpart_d [N][tiles][C] partial sums of dy*h (bpx_dot_stats) -> ds[n,c]; dW1 / db1 / dW2 / db2 are ACCUMULATED (+=; db may be
 * null) with the samples summed in index order (deterministic); off_d[n][c] = d mean[n,c] / voxels, the offset of the affine pass
 * that carries the pooled gradient back to every voxel. */
int bpx_gate_mlp_fwd(const float* part_d, int N, int tiles, int C, int64_t voxels, const float* w1_d, const float* b1_d, const float* w2_d,
                     const float* b2_d, int R, int act, float* s_d, float* saved_d, bpx_stream_t stream);
int bpx_gate_mlp_bwd(const float* dpart_d, int N, int tiles, int C, int64_t voxels, const float* s_d, const float* saved_d, const float* w1_d,
                     const float* w2_d, int R, int act, float* dw1_d, float* db1_d, float* dw2_d, float* db2_d, float* off_d,
                     bpx_stream_t stream);

/* MaxPool3d (sz,2,2), sz = z_down of the level = 1 or 2 (resunet.py:256-257) + statistics of the pooled tensor.
 * (D,H,W) = input extents. */
int bpx_maxpool3d_fwd(int dtype, int N, int D, int H, int W, int sz, bpx_tensor x, bpx_tensor y, float* stats_part_d,
                      bpx_stream_t stream);
int bpx_maxpool3d_stats_tiles(int dtype, int D, int H, int W, int sz, int C);
/* dx = addend + scatter(dy to the first maximal element of each window) */
int bpx_maxpool3d_bwd(int dtype, int N, int D, int H, int W, int sz, bpx_tensor x, bpx_tensor dy, bpx_tensor addend,
                      bpx_tensor dx, bpx_stream_t stream);
/* The same (addend required) with the rank-1 shortcut weight gradient of the first residual block riding along (round 6): dw_d[co] (16, 1, 1, 1, 1) =
 * sum_v img_d[v] * dx[v][co] over the values this call STORES - what bpx_conv1x1_c1_wgrad computes from dx afterwards (blocks.py ResConvBlock `shortcut(x)`
 * on a one-channel image).  16 channels, 16-bit storage; _workspace answers the bytes of ws_d or 0 = call the two entry points (BPX_POOL_R1=0 forces that).
 * Between bpx_wgrad_defer_begin and _flush dw_d is complete at the flush and ws_d must stay untouched until then. */
int64_t bpx_maxpool3d_bwd_r1_workspace(int dtype, int N, int D, int H, int W, int sz, int C);
int bpx_maxpool3d_bwd_r1(int dtype, int N, int D, int H, int W, int sz, bpx_tensor x, bpx_tensor dy, bpx_tensor addend, bpx_tensor dx,
                         const float* img_d, float* dw_d, void* ws_d, int64_t ws_bytes, bpx_stream_t stream);

/* Output head: Conv3d k=1 to `Cout` (<= 4) fp32 channels (resunet.py:346-348) with the head
 * activation (base_workflow.py:1403-1457) fused: head_act holds one 4-bit code per output channel (channel 0 in the low
 * nibble): 0 = linear (logits), 1 = sigmoid, 2 = tanh, 3 = softmax, consecutive softmax channels forming one group.
 * out is (N,Cout,D,H,W) fp32 contiguous per channel plane with arbitrary strides given in elements. */
int bpx_head_fwd(int dtype, int64_t voxels_per_sample, int N, bpx_tensor x, const float* w_d /* [Cout][Cin] */,
                 const float* b_d, int Cout, int head_act, float* out_d, int64_t out_stride_n, int64_t out_stride_c,
                 bpx_stream_t stream);
/* bwd: dx, dW (overwritten) and db (added to).  The parameter gradients are sums of per-workgroup partials held in ws_d
 * (>= bpx_head_bwd_workspace bytes) and combined in a fixed order right after the kernel - no atomics, bit-reproducible from run to run. */
int64_t bpx_head_bwd_workspace(int Cin, int Cout);
int bpx_head_bwd(int dtype, int64_t voxels_per_sample, int N, bpx_tensor x, const float* w_d, int Cout,
                 const float* dout_d, int64_t stride_n, int64_t stride_c, bpx_tensor dx, float* dw_d, float* db_d,
                 void* ws_d, int64_t ws_bytes, bpx_stream_t stream);

/* First layer, Cin = 1 (fp32 image in, blocks.py:154 with in_size = 1): direct convolution. */
int bpx_conv3d_c1_fwd(int dtype, int N, int D, int H, int W, const float* img_d, const float* w_d /* (Cout,1,3,3,3) */,
                      const float* bias_d, bpx_tensor y, float* stats_part_d, bpx_stream_t stream);
int bpx_conv3d_c1_stats_tiles(int D, int H, int W);
/* dW (Cout,1,3,3,3) is overwritten, db is added to; partial sums in ws_d (deterministic).  With bpx_wgrad_defer_begin active the
 * combination runs at the flush, like that of bpx_conv3d_wgrad: ws_d must stay untouched until then. */
int64_t bpx_conv3d_c1_wgrad_workspace(int Cout);
int bpx_conv3d_c1_wgrad(int dtype, int N, int D, int H, int W, const float* img_d, bpx_tensor dy,
                        float* dw_d, float* db_d, void* ws_d, int64_t ws_bytes, bpx_stream_t stream);
/* bpx_norm_bwd_apply folded into bpx_conv3d_c1_wgrad: dy = a * g + b * t + c0 (coef_d of bpx_norm_bwd_finalize, t = the first conv's raw output)
 * is formed while the tile is staged and never stored - the first layer has no input gradient, so the weight gradient is dy's only consumer
 * (blocks.py:154-157 under autograd, first ConvBlock of the encoder).  dtype BF16, or MIX16 (t fp16); needs W > 8 and dense 16-byte aligned g / t. */
int bpx_conv3d_c1_wgrad_nb_supported(int dtype, int W);
int bpx_conv3d_c1_wgrad_nb(int dtype, int N, int D, int H, int W, const float* img_d, bpx_tensor g, bpx_tensor t, const bpx_nbwd_coef* coef_d,
                           float* dw_d, float* db_d, void* ws_d, int64_t ws_bytes, bpx_stream_t stream);

/* Shortcut of the first block (Conv3d 1 -> Cout, k = 1, blocks.py:1372 with in_size = 1): dW[co] = sum_v img[v]*dy[v][co]
 * (overwritten; partial sums in ws_d as for bpx_conv3d_c1_wgrad, deterministic, deferrable). */
int64_t bpx_conv1x1_c1_wgrad_workspace(int Cout);
int bpx_conv1x1_c1_wgrad(int dtype, int64_t voxels_total, const float* img_d, bpx_tensor dy, float* dw_d, void* ws_d, int64_t ws_bytes,
                         bpx_stream_t stream);

/* Super-resolution "pre" up-sampling of the 1-channel image: ConvTranspose3d(1, 1, kernel = stride = (fz, fy, fx))
 * (biapy/models/resunet.py:206-213, :368-369).  fwd writes channel 0 of a dense 16-channel NDHWC tensor of the storage dtype
 * (extents fz*D, fy*H, fx*W; channels 1..15 = 0) so that the first residual block can run on the 16-channel kernels with its weights
 * zero-padded.  bwd: partials_d[(t * bpx_upsample_c1_blocks(N*D*H*W) + b) * 2 + {0, 1}] = partial sums of img * g and of g for tap
 * t = (a*fy + b)*fx + c, g = channel 0 of the gradient of that tensor; the caller reduces them (dW[t] = sum of the first, dbias = sum
 * over all taps of the second). */
int bpx_upsample_c1_fwd(int dtype, int N, int D, int H, int W, int fz, int fy, int fx, const float* img_d, const float* w_d, const float* bias_d,
                        void* out16_d, bpx_stream_t stream);
int bpx_upsample_c1_blocks(int64_t voxels_in);
int bpx_upsample_c1_bwd(int dtype, int N, int D, int H, int W, int fz, int fy, int fx, const float* img_d, const void* dx16_d, float* partials_d,
                        bpx_stream_t stream);

/* dtype conversion helpers (NDHWC, strided channel slices) */
int bpx_cast(int src_dtype, const void* src_d, int dst_dtype, void* dst_d, int64_t n, bpx_stream_t stream);

/* ---- binary segmentation loss (1-channel head) ---------------------------------------------------------------------------
 * Replaces biapy/engine/metrics.py:493-586 (CrossEntropyLoss_wrapper -> BCEWithLogitsLoss), :726-762 (DiceLoss, batch_dice),
 * :764-973 (DiceCELoss, binary case) and the counts of :138-232 (jaccard_index at threshold 0.5) with one streaming pass each
 * way.  bpx_seg_loss_sums writes bpx_seg_loss_blocks(n) rows of {sum bce, sum p*t, sum p, sum t, |P&T|, |P|T|}
 * (p = sigmoid(logit), P = p > 0.5, T = t > 0.5); the caller reduces the rows (deterministic) and forms the loss.
 * bpx_seg_loss_bwd: dlogits = a (p - t) - p (1 - p) (b t - c) with coef_d = {a, b, c} on the device
 * (a = w_ce*g/n, b = 2 w_dice*g/(U+s), c = w_dice*g*(2I+s)/(U+s)^2; biapy_amd/losses.py). fp32 planar logits and targets. */
int bpx_seg_loss_blocks(int64_t n);
int bpx_seg_loss_sums(const float* logits_d, const float* target_d, int64_t n, float* partials_d, bpx_stream_t stream);
int bpx_seg_loss_bwd(const float* logits_d, const float* target_d, int64_t n, const float* coef_d, float* dlogits_d, bpx_stream_t stream);
/* The scalar tail of the same losses on the device (no host-side element-wise launches inside a captured training step):
 *   bpx_seg_loss_finish   : sums_d[6] (double) = the fixed-order column sums of the `blocks` partial rows; loss_d (float) =
 *                           w_ce * S0 / n + w_dice * (1 - (2 S1 + smooth) / (S2 + S3 + smooth))
 *   bpx_seg_loss_bwd_fused: bpx_seg_loss_bwd with a, b, c formed in the kernel from sums_d and the upstream gradient gup_d[0] (device float). */
int bpx_seg_loss_finish(const float* partials_d, int blocks, int64_t n, float w_ce, float w_dice, float smooth, double* sums_d, float* loss_d,
                        bpx_stream_t stream);
int bpx_seg_loss_bwd_fused(const float* logits_d, const float* target_d, int64_t n, const double* sums_d, const float* gup_d, float w_ce, float w_dice,
                           float smooth, float* dlogits_d, bpx_stream_t stream);

/* ---- per-channel losses of a multi-channel head (row X / cfg 4: instance segmentation with B, C, D channels) ------------------------------
 * Replaces biapy/engine/metrics.py:1418-1810 (instance_segmentation_loss; plain channels: no masks, class re-balancing or border
 * weights) composed with the training-time head activation the workflow applies before it (base_workflow.py:1403-1457: ce_* channels
 * stay logits, other channels - 'D': tanh, instance_seg.py:405-409 - are activated).  codes: one 4-bit code per channel (channel 0
 * in the low nibble) = kind | act << 2; kind 0 = BCE with logits, 1 = MSE, 2 = L1; act (of the logit, for MSE / L1) 0 = linear,
 * 1 = tanh, 2 = sigmoid.  Planar fp32 logits / targets [N][C][voxels], C <= 8.
 *   bpx_chan_loss_sums: partials_d[(n*C + c) * bpx_chan_loss_blocks(voxels) + b] = partial sums of the per-element loss terms;
 *                       the caller reduces them (deterministic) and forms  sum_c w_c * S_c / (N * voxels).
 *   bpx_chan_loss_bwd : dlogits = coef_d[c] * d term / d logit  with coef_d[c] = upstream gradient * w_c / (N * voxels) on the device. */
int bpx_chan_loss_blocks(int64_t voxels);
int bpx_chan_loss_sums(const float* logits_d, const float* target_d, int N, int C, int64_t voxels, unsigned codes, float* partials_d,
                       bpx_stream_t stream);
int bpx_chan_loss_bwd(const float* logits_d, const float* target_d, int N, int C, int64_t voxels, unsigned codes, const float* coef_d,
                      float* dlogits_d, bpx_stream_t stream);
/* Multi-class cross entropy of the semantic-segmentation head and the counts of the multi-class IoU (replaces biapy/engine/metrics.py:493-586
 * CrossEntropyLoss_wrapper with num_classes > 2 = torch.nn.CrossEntropyLoss(ignore_index, weight), mean reduction, and the confusion counts behind
 * :138-232 jaccard_index).  logits (N, C, voxels) fp32 planar, 2 <= C <= 8; target (N, 1, voxels) class ids as floats; class_w_d: C floats or NULL;
 * labels equal to ignore_index (or outside [0, C)) are not counted.
 *   bpx_softmax_ce_sums  : partials_d[N * bpx_softmax_ce_blocks(voxels)][bpx_softmax_ce_row()] = {sum w nll, sum w, tp[8], pred[8], tgt[8]} per block
 *   bpx_softmax_ce_finish: sums_d[bpx_softmax_ce_row()] (double, fixed summation order), loss_d = sums[0] / sums[1]
 *   bpx_softmax_ce_bwd   : dlogits = gup_d[0] * w[y] / sums[1] * (softmax - onehot(y)), 0 for uncounted labels. */
int bpx_softmax_ce_blocks(int64_t voxels);
int bpx_softmax_ce_row(void);
int bpx_softmax_ce_sums(const float* logits_d, const float* target_d, int N, int C, int64_t voxels, int ignore_index, const float* class_w_d,
                        float* partials_d, bpx_stream_t stream);
int bpx_softmax_ce_finish(const float* partials_d, int N, int64_t voxels, double* sums_d, float* loss_d, bpx_stream_t stream);
int bpx_softmax_ce_bwd(const float* logits_d, const float* target_d, int N, int C, int64_t voxels, int ignore_index, const float* class_w_d,
                       const double* sums_d, const float* gup_d, float* dlogits_d, bpx_stream_t stream);
/*   bpx_chan_loss_finish  : loss_d = sum_c weights_d[c] * (fixed-order double sum of channel c's partials) / (N * voxels)
 *   bpx_chan_loss_bwd_fused: bpx_chan_loss_bwd with coef[c] = gup_d[0] * weights_d[c] / (N * voxels) formed in the kernel. */
int bpx_chan_loss_finish(const float* partials_d, int N, int C, int64_t voxels, const float* weights_d, float* loss_d, bpx_stream_t stream);
int bpx_chan_loss_bwd_fused(const float* logits_d, const float* target_d, int N, int C, int64_t voxels, unsigned codes, const float* weights_d,
                            const float* gup_d, float* dlogits_d, bpx_stream_t stream);

/* ---- attention gate of ResUNet++ (biapy/models/blocks.py:2168-2298, `return out * x2`: a 1-channel map times a C-channel tensor) ----
 * fwd: y[v][c] = a[v] * x[v][c], a = the FIRST channel of tensor `a` (channel stride a.ld);  bwd: dx = dy * a and
 * da16[v][0] = sum_c dy[v][c] * x[v][c], da16[v][1..15] = 0 (a dense 16-channel tensor: the gradient of the zero-padded 16-output
 * 1x1 convolution that produced the gate).  total_voxels = N * voxels per sample. */
int bpx_gate_mul_fwd(int dtype, int64_t total_voxels, bpx_tensor a, bpx_tensor x, bpx_tensor y, bpx_stream_t stream);
int bpx_gate_mul_bwd(int dtype, int64_t total_voxels, bpx_tensor dy, bpx_tensor a, bpx_tensor x, bpx_tensor dx, void* da16_d, bpx_stream_t stream);

/* ---- scans either side of the network ---------------------------------------------------------------
 * Input normalisation (biapy/data/norm.py: percentile_clip :395-473 -> np.percentile, zero_mean_unit_variance_normalization
 * :586-645) and the Otsu binarisation after the merge (biapy/engine/semantic_seg.py:418-431) on volumes that live on the device.
 *   bpx_select_kth_f32 : exact k-th smallest element (0-based), 4-pass radix select, result written to out_d (device float)
 *   bpx_minmax_f32     : per-block {min, max} partials, bpx_scan_blocks(n) rows
 *   bpx_moment_f32     : per-block partial sums of (x - center)^power in double (power 1 or 2), bpx_scan_blocks(n) rows
 *   bpx_histogram_f32  : np.histogram(a, bins=nbins, range=(first, last)) of a float32 array, bit-identical counts;
 *                        edges_d = the float32 edge table NumPy builds (np.linspace(first, last, nbins + 1, dtype=float32));
 *                        counts_d (uint64[nbins]) is ACCUMULATED into (zero it first)
 *   bpx_threshold_u8   : out = x > thr
 *   bpx_clip_affine_f32: out = (clip(x, lo, hi) - sub) / div in float32 operations */
int bpx_scan_blocks(int64_t n);
int64_t bpx_select_workspace(void);
int bpx_select_kth_f32(const float* x_d, int64_t n, int64_t k, float* out_d, void* ws_d, bpx_stream_t stream);
int bpx_minmax_f32(const float* x_d, int64_t n, float* partials_d, bpx_stream_t stream);
int bpx_moment_f32(const float* x_d, int64_t n, double center, int power, double* partials_d, bpx_stream_t stream);
int bpx_histogram_f32(const float* x_d, int64_t n, float first_edge, float last_edge, int nbins, const float* edges_d,
                      unsigned long long* counts_d, bpx_stream_t stream);
int bpx_threshold_u8(const float* x_d, int64_t n, float thr, uint8_t* out_d, bpx_stream_t stream);
int bpx_clip_affine_f32(const float* x_d, int64_t n, float lo, float hi, float sub, float div, float* out_d, bpx_stream_t stream);
/* Class head of the sliding-window harness (biapy/engine/base_workflow.py:2135-2141, `separated_class_channel`): the last k channels of a
 * (voxels, C) float32 volume are replaced by ONE channel holding np.argmax over them (first maximum), out_d is (voxels, C - k + 1). */
int bpx_class_argmax(const float* in_d, int64_t voxels, int C, int k, float* out_d, bpx_stream_t stream);

/* ---- test-time augmentation -------------------------------------------------------------------------
 * Signed axis permutations of a (Z,Y,X,C) float32 volume (biapy/data/post_processing/tta.py:64-196, AxisTransform: output
 * axis a comes from input axis perm[a], reversed when sign[a] == -1; 2D images use Z = 1) and the reduction of
 * post_processing.py:1349-1383.  (Z,Y,X) are always the extents of the UN-oriented volume.
 *   bpx_tta_orient     : out = t.apply(in); out has extents (n[perm[0]], n[perm[1]], n[perm[2]])
 *   bpx_tta_accumulate : acc (op)= t.inverse.apply(pred), op: first != 0 -> assign, else mode 0 add / 1 min / 2 max;
 *                        count_if_last > 0 with mode 0 divides the finished sum by that count (np.mean's sum / n). */
int bpx_tta_orient(const float* in_d, int Z, int Y, int X, int C, const int* perm, const int* sign, float* out_d, bpx_stream_t stream);
int bpx_tta_accumulate(const float* pred_d, int Z, int Y, int X, int C, const int* perm, const int* sign, int mode, int first,
                       int count_if_last, float* acc_d, bpx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BIAPY_AMD_H */
