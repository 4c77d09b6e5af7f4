// Does gfx950 need wait states between a 128-bit buffer store and a VALU write of its data registers when the store's soffset is an SGPR?
// LLVM's hazard recogniser (hipcc 7.2) inserts them only when soffset is NOT a register (GCNHazardRecognizer::createsVALUHazard); round 5's
// transposed-conv kernel lost two of four data registers with a register soffset.  Here the adjacency is forced in
// inline assembly for both forms, with 0 .. 2 s_nop between store and overwrite; the kernel then checks what arrived in memory.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-result scripts/probes/soffset_store_hazard.hip -o scripts/probes/soffset_store_hazard.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
template <int FORM, int NOPS, int TWO> __global__ void k(unsigned* out, unsigned soff_rt) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 1 << 20, 0x00020000);
  const unsigned lane = threadIdx.x;
  const unsigned d[4] = {lane, lane + 100u, lane + 200u, lane + 300u};
  unsigned voff = lane * 16u + (FORM == 0 ? soff_rt : 0u);
  unsigned so = __builtin_amdgcn_readfirstlane(soff_rt);
  // fixed registers v[20:23]: fill, store, (s_nop), overwrite ALL FOUR data registers - everything in one assembly block, so that neither the
  // compiler's scheduler nor its hazard recogniser can change what follows the store
#define FILL "v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_mov_b32 v22, %2\n v_mov_b32 v23, %3\n s_nop 4\n"
#define KILL "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
#define NOP0 ""
#define NOP1 "s_nop 0\n"
#define NOP2 "s_nop 1\n"
#define ST_SGPR "buffer_store_dwordx4 v[20:23], %4, %5, %6 offen\n"
#define ST_IMM "buffer_store_dwordx4 v[20:23], %4, %5, 0 offen\n"
  // TWO: a first 128-bit store (other registers, 4 KB further) right in front of the one under test, and the overwrite is the packed add that
  // followed it in the kernel (v_pk_add_f32 into the upper two data registers)
#define FILL2 "v_mov_b32 v24, %0\n v_mov_b32 v25, %1\n v_mov_b32 v26, %2\n v_mov_b32 v27, %3\n v_add_u32 v28, 0x1000, %4\n s_nop 4\n"
#define ST2_SGPR "buffer_store_dwordx4 v[24:27], v28, %5, %6 offen\n"
#define ST2_IMM "buffer_store_dwordx4 v[24:27], v28, %5, 0 offen\n"
#define KILLPK "v_pk_add_f32 v[22:23], v[24:25], v[26:27]\n v_pk_add_f32 v[20:21], v[24:25], v[26:27]\n"
#define GO(ST, NP) asm volatile(FILL ST NP KILL : : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(voff), "s"(rs), "s"(so) : "memory", "v20", "v21", "v22", "v23")
#define GO2(S2, ST, NP) asm volatile(FILL FILL2 S2 ST NP KILLPK : : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(voff), "s"(rs), "s"(so) : "memory", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28")
  if (!TWO) {
    if (FORM == 1) { if (NOPS == 0) GO(ST_SGPR, NOP0); if (NOPS == 1) GO(ST_SGPR, NOP1); if (NOPS == 2) GO(ST_SGPR, NOP2); }
    else { if (NOPS == 0) GO(ST_IMM, NOP0); if (NOPS == 1) GO(ST_IMM, NOP1); if (NOPS == 2) GO(ST_IMM, NOP2); }
  } else {
    if (FORM == 1) { if (NOPS == 0) GO2(ST2_SGPR, ST_SGPR, NOP0); if (NOPS == 1) GO2(ST2_SGPR, ST_SGPR, NOP1); if (NOPS == 2) GO2(ST2_SGPR, ST_SGPR, NOP2); }
    else { if (NOPS == 0) GO2(ST2_IMM, ST_IMM, NOP0); if (NOPS == 1) GO2(ST2_IMM, ST_IMM, NOP1); if (NOPS == 2) GO2(ST2_IMM, ST_IMM, NOP2); }
  }
}
template <int FORM, int NOPS, int TWO> int run(unsigned* dbuf, unsigned* h, const char* name) {
  hipMemset(dbuf, 0xFF, 1 << 20);
  k<FORM, NOPS, TWO><<<1, 64>>>(dbuf, 4096u);
  hipMemcpy(h, dbuf, 1 << 16, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const unsigned* p = h + (4096 + l * 16) / 4;
    for (int e = 0; e < 4; ++e) bad += p[e] != (unsigned)(l + 100 * e);
  }
  printf("%-28s %s %d s_nop between store and overwrite: %3d of 256 stored words wrong\n", name, TWO ? "second of two stores, v_pk_add_f32 overwrite," : "one store, v_mov_b32 overwrite,              ", NOPS, bad);
  return bad;
}
int main() {
  unsigned *d, *h = (unsigned*)malloc(1 << 16);
  hipMalloc(&d, 1 << 20);
  run<0, 0, 0>(d, h, "soffset = 0 (immediate)"); run<0, 1, 0>(d, h, "soffset = 0 (immediate)"); run<0, 2, 0>(d, h, "soffset = 0 (immediate)");
  run<1, 0, 0>(d, h, "soffset in an SGPR"); run<1, 1, 0>(d, h, "soffset in an SGPR"); run<1, 2, 0>(d, h, "soffset in an SGPR");
  run<0, 0, 1>(d, h, "soffset = 0 (immediate)"); run<0, 1, 1>(d, h, "soffset = 0 (immediate)"); run<0, 2, 1>(d, h, "soffset = 0 (immediate)");
  run<1, 0, 1>(d, h, "soffset in an SGPR"); run<1, 1, 1>(d, h, "soffset in an SGPR"); run<1, 2, 1>(d, h, "soffset in an SGPR");
  return 0;
}
