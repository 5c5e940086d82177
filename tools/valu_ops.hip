// Per-opcode issue cost on gfx950 (companion of tools/valu_rate.hip): for each vector opcode the step kernels use, the cycles one
// SIMD spends per wave-instruction with 4 waves resident (slowest wave's time / instructions of the four waves).  Eight
// independent instructions of the opcode per group, 8 groups per pass.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_stamp/valu_ops tools/valu_ops.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define REPS 1000
#define R8(x) x x x x x x x x
// X(index, name, text of 8 instructions over v-registers %0..%7 (32 bit), %8..%11 (64 bit), s-registers %12..%15)
#define OPS(X) \
  X(0, "v_add_u32", "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %4, %4, %0\n v_add_u32 %5, %5, %0\n v_add_u32 %6, %6, %0\n v_add_u32 %7, %7, %0\n") \
  X(1, "v_sub_u32", "v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4\n v_sub_u32 %4, %4, %0\n v_sub_u32 %5, %5, %0\n v_sub_u32 %6, %6, %0\n v_sub_u32 %7, %7, %0\n") \
  X(2, "v_and_b32", "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4\n v_and_b32 %4, %4, %0\n v_and_b32 %5, %5, %0\n v_and_b32 %6, %6, %0\n v_and_b32 %7, %7, %0\n") \
  X(3, "v_or_b32", "v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4\n v_or_b32 %4, %4, %0\n v_or_b32 %5, %5, %0\n v_or_b32 %6, %6, %0\n v_or_b32 %7, %7, %0\n") \
  X(4, "v_xor_b32", "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n v_xor_b32 %4, %4, %0\n v_xor_b32 %5, %5, %0\n v_xor_b32 %6, %6, %0\n v_xor_b32 %7, %7, %0\n") \
  X(5, "v_mov_b32", "v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7\n v_mov_b32 %4, %0\n v_mov_b32 %5, %1\n v_mov_b32 %6, %2\n v_mov_b32 %7, %3\n") \
  X(6, "v_mov_b64", "v_mov_b64 %8, %10\n v_mov_b64 %9, %11\n v_mov_b64 %10, %8\n v_mov_b64 %11, %9\n v_mov_b64 %8, %10\n v_mov_b64 %9, %11\n v_mov_b64 %10, %8\n v_mov_b64 %11, %9\n") \
  X(7, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %4\n v_lshlrev_b32 %1, 1, %5\n v_lshlrev_b32 %2, 1, %6\n v_lshlrev_b32 %3, 1, %7\n v_lshlrev_b32 %4, 1, %0\n v_lshlrev_b32 %5, 1, %1\n v_lshlrev_b32 %6, 1, %2\n v_lshlrev_b32 %7, 1, %3\n") \
  X(8, "v_lshrrev_b32 (vgpr amount)", "v_lshrrev_b32 %0, %4, %0\n v_lshrrev_b32 %1, %4, %1\n v_lshrrev_b32 %2, %4, %2\n v_lshrrev_b32 %3, %4, %3\n v_lshrrev_b32 %4, %0, %4\n v_lshrrev_b32 %5, %0, %5\n v_lshrrev_b32 %6, %0, %6\n v_lshrrev_b32 %7, %0, %7\n") \
  X(9, "v_lshl_or_b32", "v_lshl_or_b32 %0, %4, 3, %0\n v_lshl_or_b32 %1, %4, 3, %1\n v_lshl_or_b32 %2, %4, 3, %2\n v_lshl_or_b32 %3, %4, 3, %3\n v_lshl_or_b32 %4, %0, 3, %4\n v_lshl_or_b32 %5, %0, 3, %5\n v_lshl_or_b32 %6, %0, 3, %6\n v_lshl_or_b32 %7, %0, 3, %7\n") \
  X(10, "v_lshl_add_u32", "v_lshl_add_u32 %0, %4, 3, %0\n v_lshl_add_u32 %1, %4, 3, %1\n v_lshl_add_u32 %2, %4, 3, %2\n v_lshl_add_u32 %3, %4, 3, %3\n v_lshl_add_u32 %4, %0, 3, %4\n v_lshl_add_u32 %5, %0, 3, %5\n v_lshl_add_u32 %6, %0, 3, %6\n v_lshl_add_u32 %7, %0, 3, %7\n") \
  X(11, "v_bfe_u32", "v_bfe_u32 %0, %4, 3, 1\n v_bfe_u32 %1, %5, 3, 1\n v_bfe_u32 %2, %6, 3, 1\n v_bfe_u32 %3, %7, 3, 1\n v_bfe_u32 %4, %0, 3, 1\n v_bfe_u32 %5, %1, 3, 1\n v_bfe_u32 %6, %2, 3, 1\n v_bfe_u32 %7, %3, 3, 1\n") \
  X(12, "v_bfe_u32 (sgpr offset)", "v_bfe_u32 %0, %4, %12, 1\n v_bfe_u32 %1, %5, %12, 1\n v_bfe_u32 %2, %6, %12, 1\n v_bfe_u32 %3, %7, %12, 1\n v_bfe_u32 %4, %0, %13, 1\n v_bfe_u32 %5, %1, %13, 1\n v_bfe_u32 %6, %2, %13, 1\n v_bfe_u32 %7, %3, %13, 1\n") \
  X(13, "v_and_or_b32", "v_and_or_b32 %0, %4, %5, %0\n v_and_or_b32 %1, %4, %5, %1\n v_and_or_b32 %2, %4, %5, %2\n v_and_or_b32 %3, %4, %5, %3\n v_and_or_b32 %4, %0, %1, %4\n v_and_or_b32 %5, %0, %1, %5\n v_and_or_b32 %6, %0, %1, %6\n v_and_or_b32 %7, %0, %1, %7\n") \
  X(14, "v_or3_b32", "v_or3_b32 %0, %4, %5, %0\n v_or3_b32 %1, %4, %5, %1\n v_or3_b32 %2, %4, %5, %2\n v_or3_b32 %3, %4, %5, %3\n v_or3_b32 %4, %0, %1, %4\n v_or3_b32 %5, %0, %1, %5\n v_or3_b32 %6, %0, %1, %6\n v_or3_b32 %7, %0, %1, %7\n") \
  X(15, "v_add3_u32", "v_add3_u32 %0, %4, %5, %0\n v_add3_u32 %1, %4, %5, %1\n v_add3_u32 %2, %4, %5, %2\n v_add3_u32 %3, %4, %5, %3\n v_add3_u32 %4, %0, %1, %4\n v_add3_u32 %5, %0, %1, %5\n v_add3_u32 %6, %0, %1, %6\n v_add3_u32 %7, %0, %1, %7\n") \
  X(16, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %4, %1\n v_bcnt_u32_b32 %2, %4, %2\n v_bcnt_u32_b32 %3, %4, %3\n v_bcnt_u32_b32 %4, %0, %4\n v_bcnt_u32_b32 %5, %0, %5\n v_bcnt_u32_b32 %6, %0, %6\n v_bcnt_u32_b32 %7, %0, %7\n") \
  X(17, "v_ffbl_b32", "v_ffbl_b32 %0, %4\n v_ffbl_b32 %1, %5\n v_ffbl_b32 %2, %6\n v_ffbl_b32 %3, %7\n v_ffbl_b32 %4, %0\n v_ffbl_b32 %5, %1\n v_ffbl_b32 %6, %2\n v_ffbl_b32 %7, %3\n") \
  X(18, "v_min_u32", "v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4\n v_min_u32 %4, %4, %0\n v_min_u32 %5, %5, %0\n v_min_u32 %6, %6, %0\n v_min_u32 %7, %7, %0\n") \
  X(19, "v_cmp_eq_u32 -> vcc", "v_cmp_eq_u32 vcc, %0, %4\n v_cmp_eq_u32 vcc, %1, %4\n v_cmp_eq_u32 vcc, %2, %4\n v_cmp_eq_u32 vcc, %3, %4\n v_cmp_eq_u32 vcc, %4, %0\n v_cmp_eq_u32 vcc, %5, %0\n v_cmp_eq_u32 vcc, %6, %0\n v_cmp_eq_u32 vcc, %7, %0\n") \
  X(20, "v_cmp_ne_u64 -> vcc", "v_cmp_ne_u64 vcc, %8, %9\n v_cmp_ne_u64 vcc, %9, %10\n v_cmp_ne_u64 vcc, %10, %11\n v_cmp_ne_u64 vcc, %11, %8\n v_cmp_ne_u64 vcc, %8, %10\n v_cmp_ne_u64 vcc, %9, %11\n v_cmp_ne_u64 vcc, %10, %8\n v_cmp_ne_u64 vcc, %11, %9\n") \
  X(21, "v_cndmask_b32 (vcc, set once)", "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %0, vcc\n v_cndmask_b32 %5, %5, %0, vcc\n v_cndmask_b32 %6, %6, %0, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n") \
  X(22, "v_cndmask_b32 (sgpr pair)", "v_cndmask_b32 %0, %0, %4, s[30:31]\n v_cndmask_b32 %1, %1, %4, s[30:31]\n v_cndmask_b32 %2, %2, %4, s[30:31]\n v_cndmask_b32 %3, %3, %4, s[30:31]\n v_cndmask_b32 %4, %4, %0, s[30:31]\n v_cndmask_b32 %5, %5, %0, s[30:31]\n v_cndmask_b32 %6, %6, %0, s[30:31]\n v_cndmask_b32 %7, %7, %0, s[30:31]\n") \
  X(23, "v_readlane_b32 (const lane)", "v_readlane_b32 %12, %0, 3\n v_readlane_b32 %13, %1, 7\n v_readlane_b32 %14, %2, 11\n v_readlane_b32 %15, %3, 13\n v_readlane_b32 %12, %4, 3\n v_readlane_b32 %13, %5, 7\n v_readlane_b32 %14, %6, 11\n v_readlane_b32 %15, %7, 13\n") \
  X(24, "v_readfirstlane_b32", "v_readfirstlane_b32 %12, %0\n v_readfirstlane_b32 %13, %1\n v_readfirstlane_b32 %14, %2\n v_readfirstlane_b32 %15, %3\n v_readfirstlane_b32 %12, %4\n v_readfirstlane_b32 %13, %5\n v_readfirstlane_b32 %14, %6\n v_readfirstlane_b32 %15, %7\n") \
  X(25, "v_writelane_b32", "v_writelane_b32 %0, %12, 3\n v_writelane_b32 %1, %13, 7\n v_writelane_b32 %2, %14, 11\n v_writelane_b32 %3, %15, 13\n v_writelane_b32 %4, %12, 4\n v_writelane_b32 %5, %13, 8\n v_writelane_b32 %6, %14, 12\n v_writelane_b32 %7, %15, 14\n") \
  X(26, "v_add_u32 dpp row_shr", "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %7, %7 row_shr:8 row_mask:0xf bank_mask:0xf\n") \
  X(27, "v_add_u32 dpp row_bcast15", "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_u32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %5, %5, %5 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_u32_dpp %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n") \
  X(28, "v_lshlrev_b64", "v_lshlrev_b64 %8, 1, %8\n v_lshlrev_b64 %9, 1, %9\n v_lshlrev_b64 %10, 1, %10\n v_lshlrev_b64 %11, 1, %11\n v_lshlrev_b64 %8, 2, %8\n v_lshlrev_b64 %9, 2, %9\n v_lshlrev_b64 %10, 2, %10\n v_lshlrev_b64 %11, 2, %11\n") \
  X(29, "v_lshl_add_u64", "v_lshl_add_u64 %8, %8, 1, %9\n v_lshl_add_u64 %9, %9, 1, %10\n v_lshl_add_u64 %10, %10, 1, %11\n v_lshl_add_u64 %11, %11, 1, %8\n v_lshl_add_u64 %8, %8, 2, %10\n v_lshl_add_u64 %9, %9, 2, %11\n v_lshl_add_u64 %10, %10, 2, %8\n v_lshl_add_u64 %11, %11, 2, %9\n") \
  X(30, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, %4, %0\n v_mbcnt_lo_u32_b32 %1, %4, %1\n v_mbcnt_lo_u32_b32 %2, %4, %2\n v_mbcnt_lo_u32_b32 %3, %4, %3\n v_mbcnt_hi_u32_b32 %4, %0, %4\n v_mbcnt_hi_u32_b32 %5, %0, %5\n v_mbcnt_hi_u32_b32 %6, %0, %6\n v_mbcnt_hi_u32_b32 %7, %0, %7\n") \
  X(31, "v_bitop3_b32", "v_bitop3_b32 %0, %4, %5, %0 bitop3:0x96\n v_bitop3_b32 %1, %4, %5, %1 bitop3:0x96\n v_bitop3_b32 %2, %4, %5, %2 bitop3:0x96\n v_bitop3_b32 %3, %4, %5, %3 bitop3:0x96\n v_bitop3_b32 %4, %0, %1, %4 bitop3:0x96\n v_bitop3_b32 %5, %0, %1, %5 bitop3:0x96\n v_bitop3_b32 %6, %0, %1, %6 bitop3:0x96\n v_bitop3_b32 %7, %0, %1, %7 bitop3:0x96\n") \
  X(32, "v_add_f32", "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %4, %4, %0\n v_add_f32 %5, %5, %0\n v_add_f32 %6, %6, %0\n v_add_f32 %7, %7, %0\n") \
  X(33, "v_and_b32 (sgpr operand)", "v_and_b32 %0, %12, %0\n v_and_b32 %1, %12, %1\n v_and_b32 %2, %12, %2\n v_and_b32 %3, %12, %3\n v_and_b32 %4, %13, %4\n v_and_b32 %5, %13, %5\n v_and_b32 %6, %13, %6\n v_and_b32 %7, %13, %7\n") \
  X(34, "v_add_u32 (sgpr operand)", "v_add_u32 %0, %12, %0\n v_add_u32 %1, %12, %1\n v_add_u32 %2, %12, %2\n v_add_u32 %3, %12, %3\n v_add_u32 %4, %13, %4\n v_add_u32 %5, %13, %5\n v_add_u32 %6, %13, %6\n v_add_u32 %7, %13, %7\n") \
  X(35, "v_add_u32 (other sources)", "v_add_u32 %0, %4, %5\n v_add_u32 %1, %5, %6\n v_add_u32 %2, %6, %7\n v_add_u32 %3, %7, %4\n v_add_u32 %4, %0, %1\n v_add_u32 %5, %1, %2\n v_add_u32 %6, %2, %3\n v_add_u32 %7, %3, %0\n") \
  X(36, "v_and_b32 (other sources)", "v_and_b32 %0, %4, %5\n v_and_b32 %1, %5, %6\n v_and_b32 %2, %6, %7\n v_and_b32 %3, %7, %4\n v_and_b32 %4, %0, %1\n v_and_b32 %5, %1, %2\n v_and_b32 %6, %2, %3\n v_and_b32 %7, %3, %0\n") \
  X(37, "v_perm_b32", "v_perm_b32 %0, %4, %5, %0\n v_perm_b32 %1, %4, %5, %1\n v_perm_b32 %2, %4, %5, %2\n v_perm_b32 %3, %4, %5, %3\n v_perm_b32 %4, %0, %1, %4\n v_perm_b32 %5, %0, %1, %5\n v_perm_b32 %6, %0, %1, %6\n v_perm_b32 %7, %0, %1, %7\n") \
  X(38, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4\n v_pk_add_u16 %4, %4, %0\n v_pk_add_u16 %5, %5, %0\n v_pk_add_u16 %6, %6, %0\n v_pk_add_u16 %7, %7, %0\n") \
  X(39, "v_mov_b32 dpp row_shr", "v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n") \
  X(40, "v_add_u32 + s_nop 0 (per pair)", "v_add_u32 %0, %0, %4\n s_nop 0\n v_add_u32 %1, %1, %4\n s_nop 0\n v_add_u32 %2, %2, %4\n s_nop 0\n v_add_u32 %3, %3, %4\n s_nop 0\n v_add_u32 %4, %4, %0\n s_nop 0\n v_add_u32 %5, %5, %0\n s_nop 0\n v_add_u32 %6, %6, %0\n s_nop 0\n v_add_u32 %7, %7, %0\n s_nop 0\n") \
  X(41, "v_add_u32 + s_waitcnt (per pair)", "v_add_u32 %0, %0, %4\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %4\n s_waitcnt vmcnt(0)\n v_add_u32 %2, %2, %4\n s_waitcnt lgkmcnt(0)\n v_add_u32 %3, %3, %4\n s_waitcnt vmcnt(0)\n v_add_u32 %4, %4, %0\n s_waitcnt lgkmcnt(0)\n v_add_u32 %5, %5, %0\n s_waitcnt vmcnt(0)\n v_add_u32 %6, %6, %0\n s_waitcnt lgkmcnt(0)\n v_add_u32 %7, %7, %0\n s_waitcnt vmcnt(0)\n") \
  X(42, "v_bfe_u32 + s_add_u32 (per pair)", "v_bfe_u32 %0, %4, 3, 1\n s_add_u32 %12, %12, %13\n v_bfe_u32 %1, %5, 3, 1\n s_and_b32 %13, %13, %14\n v_bfe_u32 %2, %6, 3, 1\n s_add_u32 %14, %14, %15\n v_bfe_u32 %3, %7, 3, 1\n s_xor_b32 %15, %15, %12\n v_bfe_u32 %4, %0, 3, 1\n s_add_u32 %12, %12, %13\n v_bfe_u32 %5, %1, 3, 1\n s_and_b32 %13, %13, %14\n v_bfe_u32 %6, %2, 3, 1\n s_add_u32 %14, %14, %15\n v_bfe_u32 %7, %3, 3, 1\n s_xor_b32 %15, %15, %12\n") \
  X(43, "v_bfe_u32 + 2 scalar (per triple)", "v_bfe_u32 %0, %4, 3, 1\n s_add_u32 %12, %12, %13\n s_and_b32 %13, %13, %14\n v_bfe_u32 %1, %5, 3, 1\n s_add_u32 %14, %14, %15\n s_xor_b32 %15, %15, %12\n v_bfe_u32 %2, %6, 3, 1\n s_add_u32 %12, %12, %13\n s_and_b32 %13, %13, %14\n v_bfe_u32 %3, %7, 3, 1\n s_add_u32 %14, %14, %15\n s_xor_b32 %15, %15, %12\n v_bfe_u32 %4, %0, 3, 1\n s_add_u32 %12, %12, %13\n s_and_b32 %13, %13, %14\n v_bfe_u32 %5, %1, 3, 1\n s_add_u32 %14, %14, %15\n s_xor_b32 %15, %15, %12\n v_bfe_u32 %6, %2, 3, 1\n s_add_u32 %12, %12, %13\n s_and_b32 %13, %13, %14\n v_bfe_u32 %7, %3, 3, 1\n s_add_u32 %14, %14, %15\n s_xor_b32 %15, %15, %12\n") \
  X(44, "v_bfe_u32 + s_nop 0 (per pair)", "v_bfe_u32 %0, %4, 3, 1\n s_nop 0\n v_bfe_u32 %1, %5, 3, 1\n s_nop 0\n v_bfe_u32 %2, %6, 3, 1\n s_nop 0\n v_bfe_u32 %3, %7, 3, 1\n s_nop 0\n v_bfe_u32 %4, %0, 3, 1\n s_nop 0\n v_bfe_u32 %5, %1, 3, 1\n s_nop 0\n v_bfe_u32 %6, %2, 3, 1\n s_nop 0\n v_bfe_u32 %7, %3, 3, 1\n s_nop 0\n") \
  X(45, "v_add_u32 + v_bfe_u32 (per pair)", "v_add_u32 %0, %0, %4\n v_bfe_u32 %1, %5, 3, 1\n v_add_u32 %2, %2, %4\n v_bfe_u32 %3, %7, 3, 1\n v_add_u32 %4, %4, %0\n v_bfe_u32 %5, %1, 3, 1\n v_add_u32 %6, %6, %0\n v_bfe_u32 %7, %3, 3, 1\n v_add_u32 %0, %0, %4\n v_bfe_u32 %1, %5, 3, 1\n v_add_u32 %2, %2, %4\n v_bfe_u32 %3, %7, 3, 1\n v_add_u32 %4, %4, %0\n v_bfe_u32 %5, %1, 3, 1\n v_add_u32 %6, %6, %0\n v_bfe_u32 %7, %3, 3, 1\n") \
  X(46, "v_mov_b32 v, s", "v_mov_b32 %0, %12\n v_mov_b32 %1, %13\n v_mov_b32 %2, %14\n v_mov_b32 %3, %15\n v_mov_b32 %4, %12\n v_mov_b32 %5, %13\n v_mov_b32 %6, %14\n v_mov_b32 %7, %15\n") \
  X(47, "v_mov_b32 v, 0", "v_mov_b32 %0, 0\n v_mov_b32 %1, 1\n v_mov_b32 %2, 2\n v_mov_b32 %3, 3\n v_mov_b32 %4, 4\n v_mov_b32 %5, 5\n v_mov_b32 %6, 6\n v_mov_b32 %7, 7\n") \
  X(48, "v_and_b32 v, 0xff(inline 63), v", "v_and_b32 %0, 63, %0\n v_and_b32 %1, 63, %1\n v_and_b32 %2, 63, %2\n v_and_b32 %3, 63, %3\n v_and_b32 %4, 63, %4\n v_and_b32 %5, 63, %5\n v_and_b32 %6, 63, %6\n v_and_b32 %7, 63, %7\n") \
  X(49, "v_lshlrev_b32 (vgpr amount)", "v_lshlrev_b32 %0, %4, %0\n v_lshlrev_b32 %1, %4, %1\n v_lshlrev_b32 %2, %4, %2\n v_lshlrev_b32 %3, %4, %3\n v_lshlrev_b32 %4, %0, %4\n v_lshlrev_b32 %5, %0, %5\n v_lshlrev_b32 %6, %0, %6\n v_lshlrev_b32 %7, %0, %7\n") \
  X(50, "v_add_co_u32 / v_addc_co_u32", "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n v_add_co_u32 %4, vcc, %4, %0\n v_addc_co_u32 %5, vcc, %5, %0, vcc\n v_add_co_u32 %6, vcc, %6, %0\n v_addc_co_u32 %7, vcc, %7, %0, vcc\n") \
  X(51, "v_cmp + v_cndmask vcc (per pair)", "v_cmp_eq_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_eq_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc\n v_cmp_eq_u32 vcc, %4, %0\n v_cndmask_b32 %5, %5, %0, vcc\n v_cmp_eq_u32 vcc, %6, %0\n v_cndmask_b32 %7, %7, %0, vcc\n v_cmp_eq_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_eq_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc\n v_cmp_eq_u32 vcc, %4, %0\n v_cndmask_b32 %5, %5, %0, vcc\n v_cmp_eq_u32 vcc, %6, %0\n v_cndmask_b32 %7, %7, %0, vcc\n") \
  X(52, "v_max_u32", "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %4\n v_max_u32 %2, %2, %4\n v_max_u32 %3, %3, %4\n v_max_u32 %4, %4, %0\n v_max_u32 %5, %5, %0\n v_max_u32 %6, %6, %0\n v_max_u32 %7, %7, %0\n") \
  X(53, "v_not_b32", "v_not_b32 %0, %4\n v_not_b32 %1, %5\n v_not_b32 %2, %6\n v_not_b32 %3, %7\n v_not_b32 %4, %0\n v_not_b32 %5, %1\n v_not_b32 %6, %2\n v_not_b32 %7, %3\n") \
  X(54, "v_ashrrev_i32 (vgpr amount)", "v_ashrrev_i32 %0, %4, %0\n v_ashrrev_i32 %1, %4, %1\n v_ashrrev_i32 %2, %4, %2\n v_ashrrev_i32 %3, %4, %3\n v_ashrrev_i32 %4, %0, %4\n v_ashrrev_i32 %5, %0, %5\n v_ashrrev_i32 %6, %0, %6\n v_ashrrev_i32 %7, %0, %7\n") \
  X(55, "v_add_u32 + ds_read_b32 (per pair)", "v_add_u32 %0, %0, %4\n ds_read_b32 %1, %5\n v_add_u32 %2, %2, %4\n ds_read_b32 %3, %5\n v_add_u32 %4, %4, %0\n ds_read_b32 %6, %5\n v_add_u32 %0, %0, %4\n ds_read_b32 %7, %5\n v_add_u32 %2, %2, %4\n ds_read_b32 %1, %5\n v_add_u32 %4, %4, %0\n ds_read_b32 %3, %5\n v_add_u32 %0, %0, %4\n ds_read_b32 %6, %5\n v_add_u32 %2, %2, %4\n ds_read_b32 %7, %5\n s_waitcnt lgkmcnt(0)\n") \
  X(56, "v_add_u32 + s_cbranch (per pair)", "v_add_u32 %0, %0, %4\n s_cbranch_scc1 1f\n1:\n v_add_u32 %1, %1, %4\n s_cbranch_scc0 2f\n2:\n v_add_u32 %2, %2, %4\n s_cbranch_scc1 3f\n3:\n v_add_u32 %3, %3, %4\n s_cbranch_scc0 4f\n4:\n v_add_u32 %4, %4, %0\n s_cbranch_scc1 5f\n5:\n v_add_u32 %5, %5, %0\n s_cbranch_scc0 6f\n6:\n v_add_u32 %6, %6, %0\n s_cbranch_scc1 7f\n7:\n v_add_u32 %7, %7, %0\n s_cbranch_scc0 8f\n8:\n") \
  X(57, "v_readlane_b32 (sgpr lane)", "v_readlane_b32 %12, %0, s30\n v_readlane_b32 %13, %1, s30\n v_readlane_b32 %14, %2, s30\n v_readlane_b32 %15, %3, s30\n v_readlane_b32 %12, %4, s31\n v_readlane_b32 %13, %5, s31\n v_readlane_b32 %14, %6, s31\n v_readlane_b32 %15, %7, s31\n") \
  X(58, "v_readlane + dependent s_and (pair)", "v_readlane_b32 %12, %0, 3\n s_and_b32 %13, %12, 7\n v_readlane_b32 %14, %1, 7\n s_and_b32 %15, %14, 7\n v_readlane_b32 %12, %2, 3\n s_and_b32 %13, %12, 7\n v_readlane_b32 %14, %3, 7\n s_and_b32 %15, %14, 7\n v_readlane_b32 %12, %4, 3\n s_and_b32 %13, %12, 7\n v_readlane_b32 %14, %5, 7\n s_and_b32 %15, %14, 7\n v_readlane_b32 %12, %6, 3\n s_and_b32 %13, %12, 7\n v_readlane_b32 %14, %7, 7\n s_and_b32 %15, %14, 7\n")

template <int KIND>
__global__ void __launch_bounds__(1024) op_kernel(unsigned long long *out, unsigned *sink)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    if (KIND == 55) a5 = (threadIdx.x & 1023u) * 4u;   // (the LDS address of the read kind)
    __syncthreads();
    asm volatile("s_mov_b64 vcc, 0x5555\n\ts_mov_b32 s30, 5\n\ts_mov_b32 s31, 9" : : : "vcc", "s30", "s31");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
#define X(i, name, text) if constexpr (KIND == i) asm volatile(R8(text) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "vcc", "scc");
        OPS(X)
#undef X
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(b0 + b1 + b2 + b3) + s0 + s1 + s2 + s3 + lds[a0 & 4095u] == 0x12345u) sink[0] = 1;
}

template <int KIND>
static void run(const char *name, unsigned long long *d_out, unsigned *d_sink)
{
    printf("%-32s", name);
    for (int w : {1, 4}) {
        hipLaunchKernelGGL(op_kernel<KIND>, dim3(256), dim3(256 * w), 0, 0, d_out, d_sink);
        (void)hipDeviceSynchronize();
        const int nw = 256 * 4 * w;
        std::vector<unsigned long long> h(nw);
        (void)hipMemcpy(h.data(), d_out, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("  w=%d: SIMD %5.2f (first wave %5.2f)", w, (double)h[nw - 1] / ((double)REPS * 64.0 * w), (double)h[0] / ((double)REPS * 64.0));
    }
    printf("\n");
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long *d_out; unsigned *d_sink;
    (void)hipMalloc(&d_out, 8192 * sizeof(unsigned long long));
    (void)hipMalloc(&d_sink, 8);
    printf("cycles per wave-instruction one SIMD spends (1 and 4 waves resident)\n");
#define X(i, name, text) run<i>(name, d_out, d_sink);
    OPS(X)
#undef X
    return 0;
}
