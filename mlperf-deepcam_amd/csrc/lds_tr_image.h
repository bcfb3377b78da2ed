// The [pixel][channel] LDS image that feeds MFMA fragments through ds_read_b64_tr_b16 (wgrad384.hip's layout): quads of 64 channels,
// [quad][32 pixel rows][128 B], the four 32-byte chunks (16 channels each) of a row XOR-ed with (row >> 1) & 3, so that a half-wave's transposing
// read touches 8 consecutive rows x 32 B = all 64 banks once.
#pragma once
#include "common.h"

namespace dc {

constexpr int TRI_QUAD = 32 * 128;     // bytes of one quad (64 channels x 32 pixels)

typedef __attribute__((address_space(3))) short4v* tri_lds_short4;

// byte offset of (pixel row r, 16-byte channel group g8 of the quad) inside a quad
__device__ inline int tri_slot(int r, int g8) { return r * 128 + ((((g8 >> 1) ^ (r >> 1)) & 3) << 5) + (g8 & 1) * 16; }

// lane-constant part of a fragment address: lane (fg = lane >> 4, fr = lane & 15) supplies pixel rows 4 fg + (fr >> 2) (and + 16), columns
// 4 (fr & 3) .. of a 16-channel chunk; the chunk s of the quad adds ((s ^ tri_key(lane)) << 5)
__device__ inline int tri_base(int lane) { return (4 * (lane >> 4) + ((lane & 15) >> 2)) * 128 + 8 * (lane & 3); }
__device__ inline int tri_key(int lane) { return ((4 * (lane >> 4) + ((lane & 15) >> 2)) >> 1) & 3; }

// one MFMA operand fragment (16 channels x 32 pixels, channel-per-lane): two transposing reads, pixels {4 fg ..} and {16 + 4 fg ..} (2 KiB apart).
// EXEC must be all ones.
__device__ inline bf16x8 tri_frag(const char* p) {
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tri_lds_short4)(p));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tri_lds_short4)(p + 16 * 128));
  typedef __attribute__((ext_vector_type(8))) short short8v;
  short8v f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return __builtin_bit_cast(bf16x8, f);
}

}  // namespace dc
