// Weight gradient of the two thin 3x3 convolutions of the entry stem (16 -> 32 stride 2 on 768x1152, 32 -> 64 stride 1 on
// 384x576; deeplab_xception.py:145,149).  With so few channels the tiled GEMM kernels (wgrad.hip / wgrad384.hip) launch one
// workgroup set PER TAP, each re-reading dy, and fill 1/8 of their MFMA tile: 0.67-0.68 ms per layer against ~0.07 ms of
// compulsory traffic.  Here ONE pass produces all nine taps:
//
//     dW[t][co][ci] = sum over output pixels m of dy[m][co] * x[S*m - 1 + t][ci]        (M = co, N = (tap, ci), K = pixels)
//
//   * a workgroup walks a strip of output rows, 32 output pixels wide; per step the 32 dy pixels and the S NEW input rows of the
//     (32 S + 2)-pixel halo arrive by LDS-DMA (a ring of input rows keeps the other two rows of the 3x3 window), so both tensors
//     leave HBM once;
//   * rows stay [pixel][channel] in LDS; both MFMA operands are fetched channel-per-lane with ds_read_b64_tr_b16 -- for the x
//     operand the tap is just an address offset (row slot for ky, pixel offset for kx), the stride a multiplier;
//   * 4 waves = 2 halves of co x 2 halves of the (tap, ci) columns; a wave keeps <= 18 accumulator tiles;
//   * one partial [9][Co][Ci] per workgroup goes to the fp32 slab of wgrad.hip ([split][tap][Co][Ci]) and its fixed-order
//     reduction kernel finishes the job (deterministic).
#include "igemm.h"
#include "wgrad.h"

namespace dc {

namespace {

static __device__ __attribute__((aligned(256))) unsigned char thin_zero_page[256];
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(3))) short4v lds_s4;

struct ThinArgs {
  const void* x;
  const void* dy;
  float* slab;
  int N, Hi, Wi, Ho, Wo, ldx, lddy;
  int nseg;        // 32-pixel column segments per output row
  int nchunk;      // row chunks per image
  int rows_per;    // output rows per chunk
};

template <int CIN, int COUT, int S>
struct ThinCfg {
  static constexpr int XP = 32 * S + 2;                 // halo pixels per input row segment
  static constexpr int XROWB = XP * CIN * 2;            // bytes
  static constexpr int XSLOT = (XROWB + 1023) / 1024 * 1024;   // ring slot (whole 1-KiB DMA chunks)
  static constexpr int RX = 4 * S;                      // input-row ring: S+2 live rows + S in flight
  static constexpr int DYB = 32 * COUT * 2;             // one step of dy
  static constexpr int DYSLOT = (DYB + 1023) / 1024 * 1024;
  static constexpr int LDS = RX * XSLOT + 2 * DYSLOT;
  static constexpr int MB = COUT / 16, NBT = CIN / 16, NB = 9 * NBT;
  static constexpr int MBW = MB / 2;                    // co blocks per wave
  static constexpr int NBH = (NB + 1) / 2;              // column blocks of the first wave column (the second gets NB - NBH)
};

template <int CIN, int COUT, int S>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const ThinArgs a) {
  typedef ThinCfg<CIN, COUT, S> K;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xring = smem;
  char* dring = smem + K::RX * K::XSLOT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave >> 1, wb = wave & 1;

  // workgroup -> (image, row chunk, column segment); consecutive workgroups of an XCD take neighbouring column segments
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  int unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + xslot;
  const int seg = unit % a.nseg;
  unit /= a.nseg;
  const int chunk = unit % a.nchunk, n = unit / a.nchunk;
  const int r0 = chunk * a.rows_per, r1 = min(a.Ho, r0 + a.rows_per);
  const int qx0 = seg * 32;
  const bf16* __restrict__ xg = reinterpret_cast<const bf16*>(a.x);
  const bf16* __restrict__ dg = reinterpret_cast<const bf16*>(a.dy);

  // ---- LDS-DMA of one input row segment (iy may be outside the image: zero page) and of one step of dy
  constexpr int XSPP = CIN * 2 / 16;                   // 16-byte slots per x pixel
  constexpr int XINS = (K::XP * XSPP + 63) / 64;       // DMA instructions per input row
  constexpr int DSPP = COUT * 2 / 16;
  constexpr int DINS = 32 * DSPP / 64;
  auto issue_xrow = [&](int iy) {
    char* dst = xring + ((iy + 1 + K::RX * 4) % K::RX) * K::XSLOT;      // slot by input row (iy >= -1)
    const bool yok = (unsigned)iy < (unsigned)a.Hi;
    for (int i = wave; i < XINS; i += 4) {
      const int s = i * 64 + lane;
      const int px = s / XSPP, sub = s % XSPP;
      const int ix = qx0 * S - 1 + px;
      const bool ok = yok && px < K::XP && (unsigned)ix < (unsigned)a.Wi;
      const void* src = ok ? (const void*)(xg + (((size_t)n * a.Hi + iy) * a.Wi + ix) * a.ldx + sub * 8) : (const void*)thin_zero_page;
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(dst + i * 1024), 16, 0, 0);
    }
  };
  auto issue_dy = [&](int q) {
    char* dst = dring + (q & 1) * K::DYSLOT;
    for (int i = wave; i < DINS; i += 4) {
      const int s = i * 64 + lane;
      const int px = s / DSPP, sub = s % DSPP;
      const void* src = (const void*)(dg + (((size_t)n * a.Ho + q) * a.Wo + qx0 + px) * a.lddy + sub * 8);
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(dst + i * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[K::MBW][K::NBH];
#pragma unroll
  for (int i = 0; i < K::MBW; ++i)
#pragma unroll
    for (int j = 0; j < K::NBH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int p0 = 8 * fg + tq;   // this lane's pixel of the first transposed read (the second is p0 + 4)

  // prologue: the three input rows of the first step and its dy
  if (r0 < r1) {
    for (int iy = S * r0 - 1; iy <= S * r0 + 1; ++iy) issue_xrow(iy);
    issue_dy(r0);
  }
  for (int q = r0; q < r1; ++q) {
    __syncthreads();              // vmcnt(0) + barrier: step q has landed, and everybody is done reading step q-1
    if (q + 1 < r1) {             // next step: S new input rows and its dy (their latency hides behind co-resident workgroups)
      for (int iy = S * (q + 1) + 2 - S; iy <= S * (q + 1) + 1; ++iy) issue_xrow(iy);
      issue_dy(q + 1);
    }
    const char* dcur = dring + (q & 1) * K::DYSLOT;
    vec16 fa[K::MBW];
#pragma unroll
    for (int i = 0; i < K::MBW; ++i) {
      const int cblk = wa * K::MBW + i;
      const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dcur + p0 * (COUT * 2) + cblk * 32 + tp * 8));
      const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dcur + (p0 + 4) * (COUT * 2) + cblk * 32 + tp * 8));
      const uint2 t0 = __builtin_bit_cast(uint2, a0), t1 = __builtin_bit_cast(uint2, a1);
      fa[i].w[0] = t0.x; fa[i].w[1] = t0.y; fa[i].w[2] = t1.x; fa[i].w[3] = t1.y;
    }
#pragma unroll
    for (int j = 0; j < K::NBH; ++j) {
      const int nb = wb * K::NBH + j;                 // column block = (tap, 16-channel half)
      if (nb < K::NB) {
        const int t = nb / K::NBT, hb = nb % K::NBT;
        const int ky = t / 3, kx = t % 3;
        const char* xrow = xring + ((S * q + ky + K::RX * 4) % K::RX) * K::XSLOT;   // input row S*q - 1 + ky -> slot (iy + 1) % RX
        const short4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(xrow + (p0 * S + kx) * (CIN * 2) + hb * 32 + tp * 8));
        const short4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(xrow + ((p0 + 4) * S + kx) * (CIN * 2) + hb * 32 + tp * 8));
        const uint2 t0 = __builtin_bit_cast(uint2, b0), t1 = __builtin_bit_cast(uint2, b1);
        vec16 fb;
        fb.w[0] = t0.x; fb.w[1] = t0.y; fb.w[2] = t1.x; fb.w[3] = t1.y;
#pragma unroll
        for (int i = 0; i < K::MBW; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb), acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- partial [tap][co][ci] of this workgroup
  const int fr = lane & 15;
  float* out = a.slab + (size_t)blockIdx.x * 9 * COUT * CIN;
#pragma unroll
  for (int j = 0; j < K::NBH; ++j) {
    const int nb = wb * K::NBH + j;
    if (nb < K::NB) {
      const int t = nb / K::NBT, hb = nb % K::NBT;
      const int ci = hb * 16 + fr;
#pragma unroll
      for (int i = 0; i < K::MBW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = (wa * K::MBW + i) * 16 + fg * 4 + r;
          out[((size_t)t * COUT + co) * CIN + ci] = acc[i][j][r];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward / data gradient of the thin 3x3 convolutions (16 -> 32 stride 2, 32 -> 64, and the data gradient 64 -> 32 of the
// latter) in gather form:  out[m][co] = sum over taps t, channels ci of in[gather(m, t)][ci] * w[t][co][ci].
// The tiled GEMM kernel spends these layers on per-tile overhead (13824 tiles of 9 K steps, a quarter or half of the 128
// columns used): 230-330 us against ~70 us of traffic.  Here the pixel operand needs no LDS at all: with K = (tap, channel)
// a lane's 16-byte load of ITS pixel's channel group is exactly the MFMA fragment layout, consecutive lanes read consecutive
// pixels (NHWC rows), and the 3x3 reuse is served by L1/L2.  The whole weight set (<= 36 KiB) sits in LDS for the life of a
// persistent workgroup.  Weight rows are staged in a permuted order so that a lane ends up with 8 consecutive output
// channels per 16-byte store and the lanes of a pixel complete its row.
struct ThinFwdArgs {
  const void* in;
  const void* w;       // [tap][Cout][ldw]
  void* out;
  float* slab;         // [2][rows][Cout] or null; this kernel writes rows 0 .. gridDim.x-1 (the rest is zeroed by the launcher)
  int N, Hin, Win, Hout, Wout, ldin, ldout, ldw, is, M, slab_rows;
  int tdy[9], tdx[9], twidx[9];
  FastDiv div_hw, div_w;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void thin_fwd_kernel(const ThinFwdArgs a) {
  constexpr int KS = (9 * CIN + 31) / 32;        // K steps of 32: 5 (two taps each, the last half empty) / 9 / 18
  constexpr int MB = COUT / 16;                  // channel blocks
  constexpr int PB = 4;                          // pixel blocks of 16 per wave iteration
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const bf16* __restrict__ xin = reinterpret_cast<const bf16*>(a.in);
  const bf16* __restrict__ wg = reinterpret_cast<const bf16*>(a.w);

  // ---- weight image: wl[(ks*COUT + row)*64 + swizzled slot], row rho of a channel block set holds channel perm(rho)
  for (int i = tid; i < KS * COUT * 4; i += 256) {
    const int sl = i & 3, row = (i >> 2) % COUT, ks = i / (4 * COUT);
    // COUT = 64: perm(16 i + 4 g + r) = 32 (i >> 1) + 8 g + 4 (i & 1) + r;  COUT = 32: perm(16 i + 4 g + r) = 8 g + 4 i + r
    const int bi = row >> 4, g = (row >> 2) & 3, r = row & 3;
    const int co = COUT == 64 ? (bi >> 1) * 32 + g * 8 + (bi & 1) * 4 + r : g * 8 + bi * 4 + r;
    int tap, coff;                                 // this 16-byte slot = 8 channels of one tap
    if (CIN == 16) { tap = 2 * ks + (sl >> 1); coff = (sl & 1) * 8; }
    else if (CIN == 32) { tap = ks; coff = sl * 8; }
    else { tap = ks >> 1; coff = (ks & 1) * 32 + sl * 8; }
    vec16 v = zero16();
    if (tap < 9) v = ldg16(wg + ((size_t)a.twidx[tap] * COUT + co) * a.ldw + coff);
    *reinterpret_cast<vec16*>(smem + (ks * COUT + row) * 64 + ((sl ^ ((row >> 1) & 3)) << 4)) = v;
  }
  __syncthreads();

  // this lane's K slice: which tap(s) and channel offset for each K step
  float ssum[MB * 4], ssq[MB * 4];
#pragma unroll
  for (int c = 0; c < MB * 4; ++c) ssum[c] = ssq[c] = 0.f;
  bf16* __restrict__ yout = reinterpret_cast<bf16*>(a.out);
  const int chunks = (a.M + 255) / 256;
  // Workgroups of one XCD (equal id mod 8) walk ONE contiguous eighth of the pixel chunks together: the rows above and below a chunk
  // are then read by workgroups that share its L2 at about the same time.  (Round-robin over all chunks put the three readers of an
  // input row on three XCDs: 2.4x the compulsory bytes crossed the fabric, profiles/r04_roofline_table_b8.md.)
  int cbeg = 0, cend = chunks, cfirst = blockIdx.x, cstep = gridDim.x;
  if (gridDim.x >= 8) {
    const int xcd = blockIdx.x & 7;
    cbeg = (int)((long)chunks * xcd / 8);
    cend = (int)((long)chunks * (xcd + 1) / 8);
    cfirst = cbeg + (blockIdx.x >> 3);
    cstep = (gridDim.x - xcd + 7) >> 3;          // workgroups on this XCD
  }
  for (int chunk = cfirst; chunk < cend; chunk += cstep) {
    const int mbase = chunk * 256 + wave * 64;
    // pixel coordinates of this lane's pixel in each of the 4 pixel blocks
    int iy0[PB], ix0[PB];
    const bf16* base[PB];
    bool pok[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int m = mbase + j * 16 + fr;
      pok[j] = m < a.M;
      const int mm = pok[j] ? m : 0;
      const int n = fast_div(mm, a.div_hw);
      const int rem = mm - n * (a.Hout * a.Wout);
      const int oy = fast_div(rem, a.div_w), ox = rem - oy * a.Wout;
      iy0[j] = oy * a.is;
      ix0[j] = ox * a.is;
      base[j] = xin + (size_t)n * a.Hin * a.Win * a.ldin;
    }
    f32x4 acc[MB][PB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_step = [&](int ks, vec16 (&fb)[PB]) {
      int tap, coff;
      if (CIN == 16) { tap = 2 * ks + (fg >> 1); coff = (fg & 1) * 8; }
      else if (CIN == 32) { tap = ks; coff = fg * 8; }
      else { tap = ks >> 1; coff = (ks & 1) * 32 + fg * 8; }
      const bool tok = tap < 9;
      const int dy = a.tdy[tok ? tap : 0], dx = a.tdx[tok ? tap : 0];
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const int iy = iy0[j] + dy, ix = ix0[j] + dx;
        const bool ok = tok && pok[j] && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
        fb[j] = ok ? ldg16(base[j] + ((size_t)iy * a.Win + ix) * a.ldin + coff) : zero16();
      }
    };
    vec16 cur[PB], nxt[PB];
    load_step(0, cur);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) load_step(ks + 1, nxt);
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int row = i * 16 + fr;
        const vec16 fa = *reinterpret_cast<const vec16*>(smem + (ks * COUT + row) * 64 + ((fg ^ ((row >> 1) & 3)) << 4));
#pragma unroll
        for (int j = 0; j < PB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, cur[j]), acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < PB; ++j) cur[j] = nxt[j];
    }
    // ---- store: D rows (channels) fg*4 + r of block i, column (pixel) fr.  Vector hh of this lane = channels 32 hh + 8 fg .. +7
    // (COUT = 64) or 8 fg .. +7 (COUT = 32)
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int m = mbase + j * 16 + fr;
      if (m < a.M) {
        bf16* dst = yout + (size_t)m * a.ldout + fg * 8;
#pragma unroll
        for (int hh = 0; hh < MB / 2; ++hh) {
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = acc[2 * hh + (e >> 2)][j][e & 3];
          vec16 v;
          pack(v, f, bf16());
          stg16(dst + 32 * hh, v);
          unpack(v, f, bf16());
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            ssum[8 * hh + e] += f[e];
            ssq[8 * hh + e] = fmaf(f[e], f[e], ssq[8 * hh + e]);
          }
        }
      }
    }
  }
  if (a.slab != nullptr) {
    // fold the 16 pixel lanes, then the 4 waves through LDS; one slab row per workgroup
#pragma unroll
    for (int off = 1; off < 16; off <<= 1)
#pragma unroll
      for (int c = 0; c < MB * 4; ++c) {
        ssum[c] += __shfl_xor(ssum[c], off, 64);
        ssq[c] += __shfl_xor(ssq[c], off, 64);
      }
    __syncthreads();   // the weight image is dead
    float* red = reinterpret_cast<float*>(smem);   // [wave][2][COUT]
    if (fr == 0) {
#pragma unroll
      for (int c = 0; c < MB * 4; ++c) {
        const int ch = 32 * (c >> 3) + fg * 8 + (c & 7);
        red[(wave * 2 + 0) * COUT + ch] = ssum[c];
        red[(wave * 2 + 1) * COUT + ch] = ssq[c];
      }
    }
    __syncthreads();
    if (tid < 2 * COUT) {
      const int which = tid / COUT, ch = tid % COUT;
      const float s = (red[(0 * 2 + which) * COUT + ch] + red[(1 * 2 + which) * COUT + ch]) + (red[(2 * 2 + which) * COUT + ch] + red[(3 * 2 + which) * COUT + ch]);
      a.slab[((size_t)which * a.slab_rows + blockIdx.x) * COUT + ch] = s;
      // the slab has one row per 128 pixels of the layer, this persistent grid fewer workgroups: the rows nobody owns are zeros (two
      // hipMemsetAsync launches per call did this before: 4 x 5 us on every step's forward chain)
      for (int r = blockIdx.x + gridDim.x; r < a.slab_rows; r += gridDim.x) a.slab[((size_t)which * a.slab_rows + r) * COUT + ch] = 0.f;
    }
  }
}

// The stride-1 layers (32 -> 64 forward, its 64 -> 32 data gradient) with the input staged in LDS.  thin_fwd_kernel gathers every
// K-step fragment from global memory: a pixel's 64 / 128 bytes pass through the vector L1 nine times, and with four workgroups per CU a
// tap's 32 KiB working set evicts the previous tap's, so the nine passes are L2 reads at a CU's miss rate (2 GB per launch at 7.7 TB/s:
// 265 us for 340 MB of tensors).  Here a workgroup owns a 2 x 64 output tile: its 4 x 66 pixel halo arrives ONCE by LDS-DMA (zero page for
// the padding), the nine taps are address offsets into it.  A pixel occupies CIN/8 + 1 sixteen-byte slots (the last one a pad that is
// fetched from the zero page): with a pitch of 80 / 144 bytes the sixteen pixels of a ds_read_b128 group fall on sixteen different bank
// quads.  Same weight image, same K order, same MFMA per output element as thin_fwd_kernel: outputs bit-identical.
template <int CIN, int COUT>
struct ThinTile {
  static constexpr int TH = 2, TW = 64;
  static constexpr int SPP = CIN / 8 + 1;            // slots per pixel
  static constexpr int HH = TH + 2, HW = TW + 2;
  static constexpr int ROWS = HW * SPP;              // slots per halo row
  static constexpr int SLOTS = HH * ROWS;
  static constexpr int ITER = (SLOTS + 255) / 256;
  static constexpr int HALO = ITER * 256 * 16;
  static constexpr int KS = (9 * CIN + 31) / 32;
  static constexpr int WIMG = KS * COUT * 64;
  static constexpr int LDS = WIMG + HALO;
  static constexpr int WGS_PER_CU = (160 * 1024) / LDS;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void thin_tile_kernel(const ThinFwdArgs a) {
  typedef ThinTile<CIN, COUT> K;
  constexpr int KS = K::KS, MB = COUT / 16, PB = 2;
  static_assert(CIN == 32 || CIN == 64, "stride-1 layers only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem + K::WIMG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const bf16* __restrict__ xin = reinterpret_cast<const bf16*>(a.in);
  const bf16* __restrict__ wg = reinterpret_cast<const bf16*>(a.w);
  // ---- weight image (thin_fwd_kernel's)
  for (int i = tid; i < KS * COUT * 4; i += 256) {
    const int sl = i & 3, row = (i >> 2) % COUT, ks = i / (4 * COUT);
    const int bi = row >> 4, g = (row >> 2) & 3, r = row & 3;
    const int co = COUT == 64 ? (bi >> 1) * 32 + g * 8 + (bi & 1) * 4 + r : g * 8 + bi * 4 + r;
    int tap, coff;
    if (CIN == 32) { tap = ks; coff = sl * 8; }
    else { tap = ks >> 1; coff = (ks & 1) * 32 + sl * 8; }
    const vec16 v = ldg16(wg + ((size_t)a.twidx[tap] * COUT + co) * a.ldw + coff);
    *reinterpret_cast<vec16*>(smem + (ks * COUT + row) * 64 + ((sl ^ ((row >> 1) & 3)) << 4)) = v;
  }
  float ssum[MB * 4], ssq[MB * 4];
#pragma unroll
  for (int c = 0; c < MB * 4; ++c) ssum[c] = ssq[c] = 0.f;
  bf16* __restrict__ yout = reinterpret_cast<bf16*>(a.out);
  const int ntx = (a.Wout + K::TW - 1) / K::TW, nty = (a.Hout + K::TH - 1) / K::TH;
  const int ntiles = a.N * nty * ntx;
  // one contiguous band of tiles per XCD (see thin_fwd_kernel)
  int cend = ntiles, cfirst = blockIdx.x, cstep = gridDim.x;
  if (gridDim.x >= 8) {
    const int xcd = blockIdx.x & 7;
    const int cbeg = (int)((long)ntiles * xcd / 8);
    cend = (int)((long)ntiles * (xcd + 1) / 8);
    cfirst = cbeg + (blockIdx.x >> 3);
    cstep = (gridDim.x - xcd + 7) >> 3;
  }
  const uintptr_t zp = (uintptr_t)thin_zero_page;
  // this wave's two pixel blocks of 16: block b = row b / 4, columns 16 * (b % 4) ..
  int prow[PB], pcol[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int b = wave * PB + j;
    prow[j] = b >> 2;
    pcol[j] = (b & 3) * 16 + fr;
  }
  for (int tile = cfirst; tile < cend; tile += cstep) {
    const int tx = tile % ntx;
    const int r = tile / ntx;
    const int ty = r % nty, n = r / nty;
    const int y0 = ty * K::TH, x0 = tx * K::TW;
    __syncthreads();                 // the weight image is written / the previous tile's fragments have been read
    const bf16* base = xin + (size_t)n * a.Hin * a.Win * a.ldin;
#pragma unroll
    for (int it = 0; it < K::ITER; ++it) {
      const int slot = it * 256 + tid;
      const int hy = slot / K::ROWS, rem = slot - hy * K::ROWS;
      const int hx = rem / K::SPP, sp = rem - hx * K::SPP;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = slot < K::SLOTS && sp < CIN / 8 && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      const uintptr_t src = ok ? (uintptr_t)(base + ((size_t)iy * a.Win + ix) * a.ldin + sp * 8) : zp;
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(halo + (it * 256 + wave * 64) * 16), 16, 0, 0);
    }
    __syncthreads();                 // vmcnt(0) + barrier: the halo tile has landed
    f32x4 acc[MB][PB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      int tap, coff;
      if (CIN == 32) { tap = ks; coff = fg; }                       // coff: 16-byte slot inside the pixel
      else { tap = ks >> 1; coff = (ks & 1) * 4 + fg; }
      const int dy = a.tdy[tap], dx = a.tdx[tap];
      vec16 fb[PB];
#pragma unroll
      for (int j = 0; j < PB; ++j)
        fb[j] = *reinterpret_cast<const vec16*>(halo + (((prow[j] + 1 + dy) * K::HW + pcol[j] + 1 + dx) * K::SPP + coff) * 16);
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int row = i * 16 + fr;
        const vec16 fa = *reinterpret_cast<const vec16*>(smem + (ks * COUT + row) * 64 + ((fg ^ ((row >> 1) & 3)) << 4));
#pragma unroll
        for (int j = 0; j < PB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      }
    }
    // ---- store + statistics of the stored values (thin_fwd_kernel's epilogue)
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int oy = y0 + prow[j], ox = x0 + pcol[j];
      if (oy < a.Hout && ox < a.Wout) {
        const size_t m = ((size_t)n * a.Hout + oy) * a.Wout + ox;
        bf16* dst = yout + m * a.ldout + fg * 8;
#pragma unroll
        for (int hh = 0; hh < MB / 2; ++hh) {
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = acc[2 * hh + (e >> 2)][j][e & 3];
          vec16 v;
          pack(v, f, bf16());
          stg16(dst + 32 * hh, v);
          unpack(v, f, bf16());
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            ssum[8 * hh + e] += f[e];
            ssq[8 * hh + e] = fmaf(f[e], f[e], ssq[8 * hh + e]);
          }
        }
      }
    }
  }
  if (a.slab != nullptr) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1)
#pragma unroll
      for (int c = 0; c < MB * 4; ++c) {
        ssum[c] += __shfl_xor(ssum[c], off, 64);
        ssq[c] += __shfl_xor(ssq[c], off, 64);
      }
    __syncthreads();   // the weight image is dead
    float* red = reinterpret_cast<float*>(smem);   // [wave][2][COUT]
    if (fr == 0) {
#pragma unroll
      for (int c = 0; c < MB * 4; ++c) {
        const int ch = 32 * (c >> 3) + fg * 8 + (c & 7);
        red[(wave * 2 + 0) * COUT + ch] = ssum[c];
        red[(wave * 2 + 1) * COUT + ch] = ssq[c];
      }
    }
    __syncthreads();
    if (tid < 2 * COUT) {
      const int which = tid / COUT, ch = tid % COUT;
      const float s = (red[(0 * 2 + which) * COUT + ch] + red[(1 * 2 + which) * COUT + ch]) + (red[(2 * 2 + which) * COUT + ch] + red[(3 * 2 + which) * COUT + ch]);
      a.slab[((size_t)which * a.slab_rows + blockIdx.x) * COUT + ch] = s;
      // the slab has one row per 128 pixels of the layer, this persistent grid fewer workgroups: the rows nobody owns are zeros (two
      // hipMemsetAsync launches per call did this before: 4 x 5 us on every step's forward chain)
      for (int r = blockIdx.x + gridDim.x; r < a.slab_rows; r += gridDim.x) a.slab[((size_t)which * a.slab_rows + r) * COUT + ch] = 0.f;
    }
  }
}

struct ThinPlan {
  int nseg, nchunk, rows_per, blocks;
};
static ThinPlan thin_plan(int N, int Ho, int Wo) {
  ThinPlan p;
  p.nseg = Wo / 32;
  int want = 512 / (N * p.nseg);       // ~2 workgroups per CU; their partials cost 9*Co*Ci*4 bytes each
  if (want < 1) want = 1;
  if (want > Ho) want = Ho;
  p.rows_per = cdiv(Ho, want);
  p.nchunk = cdiv(Ho, p.rows_per);
  p.blocks = N * p.nseg * p.nchunk;
  return p;
}

}  // namespace

bool thin_wgrad_eligible(const dc_conv_desc& d, int Hi, int Wi) {
  if (d.dtype != DC_BF16 || d.transposed || d.k != 3 || d.pad != 1 || d.dil != 1) return false;
  const bool stem = d.cin == 16 && d.cout == 32 && d.stride == 2, conv2 = d.cin == 32 && d.cout == 64 && d.stride == 1;
  if (!stem && !conv2) return false;
  const int Wo = (Wi - 1) / d.stride + 1;
  return Wo % 32 == 0 && Hi >= 2 && Wi >= 2;
}

int thin_wgrad_splits(const dc_conv_desc& d, int N, int Hi, int Wi) {
  const int Ho = (Hi - 1) / d.stride + 1, Wo = (Wi - 1) / d.stride + 1;
  return thin_plan(N, Ho, Wo).blocks;
}

// writes thin_wgrad_splits(...) partials [tap][Co][Ci] into the slab; the caller reduces them (wgrad_reduce_kernel)
int launch_thin_wgrad(const dc_conv_desc& d, int N, int Hi, int Wi, const void* x, int ldx, const void* dy, int lddy, float* slab,
                      hipStream_t st) {
  ThinArgs a;
  a.x = x; a.dy = dy; a.slab = slab;
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ho = (Hi - 1) / d.stride + 1; a.Wo = (Wi - 1) / d.stride + 1; a.ldx = ldx; a.lddy = lddy;
  const ThinPlan p = thin_plan(N, a.Ho, a.Wo);
  a.nseg = p.nseg; a.nchunk = p.nchunk; a.rows_per = p.rows_per;
  DC_REQUIRE((long)N * Hi * Wi < (1L << 31), "dc_conv_wgrad: tensor too large for the thin path");
  if (d.cin == 16) {
    typedef ThinCfg<16, 32, 2> K;
    DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_wgrad_kernel<16, 32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS));
    hipLaunchKernelGGL((thin_wgrad_kernel<16, 32, 2>), dim3(p.blocks), dim3(256), K::LDS, st, a);
  } else {
    typedef ThinCfg<32, 64, 1> K;
    DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_wgrad_kernel<32, 64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS));
    hipLaunchKernelGGL((thin_wgrad_kernel<32, 64, 1>), dim3(p.blocks), dim3(256), K::LDS, st, a);
  }
  DC_CHECK_LAUNCH();
  return 0;
}

// Gather-form forward / data gradient of a thin 3x3 convolution (see thin_fwd_kernel).  `slab_rows` is the row count the caller's
// statistics slab was sized for (dc_conv_stat_rows); rows this kernel does not write are zeroed.
static int g_thin_tile = 1;      // tuning switch "thin_tile": 0 = the gather-form kernel for the stride-1 layers too
void thin_set_tile(int v) { g_thin_tile = v ? 1 : 0; }

bool thin_fwd_eligible(const GatherGeom& g, int dtype, int bias, int accumulate, int out32) {
  if (dtype != DC_BF16 || bias || accumulate || out32 || g.os != 1 || g.ntaps != 9) return false;
  return (g.Cin == 16 && g.Cout == 32) || (g.Cin == 32 && g.Cout == 64) || (g.Cin == 64 && g.Cout == 32);
}

int launch_thin_fwd(const GatherGeom& g, int N, const void* in, int ldin, const void* w, int ldw, void* out, int ldout, float* slab,
                    int slab_rows, hipStream_t st) {
  ThinFwdArgs a;
  a.in = in; a.w = w; a.out = out; a.slab = slab;
  a.N = N; a.Hin = g.Hin; a.Win = g.Win; a.Hout = g.Hout; a.Wout = g.Wout; a.ldin = ldin; a.ldout = ldout; a.ldw = ldw; a.is = g.is;
  a.M = N * g.Hout * g.Wout;
  a.slab_rows = slab_rows;
  for (int t = 0; t < 9; ++t) { a.tdy[t] = g.taps[t].dy; a.tdx[t] = g.taps[t].dx; a.twidx[t] = g.taps[t].widx; }
  a.div_hw = make_fastdiv(g.Hout * g.Wout);
  a.div_w = make_fastdiv(g.Wout);
  // stride 1 with every tap one pixel away at most (the 3 x 3 "same" convolution and its data gradient): the LDS-tiled kernel
  bool tiled = g_thin_tile && g.is == 1 && g.Cin != 16;
  for (int t = 0; t < 9; ++t) tiled = tiled && a.tdy[t] >= -1 && a.tdy[t] <= 1 && a.tdx[t] >= -1 && a.tdx[t] <= 1;
  const int chunks = tiled ? N * cdiv(g.Hout, 2) * cdiv(g.Wout, 64) : cdiv(a.M, 256);
  const int wgs = !tiled ? 1024 : 256 * (g.Cin == 32 ? ThinTile<32, 64>::WGS_PER_CU : ThinTile<64, 32>::WGS_PER_CU);
  int grid = chunks < wgs ? chunks : wgs;
  if (slab != nullptr && grid > slab_rows) grid = slab_rows;
#define THIN_FWD(CI, CO)                                                                                                        \
  do {                                                                                                                          \
    constexpr int LDS = ((9 * CI + 31) / 32) * CO * 64;                                                                         \
    DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_fwd_kernel<CI, CO>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
    hipLaunchKernelGGL((thin_fwd_kernel<CI, CO>), dim3(grid), dim3(256), LDS, st, a);                                           \
  } while (0)
#define THIN_TILE(CI, CO)                                                                                                       \
  do {                                                                                                                          \
    constexpr int LDS = ThinTile<CI, CO>::LDS;                                                                                  \
    DC_ONCE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_tile_kernel<CI, CO>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
    hipLaunchKernelGGL((thin_tile_kernel<CI, CO>), dim3(grid), dim3(256), LDS, st, a);                                          \
  } while (0)
  if (g.Cin == 16) THIN_FWD(16, 32);
  else if (g.Cin == 32) { if (tiled) THIN_TILE(32, 64); else THIN_FWD(32, 64); }
  else { if (tiled) THIN_TILE(64, 32); else THIN_FWD(64, 32); }
#undef THIN_TILE
#undef THIN_FWD
  DC_CHECK_LAUNCH();
  return 0;
}

}  // namespace dc
