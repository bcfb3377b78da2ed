// Host-side geometry shared by the dense-conv kernels: every dense op (conv forward, conv data-gradient,
// transposed conv forward, transposed conv data-gradient) is written in GATHER FORM
//
//     out[n, qy*os + py, qx*os + px, :] = sum over taps t of phase (py,px):  W[t.widx] . in[n, qy*is + t.dy, qx*is + t.dx, :]
//
// so one MFMA kernel covers all of them; the phase split (os = 2) is the sub-pixel decomposition of a
// stride-2 transposed convolution (1/2/2/4 taps per output parity class), which avoids multiplying zeros.
#pragma once
#include "common.h"

namespace dc {

struct Tap {
  int dy, dx, widx, phase;
};

// Division by a launch-invariant divisor without the ~35-instruction integer division sequence: q = mulhi(n, M) >> sh, exact
// for 0 <= n < 2^31 (M = ceil(2^(31+s) / d), s = ceil(log2 d), sh = s - 1; d == 1 is flagged by M == 0).
struct FastDiv {
  unsigned M;
  int sh;
  int d;
};
inline FastDiv make_fastdiv(int d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) { f.M = 0; f.sh = 0; return f; }
  int s = 0;
  while ((1L << s) < d) ++s;
  f.M = (unsigned)(((1UL << (31 + s)) + (unsigned long)d - 1) / (unsigned long)d);
  f.sh = s - 1;
  return f;
}
__device__ inline int fast_div(int n, const FastDiv& f) { return f.M == 0 ? n : (int)(__umulhi((unsigned)n, f.M) >> f.sh); }

struct GatherGeom {
  int os, is;      // output / input stride of the phase grid
  int ntaps;
  Tap taps[9];     // sorted by phase
  int phase_beg[5];     // taps of phase ph: [phase_beg[ph], phase_beg[ph+1])
  FastDiv div_hw, div_w;   // by Qh*Qw and by Qw
  int Hin, Win, Cin;    // gathered tensor extents
  int Hout, Wout, Cout; // produced tensor extents
  int Qh, Qw;           // phase grid extents (Hout/os, Wout/os)
};

// pixel index m of the phase grid (0 <= m < N*Qh*Qw) -> image n, grid position (qy, qx)
__device__ inline void grid_pixel(const GatherGeom& g, int m, int& n, int& qy, int& qx) {
  n = fast_div(m, g.div_hw);
  const int rem = m - n * (g.Qh * g.Qw);
  qy = fast_div(rem, g.div_w);
  qx = rem - qy * g.Qw;
}

enum GatherMode { kFwd = 0, kDgrad = 1 };

// Builds the tap table.  Hi, Wi are ALWAYS the forward-input extents of the layer.
// Returns false when the geometry is unsupported (odd extents under a stride-2 phase split).
inline bool build_geom(const dc_conv_desc& d, int Hi, int Wi, GatherMode mode, GatherGeom* g) {
  int Ho, Wo;
  const int k = d.transposed ? 3 : d.k;
  if (d.transposed) {
    Ho = 2 * Hi;
    Wo = 2 * Wi;
  } else {
    Ho = (Hi + 2 * d.pad - d.dil * (k - 1) - 1) / d.stride + 1;
    Wo = (Wi + 2 * d.pad - d.dil * (k - 1) - 1) / d.stride + 1;
  }
  g->ntaps = 0;
  // "scatter" relation o = i*s - p + kk*dil, written from the side that is being produced
  const int s = d.transposed ? 2 : d.stride;
  const int p = d.transposed ? 1 : d.pad;
  const int dil = d.transposed ? 1 : d.dil;
  // A layer is "direct" when the produced tensor is on the o side of  o*s' ... ; two cases:
  //  direct  (conv fwd, convT dgrad): produced index q, gathered index q*s - p + kk*dil
  //  inverse (conv dgrad, convT fwd): produced index i with gathered o = (i + p - kk*dil)/s when divisible
  const bool direct = (d.transposed != 0) == (mode == kDgrad);
  if (direct) {
    g->os = 1;
    g->is = s;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) g->taps[g->ntaps++] = Tap{ky * dil - p, kx * dil - p, ky * k + kx, 0};
  } else {
    g->os = s;
    g->is = 1;
    for (int py = 0; py < s; ++py)
      for (int px = 0; px < s; ++px)
        for (int ky = 0; ky < k; ++ky)
          for (int kx = 0; kx < k; ++kx) {
            const int ay = py + p - ky * dil, ax = px + p - kx * dil;
            if (ay % s != 0 || ax % s != 0) continue;
            g->taps[g->ntaps++] = Tap{ay / s, ax / s, ky * k + kx, py * s + px};
          }
  }
  const bool produces_fwd_out = (mode == kFwd);
  if (produces_fwd_out) {
    g->Hin = Hi; g->Win = Wi; g->Cin = d.cin;
    g->Hout = Ho; g->Wout = Wo; g->Cout = d.cout;
  } else {
    g->Hin = Ho; g->Win = Wo; g->Cin = d.cout;
    g->Hout = Hi; g->Wout = Wi; g->Cout = d.cin;
  }
  if (g->Hout % g->os != 0 || g->Wout % g->os != 0) return false;
  g->Qh = g->Hout / g->os;
  g->Qw = g->Wout / g->os;
  for (int ph = 0; ph <= 4; ++ph) {
    int c = 0;
    while (c < g->ntaps && g->taps[c].phase < ph) ++c;   // taps are generated in phase order
    g->phase_beg[ph] = c;
  }
  g->div_hw = make_fastdiv(g->Qh * g->Qw);
  g->div_w = make_fastdiv(g->Qw);
  return true;
}

}  // namespace dc
