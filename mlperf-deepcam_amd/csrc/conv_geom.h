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

struct GatherGeom {
  int os, is;      // output / input stride of the phase grid
  int ntaps;
  Tap taps[9];
  int Hin, Win, Cin;    // gathered tensor extents
  int Hout, Wout, Cout; // produced tensor extents
  int Qh, Qw;           // phase grid extents (Hout/os, Wout/os)
};

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
  return true;
}

}  // namespace dc
