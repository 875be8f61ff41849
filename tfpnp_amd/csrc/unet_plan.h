// Arena plan of the UNet activations (shared by the forward and backward orchestration).
#pragma once
#include "common.h"

namespace pnpx {

// ----------------------------------------------------------------------------------------- arena plan
struct Act {              // one activation tensor in the arena
  size_t off = 0;         // byte offset
  int C = 0, H = 0, W = 0;
};
struct UNetPlan {
  // per level l (resolution H>>l, W>>l, channels 32<<l)
  Act in0;                // network input (2 channels; 16 in HS mode, the upper 14 zero) @ level 0
  Act a[5], b[5];         // ConvBlock temporaries of the encoder blocks
  Act da[4], db[4];       // ... and of the decoder blocks (kept apart so a backward pass finds both sets)
  Act x[5];               // encoder outputs x1..x5 (skips)
  Act p[5];               // pooled inputs of level l (l >= 1): channels 16<<l
  Act u[4];               // upsampled decoder inputs at level l (l <= 3): channels 64<<l
  Act y[4];               // decoder outputs at level l (l <= 3)
  size_t total = 0;       // bytes, for capB images
  // conv_mode 0, r6: scratch of the K-split Winograd launches (conv3x3_wino8.hip KSPLIT): the partial-sum slab, per image
  size_t ks_part = 0;
  size_t ks_part_per_image = 0;      // bytes
};
size_t conv3x3_wino8_ksplit_bytes_per_image();

inline size_t act_bytes_per_image(int mode, int C, int h, int w) {
  if (mode == CONV_HS) return (size_t)((C + 7) / 8) * (h + 2) * (w + 2) * 32;
  return (size_t)C * padded_h(h) * padded_w(w) * sizeof(float);
}

inline UNetPlan make_plan(int mode, int capB, int H, int W) {
  UNetPlan P;
  size_t off = 0;
  auto add = [&](Act& d, int C, int h, int w) {
    d.off = off;
    d.C = C;
    d.H = h;
    d.W = w;
    off += act_bytes_per_image(mode, C, h, w) * (size_t)capB;
    off = (off + 255) & ~(size_t)255;
  };
  add(P.in0, mode == CONV_HS ? 16 : 2, H, W);
  for (int l = 0; l < 5; ++l) {
    const int h = H >> l, w = W >> l, c = 32 << l;
    add(P.a[l], c, h, w);
    add(P.b[l], c, h, w);
    add(P.x[l], c, h, w);
    if (l >= 1) add(P.p[l], c / 2, h, w);
    if (l <= 3) {
      add(P.da[l], c, h, w);
      add(P.db[l], c, h, w);
      add(P.u[l], 2 * c, h, w);
      add(P.y[l], c, h, w);
    }
  }
  off += 1u << 20;             // 1 MiB slack: overhanging tiles read (never write) past their tensor
  if (mode != CONV_HS) {
    P.ks_part_per_image = conv3x3_wino8_ksplit_bytes_per_image();
    P.ks_part = off;
    off += P.ks_part_per_image * (size_t)capB;
  }
  P.total = off;
  return P;
}


}  // namespace pnpx
