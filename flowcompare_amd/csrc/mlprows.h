// Row-resident coupling MLP chain (mlprows.hip): parameters of one launch.
#pragma once
#include "common.h"

namespace fc {

constexpr int MR_HID = 512;            // hidden width the kernel is built for (every shipped coupling net: hidden_dims 512 x n)
constexpr int MR_NB = MR_HID / 32;     // blocks of 32 output features per layer
constexpr int MR_MAXL = 8;             // layers per chain (in_layer + up to 7 hidden layers)

struct MlpRowsLayer {
    const unsigned short* Wf;          // fragment-major fp16 limb image [MR_NB][ks][2][64][8] (PackedLinear::Wf)
    const float* bias;                 // [512]
    const float* colvec;               // layer 0 only: rank-1 extra-context column, or null
    int ks;                            // k16 steps of the image (layer 0: K_pad / 16 rounded up to 16 / 24 / 32; hidden layers: 32)
    int res;                           // hbuf index of the residual (nets.py odd hidden layer) or -1
    int in, out;                       // hbuf indices of the input (layers >= 1) and of the output (all but the last layer)
};

struct MlpRowsParams {
    const float* A[3]; int lda[3]; int segk[3];     // layer 0 input: up to three fp32 segments side by side, k16 steps per segment
    const float* rowscal;                           // [rows] extra context per point (with L[0].colvec) or null
    int nlayers;
    MlpRowsLayer L[MR_MAXL];
    unsigned short* hbuf[3];                        // three rotating fragment-major activation images [rows / 32][32][2][64][8] fp16
    unsigned short* out16;                          // last layer's output: row-major limb image [rows][32][hi 16 | lo' 16] (GemmEpi::A16)
    int* ovf;                                       // split-fp16 range flag (common.h Fp16Guard)
    float out_s1, out_s2;                           // the last layer's limb split: hi = rn16(v s1), lo = rn16((v s1 - hi) s2): (1, 2048) or the one-accumulator form (kOneAccActScale, 1)
    float range_limit;                              // |activation| at or beyond this raises the flag (65504 / out_s1)
    int rows_valid;
    unsigned long long* stamps;                     // diagnostic knob 20 = 4 (in-kernel phase stamps), else null
};

}  // namespace fc
