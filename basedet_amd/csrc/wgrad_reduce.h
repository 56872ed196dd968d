// Reduce descriptors shared by the weight-gradient translation units (conv_wgrad.hip holds the kernel; the ring kernels fill entries).
#pragma once
#include <stdint.h>

constexpr int BD_RED_MAX = 40;          // entries per launch (kernel argument block: 40 x 88 B + 8 B < 4 KB)

struct BdRedEntry {
    const float* slab;
    float* dw;
    const float* row_scale;
    int kind;               // 0: slab[split][n] in the result's order; 1: 3x3 ring register rows; 2: 1x1 ring register rows
    int splits, accumulate;
    int n4;                 // f32x4 elements per split
    int block_begin;        // first block of this entry inside the launch (filled by the launcher)
    int row_len;            // kind 0: floats per output channel (row scale index = float index / row_len)
    int regs, co_tiles, tci, tco, fi, fj, Cin, Cout;       // kinds 1 / 2
};

struct BdRedBatch {
    int count, pad_;
    BdRedEntry e[BD_RED_MAX];
};
