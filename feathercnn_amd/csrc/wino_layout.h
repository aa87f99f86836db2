// wino_layout.h -- where element (xi, row r, column p) of the Winograd scratch tensors V (rows = input channels) and M (rows = output
// channels) lives (round 4).
//
//   whole rows (rounds 1-3):   [64 xi][rows][Pp]                 -- BP = Pp, one column block
//   column blocks of BP:       [Pp / BP][64 xi][rows][BP]        -- BP a power of two that the GEMM's column tile divides
//
// In the blocked form a GEMM B tile (16 channels x BN columns of one xi) is ONE contiguous run of memory when BN = BP (4 KB at BP = 64), a
// GEMM output tile (128 rows x 64 columns) one run of 32 KB, and everything a transform block of 64 tiles touches -- 64 xi x its channels --
// falls inside one window of 64 * rows * BP floats (1 MB at 64 channels) instead of 64 pieces a whole xi plane (rows * Pp floats: 11.8 MB for
// VGG-16's conv1_2 at batch 32) apart.  The reference's own packing [C][T/4][16][4][4] (avx/winograd_kernels_F63.cpp:377) is the CPU analogue.
// One address function for both forms, so every kernel is written once: offset = (p >> shift) * blk + xi * xis + r * bp + (p & mask).
#pragma once

#include <stddef.h>

namespace fhip
{

struct WinoLayout
{
    int shift;     // log2(BP); 31 for the whole-row form (p >> 31 == 0 for every column)
    unsigned mask; // BP - 1; 0x7fffffff for the whole-row form
    int bp;        // row pitch: BP (whole rows: Pp)
    size_t blk;    // floats per column block = (frequency points) * rows * BP (whole rows: unused, 0)
    size_t xis;    // floats from xi to xi + 1 inside a block = rows * bp
    // offset of column p in row 0 of xi 0
    __host__ __device__ __forceinline__ size_t col(int p) const { return (size_t)((unsigned)p >> shift) * blk + ((unsigned)p & mask); }
};

// BP >= Pp (or 0) -> whole rows; nxi = frequency points of the transform (64 for F(6,3), 36 for F(4,3))
inline WinoLayout wino_layout(int rows, int Pp, int BP, int nxi = 64)
{
    WinoLayout L;
    if (BP <= 0 || BP >= Pp)
    {
        L.shift = 31;
        L.mask = 0x7fffffffu;
        L.bp = Pp;
        L.blk = 0;
    }
    else
    {
        int s = 0;
        while ((1 << s) < BP) ++s;
        L.shift = s;
        L.mask = (unsigned)BP - 1u;
        L.bp = BP;
        L.blk = (size_t)nxi * rows * BP;
    }
    L.xis = (size_t)rows * L.bp;
    return L;
}

} // namespace fhip
