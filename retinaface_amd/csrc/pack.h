// pack.h -- index math shared by the host packers (engine.cpp), the HIP kernels and the host unit
// tests (tests/csrc/test_pack.cpp): where element W[cout][k] of a GEMM-shaped weight lives in the
// MFMA-fragment-ordered buffer, and which (cout, pixel) a lane's accumulator register holds.
//
// All GEMMs in this engine are computed "swapped": D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],
// weights as the MFMA A operand (rows = output channels), NHWC activations as the B operand
// (columns = pixels).  With the gfx950 16x16 layouts (cdna_hip_programming.md section 3):
//   A operand: lane l holds A[row = l & 15][k = (l >> 4) * KPL + e], e < KPL
//   B operand: lane l holds B[k = (l >> 4) * KPL + e][col = l & 15]
//   C / D    : lane l, register r holds D[row = (l >> 4) * 4 + r][col = l & 15]
// so a lane ends up with 4 *consecutive output channels* of one pixel -- an 8-byte (fp16) NHWC store --
// and the B fragment is KPL consecutive channels of one pixel -- a single ds_read_b128 from an NHWC tile.
//   fp16: v_mfma_f32_16x16x32_f16  K = 32, KPL = 8      fp32: v_mfma_f32_16x16x4_f32  K = 4, KPL = 1
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#if defined(__HIPCC__)
#define RF_HD __host__ __device__
#else
#define RF_HD
#endif

namespace rf {

// Packed weight buffer: [cout_tile = cout/16][k_chunk = k/K][lane 0..63][KPL] elements.
// One (cout_tile, k_chunk) = one wave-wide A fragment = 64 * KPL contiguous elements, so a wave loads it with
// one fully coalesced 16-byte-per-lane (fp16) global load.
RF_HD inline size_t packed_weight_index(int cout, int k, int k_chunks, int K, int KPL) {
    int ct = cout >> 4, row = cout & 15;
    int kc = k / K, kk = k % K;
    int lane = (kk / KPL) * 16 + row;
    return ((size_t)(ct * k_chunks + kc) * 64 + lane) * KPL + (kk % KPL);
}

RF_HD inline int k_chunks_for(int k_total, int K) { return (k_total + K - 1) / K; }

// accumulator register r of lane l in tile (ct, pt): which output channel / which pixel of the block tile
RF_HD inline int acc_cout(int ct, int lane, int r) { return ct * 16 + (lane >> 4) * 4 + r; }
RF_HD inline int acc_pixel(int pt, int lane) { return pt * 16 + (lane & 15); }

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed, speed only).
// Give every XCD one contiguous range of logical tile ids so that (a) neighbouring tiles, which share
// halo rows, and (b) the same image's tiles in consecutive layers land on the same XCD and hit its 4 MiB L2.
// Bijective for any nblk (cdna_hip_programming.md section 5, "XCD swizzle must be bijective").
RF_HD inline int xcd_remap(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7;
    int xcd = bid & 7, i = bid >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + i;
}

// Depthwise 3x3 on the matrix cores (fp16 engine): for a group of 16 channels the stencil is a dense 3x3 conv 16 -> 16 whose
// weight matrix is diagonal, K = 9 taps x 16 channels = 144 -> 5 MFMA K-chunks of 32 (k = tap*16 + c, like K_c).  Lane l of
// A fragment (group g, chunk kc) holds row c' = l & 15, columns k = kc*32 + (l >> 4)*8 + e, e < 8: at most ONE of its 8
// halves is non-zero (c == c'), so the fragment is rebuilt in registers from one dword per lane: the fp16 weight bits,
// already shifted into the half it occupies; the dword index inside the 4-dword fragment is ((c' & 7) >> 1) for every kc.
// Returns that dword for lane `lane` given w9[tap] = the 9 fp16 weight bit patterns of channel g*16 + (lane & 15).
RF_HD inline uint32_t dw_mma_dword(int kc, int lane, const uint16_t *w9) {
    const int c = lane & 15, kgrp = lane >> 4;
    const int tap = kc * 2 + (kgrp >> 1);
    if (tap >= 9 || (c >> 3) != (kgrp & 1)) return 0u;       // this lane's 8 columns do not contain channel c
    return (c & 1) ? ((uint32_t)w9[tap] << 16) : (uint32_t)w9[tap];
}
RF_HD inline int dw_mma_dword_index(int lane) { return ((lane & 15) & 7) >> 1; }
constexpr int kDwMmaChunks = 5;

// The same for the int8 engine (v_mfma_i32_16x16x64_i8, K = 64 = 4 taps x 16 channels per chunk, 144 -> 3 chunks): lane l of A
// fragment (group g, chunk kc) holds row c' = l & 15, columns k = (l >> 4)*16 + e, e < 16, i.e. tap kc*4 + (l >> 4), all 16
// channels: one non-zero BYTE (e == c') -> dword (c' >> 2) of the 4-dword fragment, byte (c' & 3).  The taps are 15-bit
// integers w = 128*hi + lo carried by TWO such fragments (hi, lo in [-127, 127]): acc = 128*acc_hi + acc_lo, so the depthwise
// weights lose nothing to 8-bit quantisation (the round-1 engine ran this stencil in fp32 on the VALU for that reason).
RF_HD inline uint32_t dw_mma_dword_i8(int kc, int lane, const int8_t *w9) {
    const int c = lane & 15, tap = kc * 4 + (lane >> 4);
    if (tap >= 9) return 0u;
    return (uint32_t)(uint8_t)w9[tap] << (8 * (c & 3));
}
RF_HD inline int dw_mma_dword_index_i8(int lane) { return (lane & 15) >> 2; }
constexpr int kDwMmaChunksI8 = 3;
// w (real, in output quanta per input quantum) -> 15-bit integer split; scale = amax / kDwI8Range per channel
constexpr int kDwI8Range = 127 * 128;
RF_HD inline void dw_i8_split(int w_int, int8_t *hi, int8_t *lo) {
    int h = (w_int >= 0 ? w_int + 64 : w_int - 64) / 128;          // round to nearest multiple of 128
    if (h > 127) h = 127;
    if (h < -127) h = -127;
    *hi = (int8_t)h;
    *lo = (int8_t)(w_int - 128 * h);                               // [-64, 64] (up to +-127 only when h saturates; it cannot: |w| <= 16256)
}

// stem2's conv0 phase with raw-row staging (kernels.hip stem2_kernel, round 6): which conv0 pixel a lane of a wave's k-th MFMA tile computes, where its
// bytes lie in the staged patch and where its result goes -- the same for every workgroup, so it is a TABLE ([4 waves][kStem2C0Tiles][64 lanes] x uint2)
// instead of ~9 VALU instructions of index arithmetic per tile:
//   x = LDS byte offset of the lane's first dword of MFMA 1 (row 2 hy + (kb >> 1), dword 3 m + parity + 2 (kb & 1))  |  (byte offset of its fp32 result in the
//       conv0 tile, plane kb, << 16);   y = hy | hx << 8 | valid << 16   (valid: a real pixel of the 17 x 19 region in a lane that holds real channels)
// Geometry = Stem2Cfg<8>: conv0 region 17 x 19, 10 even / 9 odd columns, tiles of 16 same-parity pixels, waves 0,1 even (tiles w, w + 2, ..), waves 2,3 odd.
constexpr int kStem2R0H = 17, kStem2R0W = 19, kStem2C0Tiles = 6, kStem2C0Plane = ((kStem2R0H * kStem2R0W + 15) / 16) * 16 * 4;      // floats per 4-channel plane
// the int8 engine's stem (stem_kernel): conv0 region 10 x 34 = 17 even + 17 odd columns, 256-byte rows, windows start at dword 1 (even) / 3 (odd)
constexpr int kStemR0H = 10, kStemR0W = 34, kStemC0Tiles = 6, kStemC0Plane = ((kStemR0H * kStemR0W + 15) / 16) * 16 * 4;
inline std::vector<uint32_t> conv0_raw_table(int r0h, int r0w, int tiles_per_wave, int plane, int row_dwords, int d0_even, int d0_odd) {
    std::vector<uint32_t> t((size_t)4 * tiles_per_wave * 64 * 2, 0u);
    for (int wave = 0; wave < 4; wave++) {
        const int par = wave >> 1, wp = par ? r0w / 2 : (r0w + 1) / 2, d0 = par ? d0_odd : d0_even;
        for (int k = 0; k < tiles_per_wave; k++)
            for (int lane = 0; lane < 64; lane++) {
                const int tile = (wave & 1) + 2 * k, kb = lane >> 4;
                const int qq = tile * 16 + (lane & 15), hy = qq / wp, m = qq % wp, hx = 2 * m + par;
                const bool real = hy < r0h;
                // lanes past the region read row 0 (any finite bytes) and write nothing
                const uint32_t p1 = real ? (uint32_t)(((2 * hy + (kb >> 1)) * row_dwords + 3 * m + d0 + 2 * (kb & 1)) * 4) : (uint32_t)((kb & 1) * 8);
                const uint32_t out = real && lane < 32 ? (uint32_t)((kb * plane + (hy * r0w + hx) * 4) * 4) : 0u;
                uint32_t *e = &t[(((size_t)wave * tiles_per_wave + k) * 64 + lane) * 2];
                e[0] = p1 | (out << 16);
                e[1] = (uint32_t)(real ? hy : 0) | ((uint32_t)(real ? hx : 0) << 8) | ((real && lane < 32) ? 1u << 16 : 0u);
            }
    }
    return t;
}
inline std::vector<uint32_t> stem2_conv0_table() { return conv0_raw_table(kStem2R0H, kStem2R0W, kStem2C0Tiles, kStem2C0Plane, 32, 0, 1); }
inline std::vector<uint32_t> stem_conv0_table() { return conv0_raw_table(kStemR0H, kStemR0W, kStemC0Tiles, kStemC0Plane, 64, 1, 3); }

// Stem pointwise (8 -> 16 channels) as ONE v_mfma_f32_16x16x32_f16 with fp32-grade operands: the depthwise result reaches the
// MFMA as an fp16 pair x = x_hi + x_lo (B operand: K group 0 = x_hi, 1 = x_lo, 2 = x_hi, 3 = 0) and the weight as w = w_hi + w_lo
// (A operand: K group 0 = w_hi, 1 = w_hi, 2 = w_lo, 3 = 0), so D = w_hi x_hi + w_hi x_lo + w_lo x_hi (the dropped w_lo x_lo term is
// 2^-22 relative).  Element (lane, e) of the A fragment for weight matrix w[cout 16][cin 8]; hi / lo are passed in as the two
// roundings the caller computed (hi = fp16(w), lo = fp16(w - hi)).
RF_HD inline bool stem_pw_slot(int lane, int *row, int *use_lo) {
    const int kgrp = lane >> 4;
    *row = lane & 15;
    *use_lo = kgrp == 2;
    return kgrp < 3;
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent-kernel bookkeeping shared by the kernels and the host unit test (tests/csrc/test_pack.cpp)
// ---------------------------------------------------------------------------------------------------------------------

// Tile walk of the persistent kernels: tile id t -> (tx, ty, img), advanced by a fixed step G without dividing (a uniform
// integer division is ~16 scalar instructions, and the tile loops are a few hundred instructions per tile).
struct TileCoord {
    int tx, ty, img;
    RF_HD TileCoord(int t, int tiles_x, int tiles_y) : tx(t % tiles_x), ty((t / tiles_x) % tiles_y), img(t / (tiles_x * tiles_y)) {}
};
struct TileStep {
    int sx, sy, si, nx, ny;
    RF_HD TileStep(int g, int tiles_x, int tiles_y)
        : sx(g % tiles_x), sy((g / tiles_x) % tiles_y), si(g / (tiles_x * tiles_y)), nx(tiles_x), ny(tiles_y) {}
    RF_HD inline void advance(TileCoord &c) const {
        c.tx += sx;
        const int cx = c.tx >= nx ? 1 : 0;
        c.tx -= cx ? nx : 0;
        c.ty += sy + cx;
        const int cy = c.ty >= ny ? 1 : 0;
        c.ty -= cy ? ny : 0;
        c.img += si + cy;
    }
};

// Grid of a persistent launch: as many workgroups as the chip keeps resident, trimmed so that every workgroup walks the same
// number of tiles (no nearly-empty last round).  At or below `min_rounds` x resident tiles the hardware dispatcher's dynamic
// one-tile-per-workgroup schedule is at least as good as walking two tiles in sequence: one workgroup per tile.
RF_HD inline int persistent_grid_size(int tiles, int resident, float min_rounds) {
    if (resident < 1) resident = 1;
    if ((float)tiles <= min_rounds * (float)resident) return tiles;
    const int rounds = (tiles + resident - 1) / resident;
    return (tiles + rounds - 1) / rounds;
}

// Image sharding of one detectBatchImages() call over G devices (multi.cpp; the same rule as retinaface_amd/shard.py
// shard_range): contiguous slices of ceil(n / G) images, trailing devices may get fewer or none.  lo[g] .. lo[g + 1].
RF_HD inline int shard_begin(int n, int G, int g) {
    const int per = (n + G - 1) / G;
    const long b = (long)g * per;
    return b < n ? (int)b : n;
}

// Row stride (in elements) of an LDS tile whose rows are read as MFMA B fragments (16 lanes = 16 consecutive pixels, 16 B each,
// 4 such groups one 16-byte column apart): measured on gfx950 (tools/probes/lds_b128.cpp) a ds_read_b128 of that pattern costs
// one replay less when the row stride in bytes is 32 mod 64 (32, 96, 160, 288 ...) than at 16 / 48 mod 64 (48, 80, 144 ...),
// and 128 / 256-byte strides are the worst.  Rows are padded up to the next such stride.
template <typename T> constexpr int lds_row(int c) {
    const int bytes = c * (int)sizeof(T);
    return (bytes + ((32 - bytes % 64) + 64) % 64) / (int)sizeof(T);
}

}  // namespace rf
