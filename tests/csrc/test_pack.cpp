// Host emulation of one wavefront executing the swapped MFMA GEMM exactly as kernels.hip indexes it
// (weights = A operand in the packed fragment order of pack.h, NHWC activations = B operand, accumulator
// register r of lane l = D[cout = (l>>4)*4 + r][pixel = l&15]) and of the 3x3 implicit-GEMM k -> (tap, channel)
// mapping, against a direct convolution.  Validates pack.h's index math for both MMA shapes
// (fp16: K=32, KPL=8; fp32: K=4, KPL=1) and the XCD remap's bijectivity.  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#include "../../retinaface_amd/csrc/pack.h"

using namespace rf;

static float frand() { return (float)(rand() % 2001 - 1000) / 500.f; }

// one 16(cout) x 16(pixel) x K MFMA on operands laid out per lane
static void mfma_emul(const std::vector<float> &a_frag /*[64][KPL]*/, const std::vector<float> &b_frag, float acc[64][4], int KPL) {
    const int kgroups = 4;
    for (int lane = 0; lane < 64; lane++)
        for (int r = 0; r < 4; r++) {
            int row = (lane >> 4) * 4 + r, col = lane & 15;
            float s = 0;
            for (int g = 0; g < kgroups; g++)
                for (int e = 0; e < KPL; e++) s += a_frag[(g * 16 + row) * KPL + e] * b_frag[(g * 16 + col) * KPL + e];
            acc[lane][r] += s;
        }
}

static int test_conv(int K, int KPL, int cin, int cout, int ksz) {
    const int TH = 4, TW = 8, P = TH * TW, HC = TW + 2, HR = TH + 2;
    const int ktot = ksz * ksz * cin, kch = k_chunks_for(ktot, K);
    std::vector<float> w((size_t)cout * ktot), x((size_t)HR * HC * cin);
    for (auto &v : w) v = frand();
    for (auto &v : x) v = frand();
    std::vector<float> packed((size_t)(cout / 16) * kch * 64 * KPL, 0.f);
    for (int o = 0; o < cout; o++)
        for (int k = 0; k < ktot; k++) packed[packed_weight_index(o, k, kch, K, KPL)] = w[(size_t)o * ktot + k];
    int bad = 0;
    for (int ct = 0; ct < cout / 16; ct++)
        for (int pt = 0; pt < P / 16; pt++) {
            float acc[64][4] = {};
            for (int kc = 0; kc < kch; kc++) {
                std::vector<float> af((size_t)64 * KPL), bf((size_t)64 * KPL, 0.f);
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < KPL; e++) af[lane * KPL + e] = packed[((size_t)(ct * kch + kc) * 64 + lane) * KPL + e];
                for (int lane = 0; lane < 64; lane++) {
                    int kb = kc * K + (lane >> 4) * KPL;
                    if (kb >= ktot) continue;
                    int p = acc_pixel(pt, lane), py = p / TW, px = p % TW;
                    int tap = kb / cin, c = kb % cin;
                    int ky = ksz == 3 ? tap / 3 : 1, kx = ksz == 3 ? tap % 3 : 1;     // 1x1: centre of the halo tile
                    for (int e = 0; e < KPL; e++) bf[lane * KPL + e] = x[((size_t)(py + ky) * HC + px + kx) * cin + c + e];
                }
                mfma_emul(af, bf, acc, KPL);
            }
            for (int lane = 0; lane < 64; lane++)
                for (int r = 0; r < 4; r++) {
                    int o = acc_cout(ct, lane, r), p = acc_pixel(pt, lane), py = p / TW, px = p % TW;
                    double ref = 0;
                    for (int ky = 0; ky < ksz; ky++)
                        for (int kx = 0; kx < ksz; kx++)
                            for (int c = 0; c < cin; c++) {
                                int yy = ksz == 3 ? py + ky : py + 1, xx = ksz == 3 ? px + kx : px + 1;
                                ref += (double)w[(size_t)o * ktot + (ky * ksz + kx) * cin + c] * x[((size_t)yy * HC + xx) * cin + c];
                            }
                    if (std::fabs(ref - acc[lane][r]) > 1e-3 * (1 + std::fabs(ref))) bad++;
                }
        }
    if (bad) printf("FAIL K=%d KPL=%d cin=%d cout=%d k=%d: %d mismatches\n", K, KPL, cin, cout, ksz, bad);
    return bad;
}

int main() {
    int bad = 0;
    for (int shape = 0; shape < 2; shape++) {
        int K = shape ? 4 : 32, KPL = shape ? 1 : 8;
        bad += test_conv(K, KPL, 8, 16, 1);
        bad += test_conv(K, KPL, 64, 32, 1);
        bad += test_conv(K, KPL, 256, 64, 1);
        bad += test_conv(K, KPL, 16, 16, 3);     // K chunk straddles taps, tail chunk half empty (fp16)
        bad += test_conv(K, KPL, 16, 32, 3);
        bad += test_conv(K, KPL, 64, 48, 3);
    }
    for (int nblk : {1, 7, 8, 9, 15, 64, 100, 3136}) {
        std::vector<int> seen(nblk, 0);
        for (int b = 0; b < nblk; b++) {
            int m = xcd_remap(b, nblk);
            if (m < 0 || m >= nblk || seen[m]++) { printf("FAIL xcd_remap nblk=%d\n", nblk); bad++; break; }
        }
    }
    // depthwise-on-MFMA: expanding the per-lane dword (dw_mma_dword / dw_mma_dword_index) must reproduce the A fragment of the
    // diagonal 16 x 144 matrix  A[c'][tap*16 + c] = (c == c') ? w[tap][c'] : 0  under the 16x16x32 A-operand layout
    {
        uint16_t w[16][9];
        for (int c = 0; c < 16; c++)
            for (int t = 0; t < 9; t++) w[c][t] = (uint16_t)(0x3c00 + 37 * c + t);        // distinct non-zero bit patterns
        for (int kc = 0; kc < kDwMmaChunks; kc++)
            for (int lane = 0; lane < 64; lane++) {
                uint32_t dword = dw_mma_dword(kc, lane, w[lane & 15]);
                uint16_t frag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int d = dw_mma_dword_index(lane);
                frag[2 * d] = (uint16_t)(dword & 0xffff);
                frag[2 * d + 1] = (uint16_t)(dword >> 16);
                for (int e = 0; e < 8; e++) {
                    int row = lane & 15, k = kc * 32 + (lane >> 4) * 8 + e, tap = k / 16, c = k % 16;
                    uint16_t want = (tap < 9 && c == row) ? w[row][tap] : 0;
                    if (frag[e] != want) { if (!bad) printf("FAIL dw_mma kc=%d lane=%d e=%d\n", kc, lane, e); bad++; }
                }
            }
    }
    // int8 variant: 16 x 144 diagonal under the 16x16x64 i8 A-operand layout (lane: row l & 15, k = (l >> 4)*16 + e), taps as
    // 128*hi + lo; and the 15-bit split itself
    {
        int8_t w[16][9];
        for (int c = 0; c < 16; c++)
            for (int t = 0; t < 9; t++) w[c][t] = (int8_t)(1 + 7 * c + t - 60);
        for (int kc = 0; kc < kDwMmaChunksI8; kc++)
            for (int lane = 0; lane < 64; lane++) {
                uint32_t dword = dw_mma_dword_i8(kc, lane, w[lane & 15]);
                int8_t frag[16] = {0};
                int d = dw_mma_dword_index_i8(lane);
                for (int b = 0; b < 4; b++) frag[4 * d + b] = (int8_t)((dword >> (8 * b)) & 0xff);
                for (int e = 0; e < 16; e++) {
                    int row = lane & 15, k = kc * 64 + (lane >> 4) * 16 + e, tap = k / 16, c = k % 16;
                    int8_t want = (tap < 9 && c == row) ? w[row][tap] : 0;
                    if (frag[e] != want) { if (!bad) printf("FAIL dw_mma_i8 kc=%d lane=%d e=%d\n", kc, lane, e); bad++; }
                }
            }
        for (int wi = -kDwI8Range; wi <= kDwI8Range; wi++) {
            int8_t hi, lo;
            dw_i8_split(wi, &hi, &lo);
            if (128 * (int)hi + (int)lo != wi || hi > 127 || hi < -127 || lo > 64 || lo < -64) { if (!bad) printf("FAIL dw_i8_split %d\n", wi); bad++; }
        }
    }
    // persistent tile walk: advancing by the decomposed step must equal decoding t + G from scratch, for awkward shapes
    for (int tiles_x : {1, 2, 7, 11, 38}) for (int tiles_y : {1, 3, 7, 28}) for (int n : {1, 3, 128}) {
        const int tiles = tiles_x * tiles_y * n;
        for (int g : {1, 2, 5, 7, tiles_x, tiles_x * tiles_y, tiles_x * tiles_y + 1, 717, 1792, tiles}) {
            if (g < 1 || g > tiles) continue;
            TileStep step(g, tiles_x, tiles_y);
            for (int first : {0, g / 2, g - 1}) {
                TileCoord c(first, tiles_x, tiles_y);
                for (int t = first; t < tiles; t += g) {
                    TileCoord want(t, tiles_x, tiles_y);
                    if (c.tx != want.tx || c.ty != want.ty || c.img != want.img) { if (!bad) printf("FAIL tile walk %dx%dx%d g=%d t=%d\n", tiles_x, tiles_y, n, g, t); bad++; break; }
                    step.advance(c);
                }
            }
        }
    }
    // persistent grid: never more workgroups than tiles or than resident, and every workgroup walks the same number of tiles +-1
    for (int tiles : {1, 8, 767, 768, 769, 896, 1536, 3584, 12544, 25088}) for (int resident : {256, 768, 1792}) {
        for (float mr : {1.0f, 1.5f}) {
            const int g = persistent_grid_size(tiles, resident, mr);
            const int lo = tiles / g, hi = (tiles + g - 1) / g;
            const bool one_each = (float)tiles <= mr * (float)resident;
            if (g < 1 || g > tiles || (!one_each && g > resident) || (one_each && g != tiles) || hi - lo > 1) { printf("FAIL persistent_grid_size(%d, %d, %.1f) = %d\n", tiles, resident, mr, g); bad++; }
        }
    }
    // LDS rows: 16-byte aligned, at least as wide as asked, 32 mod 64 bytes
    if (lds_row<unsigned short>(64) != 80 || lds_row<unsigned short>(16) != 16 || lds_row<unsigned short>(32) != 48 || lds_row<float>(64) != 72 ||
        lds_row<signed char>(64) != 96 || lds_row<unsigned short>(48) != 48) { printf("FAIL lds_row\n"); bad++; }
    // conv0 pixel tables of the stems' raw-row staging (round 6; measured-and-rejected in the kernels, kept with the probe build): every real pixel of the
    // region appears exactly once in the lanes that hold real channels, its LDS window lies inside the staged rows, its result slot inside the conv0 tile
    for (int which = 0; which < 2; which++) {
        const int r0h = which ? kStemR0H : kStem2R0H, r0w = which ? kStemR0W : kStem2R0W, plane = which ? kStemC0Plane : kStem2C0Plane;
        const int row_bytes = which ? 256 : 128, rows = 2 * r0h + 1, tiles = which ? kStemC0Tiles : kStem2C0Tiles;
        const std::vector<uint32_t> t = which ? stem_conv0_table() : stem2_conv0_table();
        std::vector<int> seen(r0h * r0w, 0);
        for (int wave = 0; wave < 4; wave++) for (int k = 0; k < tiles; k++) for (int lane = 0; lane < 64; lane++) {
            const uint32_t x = t[(((size_t)wave * tiles + k) * 64 + lane) * 2], y = t[(((size_t)wave * tiles + k) * 64 + lane) * 2 + 1];
            const uint32_t p1 = x & 0xffffu, out = x >> 16, hy = y & 0xffu, hx = (y >> 8) & 0xffu, valid = y >> 16;
            const int kb = lane >> 4;
            if (p1 % 4 || p1 + 8 + (uint32_t)(2 - (kb >> 1)) * row_bytes > (uint32_t)(rows + 5) * row_bytes) { printf("FAIL conv0 table %d: window out of the patch\n", which); bad++; }
            if (valid) {
                if (lane >= 32 || hy >= (uint32_t)r0h || hx >= (uint32_t)r0w || (hx & 1) != (uint32_t)(wave >> 1)) { printf("FAIL conv0 table %d: bad pixel\n", which); bad++; continue; }
                if (out != (uint32_t)((kb * plane + (hy * r0w + hx) * 4) * 4) || out + 16 > (uint32_t)(2 * plane * 4)) { printf("FAIL conv0 table %d: bad result slot\n", which); bad++; }
                if (kb == 0) seen[hy * r0w + hx]++;
            }
        }
        for (int v : seen) if (v != 1) { printf("FAIL conv0 table %d: a pixel is covered %d times\n", which, v); bad++; break; }
    }
    printf(bad ? "test_pack: FAILED\n" : "test_pack: ok\n");
    return bad ? 1 : 0;
}
