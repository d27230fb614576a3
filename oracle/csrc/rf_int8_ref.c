/* rf_int8_ref.c -- plain-C half of the integer-exact int8 oracle (oracle/int8_forward.py).  TEST INFRASTRUCTURE ONLY.
 *
 * The int8 engine of the reference is TensorRT 5.1's (trtnetbase.cpp:295-311: setInt8Mode + the calibration cache read by
 * trtnetbase.cpp:31-44); its kernels are closed source, so there is no reference arithmetic to restate bit for bit.  What IS
 * pinned by the reference is the calibration table (model/mnet-deconv-0517.table.int8) and the symmetric int8 scheme TensorRT
 * documents (real ~= q * scale, q in [-127, 127], per-output-channel weight scales).  This file restates, independently of the HIP
 * code, the arithmetic the repo's int8 engine defines on top of that:
 *
 *     acc  = sum_k  w_q[c][k] * x_q[k]                     exact int32
 *     y    = fmaf((float)acc, mult[c], bias[c])            ONE rounding (libm fmaf is correctly rounded)
 *     q    = (int8) rint(clamp(y, 0, 127))                 ReLU'd outputs; round half to even
 *          = (int8) clamp(rint(y), -127, 127)              outputs without ReLU
 *          = (int8) (rint(clamp(y, 0, 255)) - 128)         depthwise outputs ("mids", round 6): ReLU'd quanta 0..255 of amax / 255, stored
 *                                                          minus 128; the pointwise bias carries mult * 128 * sum_k w_q (fmaf, rfi8_bias_u8)
 *
 * Built by oracle/build.py with -ffp-contract=off: the only fused operation is the explicit fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* acc[n_pix][c] (int32) -> q[n_pix][c] (int8) */
void rfi8_requant(const int32_t *acc, const float *mult, const float *bias, long n_pix, int c, int relu, int8_t *q) {
    for (long p = 0; p < n_pix; p++)
        for (int ch = 0; ch < c; ch++) {
            const float y = fmaf((float)acc[p * c + ch], mult[ch], bias[ch]);
            float r;
            if (relu == 2) r = rintf(fminf(fmaxf(y, 0.f), 255.f)) - 128.f;
            else if (relu) r = rintf(fminf(fmaxf(y, 0.f), 127.f));
            else r = fminf(fmaxf(rintf(y), -127.f), 127.f);
            q[p * c + ch] = (int8_t)(int)r;
        }
}

/* bias of a 1x1 conv whose input is a mid stored as q - 128: bias[c] = fmaf(mult[c], (float)(128 * qsum[c]), bias[c]) */
void rfi8_bias_u8(const float *mult, const long *qsum, int c, float *bias) {
    for (int ch = 0; ch < c; ch++) bias[ch] = fmaf(mult[ch], (float)(128 * qsum[ch]), bias[ch]);
}

/* heads: real-valued outputs (no requantisation), y = fmaf(acc, mult, bias) */
void rfi8_dequant(const int32_t *acc, const float *mult, const float *bias, long n_pix, int c, float *y) {
    for (long p = 0; p < n_pix; p++)
        for (int ch = 0; ch < c; ch++) y[p * c + ch] = fmaf((float)acc[p * c + ch], mult[ch], bias[ch]);
}

/* Integer convolution, NHWC int8 input, weights int32 [cout][k][k][cin/group] (int8 values, or the 15-bit depthwise taps), zero
 * padding, -> int32 NHWC.  The slow, obviously-correct second back-end of the oracle (the first is an exact float64 BLAS conv). */
void rfi8_conv(const int8_t *x, int h, int w, int cin, const int32_t *wt, int cout, int k, int stride, int pad, int group,
               int32_t *out) {
    const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    const int cg = cin / group, og = cout / group;
    for (int oy = 0; oy < ho; oy++)
        for (int ox = 0; ox < wo; ox++)
            for (int o = 0; o < cout; o++) {
                const int g = o / og;
                int32_t acc = 0;
                for (int ky = 0; ky < k; ky++) {
                    const int iy = oy * stride - pad + ky;
                    if (iy < 0 || iy >= h) continue;
                    for (int kx = 0; kx < k; kx++) {
                        const int ix = ox * stride - pad + kx;
                        if (ix < 0 || ix >= w) continue;
                        const int8_t *xp = x + ((long)iy * w + ix) * cin + g * cg;
                        const int32_t *wp = wt + (((long)o * k + ky) * k + kx) * cg;
                        for (int c = 0; c < cg; c++) acc += wp[c] * (int32_t)xp[c];
                    }
                }
                out[((long)oy * wo + ox) * cout + o] = acc;
            }
}

/* Fused "lateral + bilinear x2 upsample(coarser)" staging of the aggregation convs (Deconvolution k4 s2 p1 g64 with the fixed
 * bilinear kernel + Crop + Eltwise SUM, prototxt :1553-1592 / :1948-1987), requantised to the `_plus` tensor's scale.
 *   out[y][x] = .5625 up[my][mx] + .1875 up[my][mx2] + .1875 up[my2][mx] + .0625 up[my2][mx2] + lat[y][x]
 *   my = y >> 1, my2 = my + 1 (y odd) or my - 1 (y even), same for x; coarse taps outside the coarse map contribute 0.
 * mode 0 (per-tensor table: the three tensors have different scales):  q = clamp(rint(fmaf(lat, a_lat, blend * a_up)), -127, 127),
 *        the blend accumulated in fp32 in the order above (every partial sum is exact: small integers times dyadic weights)
 * mode 1 (per-channel table, one common scale): q = min(rne(lat + (9a + 3b + 3c + d) / 16), 127) in integers */
void rfi8_upadd(const int8_t *lat, const int8_t *up, int h, int w, int c, float a_lat, float a_up, int mode, int8_t *out) {
    const int hh = h >> 1, wh = w >> 1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int my = y >> 1, mx = x >> 1;
            const int my2 = (y & 1) ? my + 1 : my - 1, mx2 = (x & 1) ? mx + 1 : mx - 1;
            const int ys[4] = {my, my, my2, my2}, xs[4] = {mx, mx2, mx, mx2};
            for (int ch = 0; ch < c; ch++) {
                int t[4];
                for (int q = 0; q < 4; q++)
                    t[q] = (ys[q] >= 0 && ys[q] < hh && xs[q] >= 0 && xs[q] < wh) ? up[((long)ys[q] * wh + xs[q]) * c + ch] : 0;
                const int l = lat[((long)y * w + x) * c + ch];
                int r;
                if (mode == 1) {
                    const int s16 = 9 * t[0] + 3 * t[1] + 3 * t[2] + t[3];          /* 16 x the blend, >= 0 */
                    const int tot = 16 * l + s16;                                   /* 16 x (lat + blend)   */
                    int qv = tot >> 4;
                    const int rem = tot & 15;
                    if (rem > 8 || (rem == 8 && (qv & 1))) qv++;                    /* round half to even   */
                    r = qv > 127 ? 127 : qv;
                } else {
                    float s = 0.5625f * (float)t[0];
                    s = fmaf(0.1875f, (float)t[1], s);
                    s = fmaf(0.1875f, (float)t[2], s);
                    s = fmaf(0.0625f, (float)t[3], s);
                    const float v = fmaf((float)l, a_lat, s * a_up);
                    r = (int)fminf(fmaxf(rintf(v), -127.f), 127.f);
                }
                out[((long)y * w + x) * c + ch] = (int8_t)r;
            }
        }
}
