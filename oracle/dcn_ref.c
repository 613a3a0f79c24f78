/* Second, independent CPU restatement of mmcv-full 1.4.8 modulated_deform_conv2d
 * (TEST INFRASTRUCTURE -- never linked into the product library).
 *
 * Scalar loops in the order of mmcv's modulated_deformable_im2col kernel: one column
 * element per (channel, batch, out-y, out-x), 9 taps each, 4-corner bilinear gather with
 * zero outside the image, times mask; then a plain GEMM with the [Co, C*K] weight.
 * Anchored on the reference call site model/modules/feat_prop.py:55-58 (stride 1, pad 1,
 * dilation 1, groups 1, deform_groups 16, 3x3), but written for general stride/pad/dil.
 * Layouts are the reference's: NCHW activations, offset [N, dg*2*K, Ho, Wo] with
 * (dy,dx) interleaved per tap, mask [N, dg*K, Ho, Wo], weight [Co, C, kh, kw].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float bilinear_zero(const float *im, int H, int W, float h, float w) {
    int h_low = (int)floorf(h), w_low = (int)floorf(w);
    int h_high = h_low + 1, w_high = w_low + 1;
    float lh = h - h_low, lw = w - w_low, hh = 1.f - lh, hw = 1.f - lw;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if (h_low >= 0 && w_low >= 0) v1 = im[h_low * W + w_low];
    if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * W + w_high];
    if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * W + w_low];
    if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * W + w_high];
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

/* returns 0 on success */
int dcn_ref_forward(const float *x, const float *offset, const float *mask,
                    const float *weight, const float *bias, float *out,
                    int N, int C, int H, int W, int Co, int kh, int kw,
                    int stride, int pad, int dil, int dg) {
    const int K = kh * kw;
    const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    const int cg = C / dg;
    const size_t P = (size_t)Ho * Wo;
    float *col = (float *)malloc(sizeof(float) * (size_t)C * K * P);
    if (!col) return 1;
    for (int n = 0; n < N; ++n) {
        const float *xn = x + (size_t)n * C * H * W;
        const float *on = offset + (size_t)n * dg * 2 * K * P;
        const float *mn = mask + (size_t)n * dg * K * P;
        for (int c = 0; c < C; ++c) {
            const int g = c / cg;
            for (int i = 0; i < kh; ++i)
                for (int j = 0; j < kw; ++j) {
                    const int k = i * kw + j;
                    for (int y = 0; y < Ho; ++y)
                        for (int xo = 0; xo < Wo; ++xo) {
                            const size_t p = (size_t)y * Wo + xo;
                            const float dy = on[((size_t)(g * 2 * K + 2 * k)) * P + p];
                            const float dx = on[((size_t)(g * 2 * K + 2 * k + 1)) * P + p];
                            const float m = mn[((size_t)(g * K + k)) * P + p];
                            const float h_im = y * stride - pad + i * dil + dy;
                            const float w_im = xo * stride - pad + j * dil + dx;
                            float v = 0.f;
                            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
                                v = bilinear_zero(xn + (size_t)c * H * W, H, W, h_im, w_im);
                            col[((size_t)c * K + k) * P + p] = v * m;
                        }
                }
        }
        float *outn = out + (size_t)n * Co * P;
        for (int o = 0; o < Co; ++o) {
            float *orow = outn + (size_t)o * P;
            const float b = bias ? bias[o] : 0.f;
            for (size_t p = 0; p < P; ++p) orow[p] = b;
            const float *wrow = weight + (size_t)o * C * K;
            for (int ck = 0; ck < C * K; ++ck) {
                const float wv = wrow[ck];
                const float *crow = col + (size_t)ck * P;
                for (size_t p = 0; p < P; ++p) orow[p] += wv * crow[p];
            }
        }
    }
    free(col);
    return 0;
}
