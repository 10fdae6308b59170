// NHWC convolution forward as an im2col-free implicit GEMM on MFMA:
//   y[n][oh][ow][ko] = sum_{kh,kw,ci} x[n][oh*s-p+kh][ow*s-p+kw][ci] * w[ko][kh][kw][ci]
// GEMM view: M = N*OH*OW, N = KO, K = R*S*C (gfx950 kernel: gemm_kernel.h, loader ConvFwdA).
// Replaces aten::convolution (cudnn/MIOpen/mkldnn) for the 53 bias-free convs of ResNet-50
// reached from /root/reference/virtex/modules/visual_backbones.py:68-74.
#include "conv_common.h"

using namespace vtxg;

template <class T>
static int conv_fwd_t(const ConvGeo& g, const void* x, const void* w, void* y, const void* residual,
                      int act, float* stat_parts, const float* stat_shift, int* stat_strips, hipStream_t st) {
    const int M = g.N * g.OH * g.OW, Kd = g.R * g.S * g.C;
    int strips = 0;
    auto mk_a = [&](auto& a) { a.x = (const T*)x; a.g = g; a.rows = M; a.K = Kd; };
    auto mk_b = [&](auto& b) { b.p = (const T*)w; b.ld = Kd; b.rows = g.KO; b.K = Kd; };
    if (stat_parts && sizeof(T) == 2) {
        EpiStore<T, true> ep{(T*)y, g.KO, nullptr, (const T*)residual, g.KO, nullptr, act, 1.f, make_dropout(0.f, 0), M, g.KO};
        ep.stat_parts = stat_parts; ep.stat_shift = stat_shift;
        strips = launch_auto<T, ConvFwdA, PlainKC>(mk_a, mk_b, ep, M, g.KO, Kd, 1, st);
    } else {
        EpiStore<T> ep{(T*)y, g.KO, nullptr, (const T*)residual, g.KO, nullptr, act, 1.f, make_dropout(0.f, 0), M, g.KO};
        launch_auto<T, ConvFwdA, PlainKC>(mk_a, mk_b, ep, M, g.KO, Kd, 1, st);
    }
    if (stat_strips) *stat_strips = strips;
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_conv2d_fwd(int dtype, int N, int H, int W, int C, int KO, int R, int S, int stride,
                              int pad, const void* x, const void* w, void* y, float* bn_parts,
                              const float* bn_shift, int* bn_strips, void* stream) {
    VTX_CHECK(x && w && y, VTX_ERR_ARG, "conv2d_fwd: null pointer");
    ConvGeo g;
    int rc = make_geo("conv2d_fwd", dtype, N, H, W, C, KO, R, S, stride, pad, &g);
    if (rc) return rc;
    if (bn_strips) *bn_strips = 0;
    if (dtype == VTX_BF16) return conv_fwd_t<bf16_t>(g, x, w, y, nullptr, ACT_NONE, bn_parts, bn_shift, bn_strips, (hipStream_t)stream);
    return conv_fwd_t<float>(g, x, w, y, nullptr, ACT_NONE, bn_parts, bn_shift, bn_strips, (hipStream_t)stream);
}
