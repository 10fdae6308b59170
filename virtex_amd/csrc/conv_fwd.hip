// NHWC convolution forward as an im2col-free implicit GEMM on MFMA:
//   y[n][oh][ow][ko] = sum_{kh,kw,ci} x[n][oh*s-p+kh][ow*s-p+kw][ci] * w[ko][kh][kw][ci]
// GEMM view: M = N*OH*OW, N = KO, K = R*S*C (gfx950 kernel: gemm_kernel.h, loader ConvFwdA).
// Replaces aten::convolution (cudnn/MIOpen/mkldnn) for the 53 bias-free convs of ResNet-50
// reached from /root/reference/virtex/modules/visual_backbones.py:68-74.
#include "conv_common.h"
#include "conv3x3_kernel.h"

using namespace vtxg;

int vtx_stem_stream_try(int N, int H, int W, int C, int KO, int R, int S, int stride, int pad, const void* x, const void* w,
                        void* y, const float* shift, float* parts, hipStream_t st);

template <class T>
static int conv_fwd_t(const ConvGeo& g, const void* x, const void* w, void* y, const float* bias, const void* residual,
                      int act, float* stat_parts, const float* stat_shift, int* stat_strips, hipStream_t st) {
    const int M = g.N * g.OH * g.OW, Kd = g.R * g.S * g.C;
    int strips = 0;
    auto mk_a = [&](auto& a) { a.x = (const T*)x; a.g = g; a.rows = M; a.K = Kd; };
    auto mk_b = [&](auto& b) { b.p = (const T*)w; b.ld = Kd; b.rows = g.KO; b.K = Kd; };
    bool done = false;
    // 3x3 / stride 1 / pad 1 (bf16): the kernel whose A tile serves the three taps of a filter row (conv3x3_kernel.h)
    const bool c3 = sizeof(T) == 2 && g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g_vtx_contraction_generation >= 2 &&
                    g_vtx_tile_override < 0;
    if constexpr (sizeof(T) == 2) {
        if (stat_parts) {
            EpiStore<T, STATS_FWD> ep{(T*)y, g.KO, bias, (const T*)residual, g.KO, nullptr, act, 1.f, make_dropout(0.f, 0), M, g.KO};
            ep.stat_parts = stat_parts; ep.stat_shift = stat_shift;
            if (c3) strips = conv3x3_shared_try(x, g.N, g.H, g.W, g.C, g.KO, mk_b, ep, 0, st);
            if (strips == 0) strips = launch_auto<T, ConvFwdA, PlainKC>(mk_a, mk_b, ep, M, g.KO, Kd, 1, st);
            done = true;
        }
    }
    if (!done) {
        EpiStore<T> ep{(T*)y, g.KO, bias, (const T*)residual, g.KO, nullptr, act, 1.f, make_dropout(0.f, 0), M, g.KO};
        int taken = 0;
        if constexpr (sizeof(T) == 2) { if (c3) taken = conv3x3_shared_try(x, g.N, g.H, g.W, g.C, g.KO, mk_b, ep, 0, st); }
        if (!taken) launch_auto<T, ConvFwdA, PlainKC>(mk_a, mk_b, ep, M, g.KO, Kd, 1, st);
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
    if (dtype == VTX_BF16 && bn_parts && bn_strips) {      // the packed stem: streaming kernel (stem.hip)
        const int s = vtx_stem_stream_try(N, H, W, C, KO, R, S, stride, pad, x, w, y, bn_shift, bn_parts, (hipStream_t)stream);
        if (s < 0) return s;
        if (s > 0) { *bn_strips = s; return VTX_OK; }
    }
    if (dtype == VTX_BF16) return conv_fwd_t<bf16_t>(g, x, w, y, nullptr, nullptr, ACT_NONE, bn_parts, bn_shift, bn_strips, (hipStream_t)stream);
    return conv_fwd_t<float>(g, x, w, y, nullptr, nullptr, ACT_NONE, bn_parts, bn_shift, bn_strips, (hipStream_t)stream);
}

template <class T>
static int pointwise_t(int M, int KO, int C, const void* x, const void* w, void* y, const float* bias,
                       const void* residual, int act, hipStream_t st) {
    EpiStore<T> ep{(T*)y, KO, bias, (const T*)residual, KO, nullptr, act, 1.f, make_dropout(0.f, 0), M, KO};
    launch_auto<T, PlainKC, PlainKC>([&](auto& a) { a.p = (const T*)x; a.ld = C; a.rows = M; a.K = C; },
                                     [&](auto& b) { b.p = (const T*)w; b.ld = C; b.rows = KO; b.K = C; }, ep, M, KO, C, 1, st);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

// Inference convolution with the (folded) BatchNorm, the residual add and the ReLU in the epilogue:
//   y = act(conv(x, w) + bias[ko] (+ residual)),  relu: 0 none, 1 ReLU.
// w/bias come from vtx_bn_fold.  This is the eval-mode / frozen backbone (reference: `frozen=True` puts
// the CNN in eval mode, visual_backbones.py:49-53; downstream feature extraction scripts/clf_voc07.py:165-200).
extern "C" int vtx_conv2d_infer(int dtype, int N, int H, int W, int C, int KO, int R, int S, int stride,
                                int pad, const void* x, const void* w, const float* bias, const void* residual,
                                int relu, void* y, void* stream) {
    VTX_CHECK(x && w && y, VTX_ERR_ARG, "conv2d_infer: null pointer");
    ConvGeo g;
    int rc = make_geo("conv2d_infer", dtype, N, H, W, C, KO, R, S, stride, pad, &g);
    if (rc) return rc;
    const int act = relu ? (residual ? ACT_RES_RELU : ACT_RELU) : ACT_NONE;
    if (R == 1 && S == 1 && stride == 1 && pad == 0) {   // pointwise: a plain [N*H*W][C] x [KO][C]^T GEMM
        const int M = N * H * W;
        if (dtype == VTX_BF16) return pointwise_t<bf16_t>(M, KO, C, x, w, y, bias, residual, act, (hipStream_t)stream);
        return pointwise_t<float>(M, KO, C, x, w, y, bias, residual, act, (hipStream_t)stream);
    }
    if (dtype == VTX_BF16) return conv_fwd_t<bf16_t>(g, x, w, y, bias, residual, act, nullptr, nullptr, nullptr, (hipStream_t)stream);
    return conv_fwd_t<float>(g, x, w, y, bias, residual, act, nullptr, nullptr, nullptr, (hipStream_t)stream);
}
