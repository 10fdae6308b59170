// NHWC convolution weight-gradient as a split-K implicit GEMM on MFMA:
//   dw[ko][kh][kw][ci] += sum_{n,oh,ow} dy[n][oh][ow][ko] * x[n][oh*s-p+kh][ow*s-p+kw][ci]
// GEMM view: M = KO, N = R*S*C, K = N*OH*OW (both operands pixel-major -> MC loaders);
// split-K partial sums are combined with fp32 atomics into the fp32 gradient (which is
// ACCUMULATED, like autograd's `+=`).  Replaces the weight half of aten::convolution_backward.
#include "conv_common.h"

using namespace vtxg;

template <class T>
static int conv_wgrad_t(const ConvGeo& g, const void* x, const void* dy, float* dw, int split_k, hipStream_t st) {
    const int M = g.KO, Nd = g.R * g.S * g.C, Kd = g.N * g.OH * g.OW;
    if (split_k <= 0) split_k = vtx_pick_split_k(M, Nd, Kd, 4 * Elem<T>::VEC);
    EpiAtomic ep{dw, Nd, 1.f, M, Nd};
    launch_auto<T, PlainMC, ConvWgradB>(
        [&](auto& a) { a.p = (const T*)dy; a.ld = g.KO; a.rows = M; a.K = Kd; },
        [&](auto& b) { b.x = (const T*)x; b.g = g; b.rows = Nd; b.K = Kd; }, ep, M, Nd, Kd, split_k, st);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_conv2d_wgrad(int dtype, int N, int H, int W, int C, int KO, int R, int S, int stride,
                                int pad, const void* x, const void* dy, float* dw, int split_k, void* stream) {
    VTX_CHECK(x && dy && dw, VTX_ERR_ARG, "conv2d_wgrad: null pointer");
    ConvGeo g;
    int rc = make_geo("conv2d_wgrad", dtype, N, H, W, C, KO, R, S, stride, pad, &g);
    if (rc) return rc;
    if (dtype == VTX_BF16) return conv_wgrad_t<bf16_t>(g, x, dy, dw, split_k, (hipStream_t)stream);
    return conv_wgrad_t<float>(g, x, dy, dw, split_k, (hipStream_t)stream);
}
