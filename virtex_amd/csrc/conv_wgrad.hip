// NHWC convolution weight-gradient as a split-K implicit GEMM on MFMA:
//   dw[ko][kh][kw][ci] += sum_{n,oh,ow} dy[n][oh][ow][ko] * x[n][oh*s-p+kh][ow*s-p+kw][ci]
// GEMM view: M = KO, N = R*S*C, K = N*OH*OW (both operands pixel-major -> MC loaders);
// split-K slices write plain partial tiles to a workspace that a reduce kernel adds into the fp32
// gradient (which is ACCUMULATED, like autograd's `+=`); a single slice read-modify-writes it directly.
// (fp32 atomics cost ~50 us per 16 MB pass on MI355X -- measured, tools/ablate_tn.py.)  Replaces the weight half of aten::convolution_backward.
#include "conv_common.h"

using namespace vtxg;

int vtx_conv3x3_wgrad_try(int N, int H, int W, int C, int KO, int R, int S, int stride, int pad, const void* x, const void* dy,
                          float* dw, float* ws, long ws_floats, hipStream_t st);

template <class T>
static int conv_wgrad_t(const ConvGeo& g, const void* x, const void* dy, float* dw, int split_k, float* ws,
                        long ws_floats, hipStream_t st) {
    const int M = g.KO, Nd = g.R * g.S * g.C, Kd = g.N * g.OH * g.OW;
    const int bk = 4 * Elem<T>::VEC, nkt = vtx_cdiv(Kd, bk);
    if (split_k <= 0) split_k = vtx_pick_split_k(M, Nd, Kd, bk, ws ? ws_floats : 0, 1);
    else { if (split_k > nkt) split_k = nkt; split_k = vtx_cdiv(nkt, vtx_cdiv(nkt, split_k)); }
    VTX_CHECK(split_k == 1 || (ws && (long)split_k * M * Nd <= ws_floats), VTX_ERR_WORKSPACE,
              "conv2d_wgrad: split_k=%d needs %ld workspace floats", split_k, (long)split_k * M * Nd);
    EpiStore<float> ep = vtx_splitk_epilogue(dw, Nd, 1.f, M, Nd, split_k, ws);
    launch_auto<T, PlainMC, ConvWgradB>(
        [&](auto& a) { a.p = (const T*)dy; a.ld = g.KO; a.rows = M; a.K = Kd; },
        [&](auto& b) { b.x = (const T*)x; b.g = g; b.rows = Nd; b.K = Kd; }, ep, M, Nd, Kd, split_k, st);
    if (split_k > 1) vtx_splitk_reduce(ws, split_k, M, Nd, dw, Nd, st);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_conv2d_wgrad(int dtype, int N, int H, int W, int C, int KO, int R, int S, int stride,
                                int pad, const void* x, const void* dy, float* dw, int split_k, float* workspace,
                                long workspace_floats, void* stream) {
    VTX_CHECK(x && dy && dw, VTX_ERR_ARG, "conv2d_wgrad: null pointer");
    ConvGeo g;
    int rc = make_geo("conv2d_wgrad", dtype, N, H, W, C, KO, R, S, stride, pad, &g);
    if (rc) return rc;
    workspace = vtx_splitk_region(workspace, workspace ? workspace_floats : 0, &workspace_floats, (hipStream_t)stream);   // inside a reduction batch: the free part
    if (dtype == VTX_BF16 && split_k <= 0) {             // 3x3 / stride 1: the streaming kernel (conv3x3_wgrad.hip)
        const int r = vtx_conv3x3_wgrad_try(N, H, W, C, KO, R, S, stride, pad, x, dy, dw, workspace, workspace_floats, (hipStream_t)stream);
        if (r < 0) return r;
        if (r > 0) return VTX_OK;
    }
    if (dtype == VTX_BF16) return conv_wgrad_t<bf16_t>(g, x, dy, dw, split_k, workspace, workspace_floats, (hipStream_t)stream);
    return conv_wgrad_t<float>(g, x, dy, dw, split_k, workspace, workspace_floats, (hipStream_t)stream);
}
