// NHWC convolution input-gradient as an implicit GEMM on MFMA:
//   dx[n][ih][iw][ci] = sum_{kh,kw,ko} dy[n][(ih+p-kh)/s][(iw+p-kw)/s][ko] * wt[ci][kh][kw][ko]
// (terms exist only where the divisions are exact and in range).  GEMM view: M = N*H*W,
// N = C, K = R*S*KO; `wt` is the weight pre-transposed to [C][R][S][KO] by vtx_weight_prep.
// `residual` (optional, same shape as dx) is added in the epilogue: the gradient join of a
// bottleneck's two branches costs no extra pass.
// Replaces the input-gradient half of aten::convolution_backward.
#include "conv_common.h"
#include "conv3x3_kernel.h"

using namespace vtxg;

void vtx_fill_bn_bwd(vtxg::EpiStore<bf16_t, vtxg::STATS_BWD>& ep, const VtxBnBwdFusion* f, long ld, float* parts);
int vtx_check_bn_bwd(const char* who, const VtxBnBwdFusion* f, int M, int N);

template <class T>
static int conv_dgrad_t(const ConvGeo& g, const void* dy, const void* wt, void* dx, const void* residual, hipStream_t st) {
    const int M = g.N * g.H * g.W, Kd = g.R * g.S * g.KO;
    EpiStore<T> ep{(T*)dx, g.C, nullptr, (const T*)residual, g.C, nullptr, ACT_NONE, 1.f, make_dropout(0.f, 0), M, g.C};
    auto mk_b = [&](auto& b) { b.p = (const T*)wt; b.ld = Kd; b.rows = g.C; b.K = Kd; };
    if constexpr (sizeof(T) == 2) {      // 3x3 / stride 1 / pad 1: the taps of dy slide like a forward convolution's, flipped
        if (g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g_vtx_contraction_generation >= 2 && g_vtx_tile_override < 0 &&
            conv3x3_shared_try(dy, g.N, g.H, g.W, g.KO, g.C, mk_b, ep, 1, st)) {
            VTX_LAUNCH_CHECK();
            return VTX_OK;
        }
    }
    launch_auto<T, ConvDgradA, PlainKC>(
        [&](auto& a) { a.dy = (const T*)dy; a.g = g; a.rows = M; a.K = Kd; }, mk_b, ep, M, g.C, Kd, 1, st);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

// stride 2, even H and W: four dense parity-class GEMMs (no zero taps), rows scattered in the epilogue
template <class T>
static int conv_dgrad_s2_t(const ConvGeo& g, const void* dy, const void* wt, void* dx, const void* residual,
                           hipStream_t st) {
    const int M = g.N * (g.H / 2) * (g.W / 2);
    for (int pa = 0; pa < 2; ++pa)
        for (int pb = 0; pb < 2; ++pb) {
            TapList taps; taps.n = 0;
            for (int t = 0; t < 4; ++t) taps.kh[t] = taps.kw[t] = 0;
            for (int kh = (pa + g.pad) & 1; kh < g.R; kh += 2)
                for (int kw = (pb + g.pad) & 1; kw < g.S; kw += 2)
                    if (taps.n < 4) { taps.kh[taps.n] = kh; taps.kw[taps.n] = kw; ++taps.n; }
            const int Kd = taps.n * g.KO;
            EpiStore<T> ep{(T*)dx, g.C, nullptr, (const T*)residual, g.C, nullptr, ACT_NONE, 1.f, make_dropout(0.f, 0), M, g.C};
            ep.map_on = 1; ep.map_H = g.H; ep.map_W = g.W; ep.map_pa = pa; ep.map_pb = pb;
            launch_auto<T, ConvDgradS2A, TapKC>(
                [&](auto& a) { a.dy = (const T*)dy; a.g = g; a.rows = M; a.K = Kd; a.pa = pa; a.pb = pb; a.taps = taps; },
                [&](auto& b) { b.p = (const T*)wt; b.ld = (long)g.R * g.S * g.KO; b.rows = g.C; b.K = Kd; b.logKO = g.logKO;
                               b.S = g.S; b.taps = taps; },
                ep, M, g.C, Kd, 1, st);
        }
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_conv2d_dgrad(int dtype, int N, int H, int W, int C, int KO, int R, int S, int stride,
                                int pad, const void* dy, const void* wt, void* dx, const void* residual, void* stream) {
    VTX_CHECK(dy && wt && dx, VTX_ERR_ARG, "conv2d_dgrad: null pointer");
    ConvGeo g;
    int rc = make_geo("conv2d_dgrad", dtype, N, H, W, C, KO, R, S, stride, pad, &g);
    if (rc) return rc;
    if (stride == 2 && H % 2 == 0 && W % 2 == 0 && R <= 4 && S <= 4) {
        if (dtype == VTX_BF16) return conv_dgrad_s2_t<bf16_t>(g, dy, wt, dx, residual, (hipStream_t)stream);
        return conv_dgrad_s2_t<float>(g, dy, wt, dx, residual, (hipStream_t)stream);
    }
    if (dtype == VTX_BF16) return conv_dgrad_t<bf16_t>(g, dy, wt, dx, residual, (hipStream_t)stream);
    return conv_dgrad_t<float>(g, dy, wt, dx, residual, (hipStream_t)stream);
}

// ---- the same two schedules with the BatchNorm-backward fusion in the epilogue (bf16, generation-2 kernel)
static int conv_dgrad_bn(const ConvGeo& g, const void* dy, const void* wt, void* dx, const void* residual,
                         VtxBnBwdFusion* f, hipStream_t st) {
    typedef bf16_t T;
    const int M = g.N * g.H * g.W, Kd = g.R * g.S * g.KO;
    EpiStore<T, STATS_BWD> ep{(T*)dx, g.C, nullptr, (const T*)residual, g.C, nullptr, ACT_NONE, 1.f, make_dropout(0.f, 0), M, g.C};
    vtx_fill_bn_bwd(ep, f, g.C, f->parts);
    auto mk_b = [&](auto& b) { b.p = (const T*)wt; b.ld = Kd; b.rows = g.C; b.K = Kd; };
    int strips = 0;
    if (g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g_vtx_tile_override < 0)
        strips = conv3x3_shared_try(dy, g.N, g.H, g.W, g.KO, g.C, mk_b, ep, 1, st);
    if (strips == 0)
        strips = launch_auto<T, ConvDgradA, PlainKC>([&](auto& a) { a.dy = (const T*)dy; a.g = g; a.rows = M; a.K = Kd; }, mk_b, ep, M, g.C, Kd, 1, st);
    f->strips = strips;
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

static int conv_dgrad_s2_bn(const ConvGeo& g, const void* dy, const void* wt, void* dx, const void* residual,
                            VtxBnBwdFusion* f, hipStream_t st) {
    typedef bf16_t T;
    const int M = g.N * (g.H / 2) * (g.W / 2);
    int strips = 0;
    for (int pa = 0; pa < 2; ++pa)
        for (int pb = 0; pb < 2; ++pb) {
            TapList taps; taps.n = 0;
            for (int t = 0; t < 4; ++t) taps.kh[t] = taps.kw[t] = 0;
            for (int kh = (pa + g.pad) & 1; kh < g.R; kh += 2)
                for (int kw = (pb + g.pad) & 1; kw < g.S; kw += 2)
                    if (taps.n < 4) { taps.kh[taps.n] = kh; taps.kw[taps.n] = kw; ++taps.n; }
            const int Kd = taps.n * g.KO;
            EpiStore<T, STATS_BWD> ep{(T*)dx, g.C, nullptr, (const T*)residual, g.C, nullptr, ACT_NONE, 1.f, make_dropout(0.f, 0), M, g.C};
            ep.map_on = 1; ep.map_H = g.H; ep.map_W = g.W; ep.map_pa = pa; ep.map_pb = pb;
            // every parity class writes its own strips, one after the other (the rows of the classes are disjoint)
            vtx_fill_bn_bwd(ep, f, g.C, f->parts + (size_t)strips * 2 * g.C);
            strips += launch_auto<T, ConvDgradS2A, TapKC>(
                [&](auto& a) { a.dy = (const T*)dy; a.g = g; a.rows = M; a.K = Kd; a.pa = pa; a.pb = pb; a.taps = taps; },
                [&](auto& b) { b.p = (const T*)wt; b.ld = (long)g.R * g.S * g.KO; b.rows = g.C; b.K = Kd; b.logKO = g.logKO;
                               b.S = g.S; b.taps = taps; },
                ep, M, g.C, Kd, 1, st);
        }
    f->strips = strips;
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_conv2d_dgrad_bnbwd(int dtype, int N, int H, int W, int C, int KO, int R, int S, int stride, int pad,
                                      const void* dy, const void* wt, void* dx, const void* residual, VtxBnBwdFusion* f,
                                      void* stream) {
    VTX_CHECK(f, VTX_ERR_ARG, "conv2d_dgrad_bnbwd: null fusion descriptor");
    f->strips = 0;
    if (dtype != VTX_BF16 || g_vtx_contraction_generation < 2)     // not fused: plain gradient (see virtex_amd.h)
        return vtx_conv2d_dgrad(dtype, N, H, W, C, KO, R, S, stride, pad, dy, wt, dx, residual, stream);
    VTX_CHECK(dy && wt && dx, VTX_ERR_ARG, "conv2d_dgrad_bnbwd: null pointer");
    ConvGeo g;
    int rc = make_geo("conv2d_dgrad_bnbwd", dtype, N, H, W, C, KO, R, S, stride, pad, &g);
    if (rc) return rc;
    rc = vtx_check_bn_bwd("conv2d_dgrad_bnbwd", f, N * H * W, C);
    if (rc) return rc;
    if (stride == 2 && H % 2 == 0 && W % 2 == 0 && R <= 4 && S <= 4) return conv_dgrad_s2_bn(g, dy, wt, dx, residual, f, (hipStream_t)stream);
    return conv_dgrad_bn(g, dy, wt, dx, residual, f, (hipStream_t)stream);
}
