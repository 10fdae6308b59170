// Host-side helpers shared by the NHWC implicit-GEMM convolution entry points.
#pragma once
#include "gemm_kernel.h"

namespace vtxg {

static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Validates the problem and fills the device geometry.  Returns 0 or a negative status.
static inline int make_geo(const char* who, int dtype, int N, int H, int W, int C, int KO, int R, int S,
                           int stride, int pad, ConvGeo* g) {
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "%s: bad dtype %d", who, dtype);
    VTX_CHECK(N > 0 && H > 0 && W > 0 && C > 0 && KO > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0,
              VTX_ERR_ARG, "%s: bad geometry", who);
    // C = vec/2 (bf16, 4 channels): a 16-byte chunk spans the taps (kw, kw+1) of two adjacent pixels.  Valid when the
    // first tap of every chunk has an even kw (S even), lands on an even column (pad 0, stride 2) and the row has an
    // even length, so that the second pixel is in bounds whenever the first is: the haloed 7x8 stem.
    const bool pixel_pairs = dtype == VTX_BF16 && C == 4 && pad == 0 && stride == 2 && S % 2 == 0 && W % 2 == 0;
    VTX_CHECK(is_pow2(C) && (C >= vec || pixel_pairs), VTX_ERR_SHAPE, "%s: C=%d must be a power of two >= %d (pad the "
              "stem's input channels)", who, C, vec);
    VTX_CHECK(is_pow2(KO) && KO >= vec, VTX_ERR_SHAPE, "%s: K=%d must be a power of two >= %d", who, KO, vec);
    VTX_CHECK(is_pow2(stride) && S <= 8 && R <= 8, VTX_ERR_SHAPE, "%s: stride must be a power of two, filter <= 8", who);
    g->N = N; g->H = H; g->W = W; g->C = C; g->logC = vtx_ilog2(C);
    g->KO = KO; g->logKO = vtx_ilog2(KO);
    g->R = R; g->S = S; g->rcpS = (65536 + S - 1) / S;
    g->stride = stride; g->logStride = vtx_ilog2(stride); g->pad = pad;
    g->OH = (H + 2 * pad - R) / stride + 1;
    g->OW = (W + 2 * pad - S) / stride + 1;
    VTX_CHECK(g->OH > 0 && g->OW > 0, VTX_ERR_SHAPE, "%s: empty output", who);
    g->inv_ow = 1.0f / (float)g->OW;
    g->inv_ohow = 1.0f / (float)(g->OH * g->OW);
    VTX_CHECK((long)N * H * W < VTX_PIXEL_LIMIT && (long)N * g->OH * g->OW < VTX_PIXEL_LIMIT, VTX_ERR_SHAPE,
              "%s: more than 2^30 pixels per tensor is not supported", who);
    return VTX_OK;
}

}  // namespace vtxg

int vtx_pick_split_k(int M, int N, int K, int bk, long ws_floats, int gather = 0);
vtxg::EpiStore<float> vtx_splitk_epilogue(float* C, long ldc, float alpha, int M, int N, int split_k, float* ws);
void vtx_splitk_reduce(const float* ws, int S, int M, int N, float* C, long ldc, hipStream_t st);
float* vtx_splitk_region(float* ws, long ws_floats, long* cap, hipStream_t st);      // gemm.hip: the part of the workspace a contraction may use now
