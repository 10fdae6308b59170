// BatchNorm backward of a 1x1 "expand" convolution's output FOLDED INTO THE CONVOLUTION'S WEIGHTS: no pass over the tensor.
//
// Setting (torchvision Bottleneck, reached from /root/reference/virtex/modules/visual_backbones.py:68-74; aten::
// native_batch_norm_backward of bn3 + aten::convolution_backward of conv3):
//      x3 = a3 . W3^T           conv3, 1x1: a3 [P][N] = relu(bn2(x2)), W3 [K][N], x3 [P][K]      (N = planes, K = 4 planes)
//      y  = relu(bn3(x3) + skip)
// and in backward, with dz [P][K] the (already masked) gradient wrt bn3's output and its sums s1 = sum dz, s2 = sum dz xhat:
//      dx3 = k0 (dz - s1/P - xhat s2/P),   xhat = (x3 - mu) rstd,   k0 = gamma rstd                 (BatchNorm backward)
//          = a0 dz + b1 x3 + c            a0 = k0,  b1 = -k0 (s2/P) rstd,  c = k0 ((s2/P) rstd mu - s1/P)      per channel k
// The BatchNorm backward is AFFINE per channel, and x3 is itself a linear image of a3.  So both consumers of dx3 can be written
// without dx3:
//   input gradient   dy2 = dx3 . W3           = dz . (a0 o W3)  +  a3 . H  +  bias,       H = W3^T diag(b1) W3  [N][N],  bias = W3^T c
//   weight gradient  dW3 = dx3^T . a3         = diag(a0) (dz^T a3) + diag(b1) W3 (a3^T a3) + c (colsum a3)^T
// i.e. the 3-tensor pass "read x3, read dz, write dx3" (0.6 GB per stage-2 block at bs 256) is replaced by a [P][N] x [N][N]
// product on the SMALL side of the Bottleneck and a Gram matrix of a3; x3 is not read in backward at all.  The identities are
// exact in real arithmetic; in bf16 the folded form rounds (a0 o W3), H and the [P][N] partial product where the pass form
// rounded dx3 -- tests/test_kernels.py::test_bn_backward_folded_into_conv3 holds both against the fp32 formulas.
//
// This file: the two small kernels the scheme adds (everything else is existing contractions, orchestrated by
// virtex_amd/modules/visual_backbones.py::_backward_blocks):
//   vtx_bn_bwd_fold          finalize of the sums (as vtx_bn_bwd_fused) + Wa = a0 o wt, Wb = b1 o wt (bf16), bias = wt . c, coefs
//   vtx_wgrad_fold_combine   dW[k][n] += a0[k] T[k][n] + b1[k] WG[k][n] + c[k] s[n]
#include "vtx_common.h"

int vtx_bn_bwd_finalize_only(const float* gamma, const float* save_rstd, const float* pre_partials, int pre_nparts, float* dgamma,
                             float* dbeta, float* workspace, int P, int C, hipStream_t st, const float** coef_out);

namespace {

// one workgroup per row n of wt [N][K]: a thread owns 8 consecutive k per trip
__global__ __launch_bounds__(256) void bn_fold_kernel(const bf16_t* __restrict__ wt, long ldw, const float* __restrict__ coef /* [3][K]: k0, s1/P, s2/P */,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      bf16_t* __restrict__ wa, bf16_t* __restrict__ wb, float* __restrict__ bias,
                                                      float* __restrict__ abc /* [3][K]: a0, b1, c */, int K) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int k0 = threadIdx.x * 8; k0 < K; k0 += 256 * 8) {
        Vec16<bf16_t> w; w.load(wt + (long)n * ldw + k0);
        Vec16<bf16_t> oa, ob;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            const float g = coef[k], m1 = coef[K + k], m2 = coef[2 * K + k], rs = rstd[k];
            const float a0 = g, b1 = -g * m2 * rs, c = g * (m2 * rs * mean[k] - m1);
            oa.v[j] = a0 * w.v[j];
            ob.v[j] = b1 * w.v[j];
            acc += c * w.v[j];
            if (n == 0) { abc[k] = a0; abc[K + k] = b1; abc[2 * K + k] = c; }
        }
        oa.store(wa + (long)n * K + k0);
        ob.store(wb + (long)n * K + k0);
    }
    acc = block_sum<4>(acc, red);
    if (threadIdx.x == 0) bias[n] = acc;
}

__global__ __launch_bounds__(256) void wgrad_fold_combine_kernel(float* __restrict__ dw, long ldd, const float* __restrict__ T,
                                                                 const float* __restrict__ WG, const float* __restrict__ s,
                                                                 const float* __restrict__ abc, int K, int N) {
    const long nv = (long)K * N / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const long e = i * 4;
        const int k = (int)(e / N), n = (int)(e - (long)k * N);          // N % 4 == 0: a float4 never straddles rows
        const float a0 = abc[k], b1 = abc[K + k], c = abc[2 * K + k];
        const float4 t = reinterpret_cast<const float4*>(T)[i], g = reinterpret_cast<const float4*>(WG)[i];
        const float4 sv = *reinterpret_cast<const float4*>(s + n);
        float4* dst = reinterpret_cast<float4*>(dw + (long)k * ldd + n);
        float4 d = *dst;
        d.x += a0 * t.x + b1 * g.x + c * sv.x;
        d.y += a0 * t.y + b1 * g.y + c * sv.y;
        d.z += a0 * t.z + b1 * g.z + c * sv.z;
        d.w += a0 * t.w + b1 * g.w + c * sv.w;
        *dst = d;
    }
}

}  // namespace

extern "C" int vtx_bn_bwd_fold(const float* gamma, const float* save_mean, const float* save_rstd, const float* pre_partials,
                               int pre_nparts, float* dgamma, float* dbeta, float* bn_workspace, int P, int K, const void* wt,
                               long ldw, int N, void* wa, void* wb, float* bias, float* abc, void* stream) {
    VTX_CHECK(gamma && save_mean && save_rstd && pre_partials && dgamma && dbeta && bn_workspace && wt && wa && wb && bias && abc,
              VTX_ERR_ARG, "bn_bwd_fold: null pointer");
    VTX_CHECK(P > 0 && N > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldw >= K, VTX_ERR_SHAPE,
              "bn_bwd_fold: K and the row stride of wt must be multiples of 8 (K=%d ldw=%ld)", K, ldw);
    VTX_CHECK((((uintptr_t)wt | (uintptr_t)wa | (uintptr_t)wb) & 15) == 0, VTX_ERR_SHAPE, "bn_bwd_fold: weights must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const float* coef = nullptr;
    int rc = vtx_bn_bwd_finalize_only(gamma, save_rstd, pre_partials, pre_nparts, dgamma, dbeta, bn_workspace, P, K, st, &coef);
    if (rc) return rc;
    VTX_KLAUNCH("bn_bwd_fold", 0, 2.0 * 3 * N * K + 4.0 * 8 * K, bn_fold_kernel, dim3(N), dim3(256), 0, st, (const bf16_t*)wt, ldw, coef,
                save_mean, save_rstd, (bf16_t*)wa, (bf16_t*)wb, bias, abc, K);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_wgrad_fold_combine(float* dw, long ldd, const float* T, const float* WG, const float* s, const float* abc,
                                      int K, int N, void* stream) {
    VTX_CHECK(dw && T && WG && s && abc, VTX_ERR_ARG, "wgrad_fold_combine: null pointer");
    VTX_CHECK(K > 0 && N > 0 && N % 4 == 0 && ldd % 4 == 0 && ldd >= N, VTX_ERR_SHAPE, "wgrad_fold_combine: N and ldd must be multiples of 4");
    VTX_CHECK((((uintptr_t)dw | (uintptr_t)T | (uintptr_t)WG | (uintptr_t)s) & 15) == 0, VTX_ERR_SHAPE, "wgrad_fold_combine: operands must be 16-byte aligned");
    const long nv = (long)K * N / 4;
    long g = (nv + 255) / 256;
    if (g > 2048) g = 2048;
    VTX_KLAUNCH("wgrad_fold_combine", 0, 4.0 * 4 * K * N, wgrad_fold_combine_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, dw, ldd, T, WG, s,
                abc, K, N);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
