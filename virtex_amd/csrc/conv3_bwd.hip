// The backward pass of a stage-1 Bottleneck's conv3 (1x1, 64 -> 256 channels at 56x56) as ONE streaming kernel:
//
//      dz  (gradient wrt bn3's output, masked by the producing join kernel)      [M][256] bf16  -- read once
//      x3  (bn3's input = conv3's output)                                         [M][256] bf16  -- read once
//   -> dx3 = BatchNorm3-backward apply: a0*dz - a1*x3 - (a2 - mu*a1)                (registers / LDS only: never in HBM)
//   -> dy2 = dx3 . W3          (conv3's input gradient, MFMA)  + bn2's ReLU mask + bn2's backward sums   [M][64] -- written once
//   -> dW3 += dx3^T . a3       (conv3's weight gradient, MFMA), a3 = relu(bn2(x2)) recomputed from x2   [M][64] -- read once
//
// Replaces, for torchvision's Bottleneck (/root/reference/virtex/modules/visual_backbones.py:68-74; autograd's
// native_batch_norm_backward + convolution_backward of bn3 / conv3, scripts/pretrain_virtex.py:154), three launches of
// rounds 1-4: bn_bwd_apply_fused (reads dz, x3, WRITES dx3: 3 x 411 MB at bs 256), the input-gradient GEMM with the fused
// bn2 epilogue (READS dx3 again, x2; writes dy2) and -- on the weight-gradient stream -- the k-major GEMM that READS dx3 a
// third time plus a3.  dx3 is the largest gradient tensor of the block and existed only to be re-read twice.
// Bytes per launch at bs 256 (M = 802 816): 411 + 411 + 103 + 103 = 1 028 MB against 1 233 + 617 + 514 = 2 364 MB.
//
// Shape of the kernel (HBM-bound: 12 % of the MFMA rate would do, LDS at ~30 %):
//   * one workgroup of 8 waves per CU walks 128-row blocks (block-cyclic over the grid: the chip streams one contiguous
//     16-MB window of dz / x3 at a time); conv3's transposed weights [64][256] live in LDS for the life of the workgroup;
//   * LOAD by channel slice: wave w fetches channels [32w, 32w+32) of all 128 rows of dz and x3 straight into registers
//     (16 rows x 64 B per instruction), ONE ROUND AHEAD -- 144 KB per CU in flight under the current round's arithmetic.
//     A lane then owns 8 channels for its whole life: the BatchNorm coefficients of the transform are 32 values per lane;
//   * the transformed slice goes to a row-major LDS image DX[128][256] (XOR-swizzled like the k-major images of
//     gemm_kernel.h).  That ONE image feeds both contractions: the input gradient reads it along its rows (ds_read_b128
//     fragments: contraction over channels), the weight gradient reads it transposed (ds_read_b64_tr_b16: contraction over
//     the 128 rows) -- and wave w's weight-gradient columns are exactly the slice it wrote, so the image needs no barrier
//     between a round's weight gradient and the next round's transform;
//   * input gradient by row strip (wave w: rows 16w..16w+15, all 256 channels, 32 MFMAs), epilogue through a wave-private
//     strip exactly as EpiStore<bf16, STATS_BWD> does it (same roundings: mask recomputed from x2, sums of the ROUNDED
//     gradient), plus a3 = relu((x2 - mean) * (gamma * rstd) + beta) -- bn_apply_kernel's formula -- into a second image;
//   * weight gradient by channel slice (wave w: dW3 rows 32w..32w+31 x 64 columns = 32 accumulator VGPRs per lane for the
//     whole launch), one fp32 partial [256][64] per workgroup, folded by splitk_reduce; one statistics partial per workgroup;
//   * two workgroup barriers per round of 164 KB.
// Entry: vtx_conv3_bwd_fused (below); shapes it does not take return VTX_ERR_SHAPE -- ask vtx_conv3_bwd_fused_supported first.
#include <stdlib.h>

#include "vtx_common.h"

extern int g_vtx_sw_conv3_bwd;
// bn.hip: [compaction +] bn_bwd_finalize of `parts` -> coef[3][C] inside `workspace`, dgamma / dbeta accumulated
int vtx_bn_bwd_finalize_only(const float* gamma, const float* save_rstd, const float* pre_partials, int pre_nparts, float* dgamma,
                             float* dbeta, float* workspace, int P, int C, hipStream_t st, const float** coef_out);
void vtx_splitk_reduce_now(const float* ws, int S, int M, int N, float* C, long ldc, hipStream_t st);

#ifndef VTX_CB_ABL
#define VTX_CB_ABL 0       // measurement builds only (tools/r05_s5.sh): 1 no weight gradient, 2 no input-gradient MFMAs, 4 no epilogue,
#endif                     // 8 transform = copy, 16 no dz / x3 loads -- results are wrong with any of them set

namespace {

constexpr int CB_K = 256, CB_N = 64, CB_RB = 128, CB_NW = 8, CB_T = 64 * CB_NW;
constexpr int CB_SROWB = CB_N * 2 + 16;                             // epilogue strip row: 128 B + pad
constexpr int OFF_W = 0;                                            // [2][64][64] bf16, swizzled            32 KiB
constexpr int OFF_DX = OFF_W + CB_N * CB_K * 2;                     // [128][256] bf16, swizzled             64 KiB
constexpr int OFF_A3 = OFF_DX + CB_RB * CB_K * 2;                   // [128][64] bf16, swizzled              16 KiB
constexpr int OFF_STRIP = OFF_A3 + CB_RB * CB_N * 2;                // [8 waves][16][144 B]                  18 KiB
constexpr int OFF_TC = OFF_STRIP + CB_NW * 16 * CB_SROWB;           // [4][256] f32: mu3, a0, a1, a2          4 KiB
constexpr int OFF_PAR = OFF_TC + 4 * CB_K * 4;                      // [6][64] f32: rstd2, -mean2*rstd2, gamma2, beta2, mean2, gamma2*rstd2
constexpr int CB_LDS = OFF_PAR + 6 * CB_N * 4;                      // 138 752 B

// weight image: rows of 64 k (128 B), the eight 16-byte slots of a row XOR-permuted by (n >> 1) & 7 (expand1x1.hip)
__device__ __forceinline__ int wslot(int n, int s) { return n * 8 + (s ^ ((n >> 1) & 7)); }
// 16-byte chunk swizzles of the row-major images that are also read transposed: gemm_kernel.h's swz_mc<ROWS> (0 bank
// conflicts by the PMC counters for both read shapes)
__device__ __forceinline__ int swz_dx(int chunk, int m) { return chunk ^ (((m & 3) << 1) | (((m >> 3) & 1) << 3)); }
__device__ __forceinline__ int swz_a3(int chunk, int m) { return chunk ^ ((((m >> 1) & 1) << 1) | (((m >> 3) & 1) << 2)); }

__device__ __forceinline__ void ld8(const float* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void unpack8(uint4 w, float* f) {
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(u[i] << 16); f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
}

struct Conv3BwdArgs {
    const bf16_t* dz; const bf16_t* x3; const float* coef3; const float* mean3; const float* rstd3;
    const bf16_t* wt; long ldw;
    const bf16_t* x2; const float* mean2; const float* rstd2; const float* gamma2; const float* beta2;
    bf16_t* dy2; float* parts2; float* dwp; int nrb;
};

__global__ __launch_bounds__(CB_T, 2) void conv3_bwd_fused_kernel(const Conv3BwdArgs a) {
    HIP_DYNAMIC_SHARED(char, smem)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, slot = lane >> 4;
    bf16_t* wimg = reinterpret_cast<bf16_t*>(smem + OFF_W);
    float* tc = reinterpret_cast<float*>(smem + OFF_TC);
    float* par = reinterpret_cast<float*>(smem + OFF_PAR);

    // ---- the round-ahead loads: this wave's channel slice of dz / x3 (8 x 16 rows x 64 B each) and the x2 chunks of the
    //      16 rows it finishes in the epilogue
    uint4 rdz[8], rx3[8], rx2[2];
    auto issue = [&](int rb) {
        const long m0 = (long)rb * CB_RB;
        const long o = (m0 + l15) * CB_K + 32 * wave + 8 * slot;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            rdz[j] = *reinterpret_cast<const uint4*>(a.dz + o + (long)j * 16 * CB_K);
            rx3[j] = *reinterpret_cast<const uint4*>(a.x3 + o + (long)j * 16 * CB_K);
        }
        const long o2 = (m0 + 16 * wave + (lane >> 3)) * CB_N + 8 * (lane & 7);
#pragma unroll
        for (int q = 0; q < 2; ++q) rx2[q] = *reinterpret_cast<const uint4*>(a.x2 + o2 + (long)q * 8 * CB_N);
    };
    int rb = blockIdx.x;
    if (rb < a.nrb) issue(rb);

    // ---- once per workgroup: weights, BatchNorm tables
    for (int c = tid; c < CB_N * (CB_K / 8); c += CB_T) {
        const int n = c / (CB_K / 8), s = c % (CB_K / 8);
        const uint4 v = *reinterpret_cast<const uint4*>(a.wt + (long)n * a.ldw + s * 8);
        *reinterpret_cast<uint4*>(wimg + (size_t)(s >> 3) * CB_N * 64 + wslot(n, s & 7) * 8) = v;
    }
    for (int k = tid; k < CB_K; k += CB_T) {
        // dx = k0*(dz - k1 - xhat*k2), xhat = (x - mu)*rs  ==  k0*dz - (x - mu)*(k0*k2*rs) - k0*k1   (bn_bwd_apply_fused_kernel)
        const float k0 = a.coef3[k], k1 = a.coef3[CB_K + k], k2 = a.coef3[2 * CB_K + k];
        const float a1 = k0 * k2 * a.rstd3[k];
        tc[k] = a.mean3[k]; tc[CB_K + k] = k0; tc[2 * CB_K + k] = a1; tc[3 * CB_K + k] = k0 * k1 - a.mean3[k] * a1;
    }
    for (int n = tid; n < CB_N; n += CB_T) {
        const float rs = a.rstd2[n], mu = a.mean2[n], ga = a.gamma2[n];
        par[n] = rs; par[CB_N + n] = -mu * rs; par[2 * CB_N + n] = ga; par[3 * CB_N + n] = a.beta2[n];
        par[4 * CB_N + n] = mu; par[5 * CB_N + n] = ga * rs;            // bn_fwd_finalize_kernel: scale = gamma * rstd
    }
    __syncthreads();

    f32x4_t accw[2][4];                                                // dW3 rows 32w + 16i + (lane & 15), columns 16j + 4*(lane>>4) ..
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) accw[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float s1[8], s2[8];                                                // bn2 backward sums of this lane's column chunk
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    char* strip = smem + OFF_STRIP + wave * 16 * CB_SROWB;
    const int ech = lane & 7, er0 = lane >> 3;                         // epilogue: column chunk / first row of this lane
    const int kc = 32 * wave + 8 * slot;                               // first of this lane's eight dz / x3 channels

    for (; rb < a.nrb; rb += gridDim.x) {
        const long m0 = (long)rb * CB_RB;
        // ---- (1) BatchNorm3 backward on the slice, into the DX image.  The registers of a transformed chunk are refilled
        //      with the NEXT round's chunk at once: the memory pipe is fed all through the transform (issuing the whole next
        //      round behind it left every CU with nothing in flight for the ~3 000 cycles of the transform: 258 us, 3.98 TB/s).
        const bool more = rb + (int)gridDim.x < a.nrb;                 // wave-uniform
        const uint4 x2c0 = rx2[0], x2c1 = rx2[1];
        {
            float c0[8], c1[8], c2[8];
            ld8(tc + CB_K + kc, c0); ld8(tc + 2 * CB_K + kc, c1); ld8(tc + 3 * CB_K + kc, c2);
            auto chunk = [&](int j) {
                float g[8], x[8];
                unpack8(rdz[j], g); unpack8(rx3[j], x);
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // a0*dz - a1*x - (a2 - mu*a1): two FMAs per element (bn_bwd_apply_fused_kernel centres x first: three
                    // operations; the difference is rounding of the small x-term, 2.7e-5 relative on dy2)
                    const float v0 = c0[2 * e] * g[2 * e] - (x[2 * e] * c1[2 * e] + c2[2 * e]);
                    const float v1 = c0[2 * e + 1] * g[2 * e + 1] - (x[2 * e + 1] * c1[2 * e + 1] + c2[2 * e + 1]);
                    o[e] = f2bf2(v0, v1);
                }
                const int m = 16 * j + l15;
                if (VTX_CB_ABL & 8) { o[0] = rdz[j].x ^ rx3[j].x; o[1] = rdz[j].y; o[2] = rdz[j].z; o[3] = rdz[j].w; }
                *reinterpret_cast<uint4*>(smem + OFF_DX + m * (CB_K * 2) + swz_dx(4 * wave + slot, m) * 16) = make_uint4(o[0], o[1], o[2], o[3]);
            };
            {
                // branch-free (with the refill inside `if (more)` hipcc kept both register sets alive: 134 spilled VGPRs): after
                // the last round every lane re-reads the first 16 bytes of the tensors -- one cache line, no traffic
                const long o = more ? ((long)(rb + gridDim.x) * CB_RB + l15) * CB_K + 32 * wave + 8 * slot : 0;
                const long js = more ? 16 * CB_K : 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    chunk(j);
                    vtx_loads_issued();                                // (scheduling barriers: the refill stays behind the chunk's last use)
                    if (!(VTX_CB_ABL & 16)) {
                        rdz[j] = *reinterpret_cast<const uint4*>(a.dz + o + j * js);
                        rx3[j] = *reinterpret_cast<const uint4*>(a.x3 + o + j * js);
                    }
                    vtx_loads_issued();
                }
                const long o2 = more ? ((long)(rb + gridDim.x) * CB_RB + 16 * wave + (lane >> 3)) * CB_N + 8 * (lane & 7) : 0;
                const long qs = more ? 8 * CB_N : 0;
#pragma unroll
                for (int q = 0; q < 2; ++q) rx2[q] = *reinterpret_cast<const uint4*>(a.x2 + o2 + q * qs);
            }
        }
        __syncthreads();                                               // B1: DX complete

        // ---- (2) input gradient of rows 16w .. 16w+15: dy2[m][n] = sum_k dx3[m][k] * wt[n][k]
        f32x4_t acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (!(VTX_CB_ABL & 2)) {
            const int mr = 16 * wave + l15;
            const char* arow = smem + OFF_DX + mr * (CB_K * 2);
#pragma unroll
            for (int ks = 0; ks < CB_K / 32; ++ks) {
                const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(arow + swz_dx(4 * ks + slot, mr) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(
                        wimg + (size_t)(ks >> 1) * CB_N * 64 + wslot(16 * j + l15, (ks & 1) * 4 + slot) * 8);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, acc[j], 0, 0, 0);
                }
            }
        }
        // ---- (3) epilogue: bn2's ReLU mask + backward sums (EpiStore<bf16, STATS_BWD>::stats_math, mask recomputed), a3
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<uint2*>(strip + l15 * CB_SROWB + (16 * j + 4 * slot) * 2) =
                make_uint2(f2bf2(acc[j][0], acc[j][1]), f2bf2(acc[j][2], acc[j][3]));
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < ((VTX_CB_ABL & 4) ? 0 : 2); ++q) {
            const int r = er0 + 8 * q;
            const uint4 w = *reinterpret_cast<const uint4*>(strip + r * CB_SROWB + ech * 16);
            float g[8], x[8], rs[8], sh[8], be[8], xh[8];
            unpack8(w, g); unpack8(q == 0 ? x2c0 : x2c1, x);
            const float* pl = par + 8 * ech;
            ld8(pl, rs); ld8(pl + CB_N, sh); ld8(pl + 3 * CB_N, be);
            uint32_t wo[4];
            // the mask as the forward pass took it: y2 = (x2 - mean) * scale + beta > 0 -- bn_apply_kernel's expression, which
            // the recomputed conv3 input below needs anyway (EpiStore<.., STATS_BWD> tests xhat * gamma + beta > 0: the same
            // predicate up to the rounding of a value at the threshold)
            float mu2l[8], sc2l[8], yl[8];
            ld8(pl + 4 * CB_N, mu2l); ld8(pl + 5 * CB_N, sc2l);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[e] = x[e] * rs[e] + sh[e];
                yl[e] = (x[e] - mu2l[e]) * sc2l[e] + be[e];
                g[e] = yl[e] > 0.f ? g[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {                                  // sums of what is stored (the ROUNDED gradient)
                const uint32_t u = f2bf2(g[2 * e], g[2 * e + 1]);
                wo[e] = u;
                g[2 * e] = __uint_as_float(u << 16); g[2 * e + 1] = __uint_as_float(u & 0xffff0000u);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += g[e]; s2[e] += g[e] * xh[e]; }
            *reinterpret_cast<uint4*>(a.dy2 + (m0 + 16 * wave + r) * CB_N + 8 * ech) = make_uint4(wo[0], wo[1], wo[2], wo[3]);
            // conv3's input, recomputed: y2 = relu((x2 - mean) * scale + beta) as bn_apply_kernel stored it
            uint32_t yo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) yo[e] = f2bf2(fmaxf(yl[2 * e], 0.f), fmaxf(yl[2 * e + 1], 0.f));
            const int m = 16 * wave + r;
            *reinterpret_cast<uint4*>(smem + OFF_A3 + m * (CB_N * 2) + swz_a3(ech, m) * 16) = make_uint4(yo[0], yo[1], yo[2], yo[3]);
        }
        __syncthreads();                                               // B2: A3 complete (and every wave is past its DX reads)

        // ---- (4) weight gradient of channels 32w .. 32w+31: dW3[ko][n] += sum_m dx3[m][ko] * a3[m][n]
#pragma unroll 1            // (unrolled: 256 VGPRs + scratch; rolled: 237, and the addresses differ by immediates only)
        for (int t = 0; t < ((VTX_CB_ABL & 1) ? 0 : CB_RB / 32); ++t) {
            const int ka = 32 * t + 8 * slot + (l15 >> 2), c4 = 4 * (l15 & 3);
            vtx_v4s_t ry[2][2], rx[4][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = ka + 4 * h;
                const char* drow = smem + OFF_DX + m * (CB_K * 2);
                const char* arow = smem + OFF_A3 + m * (CB_N * 2);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int rr = 32 * wave + 16 * i + c4;
                    ry[i][h] = vtx_ds_read_tr16(drow + swz_dx(rr >> 3, m) * 16 + (rr & 7) * 2);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rr = 16 * j + c4;
                    rx[j][h] = vtx_ds_read_tr16(arow + swz_a3(rr >> 3, m) * 16 + (rr & 7) * 2);
                }
            }
            vtx_ds_tr_wait();
            bf16x8_t fy[2], fx[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) fy[i] = __builtin_shufflevector(ry[i][0], ry[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int j = 0; j < 4; ++j) fx[j] = __builtin_shufflevector(rx[j][0], rx[j][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) accw[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[j], fy[i], accw[i][j], 0, 0, 0);
        }
        // no barrier here: the next round's transform overwrites only THIS wave's DX columns (which only this wave reads
        // above), and A3 / the other waves' columns are not written before the next B1
    }

    // ---- one weight-gradient partial and one statistics partial per workgroup
    {
        float* dst = a.dwp + (size_t)blockIdx.x * CB_K * CB_N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(dst + (size_t)(32 * wave + 16 * i + l15) * CB_N + 16 * j + 4 * slot) =
                    make_float4(accw[i][j][0], accw[i][j][1], accw[i][j][2], accw[i][j][3]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int msk = 8; msk < 64; msk <<= 1) { s1[e] += __shfl_xor(s1[e], msk, 64); s2[e] += __shfl_xor(s2[e], msk, 64); }
    }
    float* red = tc;                                                   // [8 waves][2][64]: the transform tables are done with
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(wave * 2 + 0) * CB_N + lane * 8 + e] = s1[e];
            red[(wave * 2 + 1) * CB_N + lane * 8 + e] = s2[e];
        }
    }
    __syncthreads();
    if (tid < 2 * CB_N) {
        const int which = tid / CB_N, c = tid % CB_N;
        float v = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < CB_NW; ++w2) v += red[(w2 * 2 + which) * CB_N + c];
        a.parts2[(size_t)blockIdx.x * 2 * CB_N + which * CB_N + c] = v;
    }
}

int g_cb_ncu = 0;

}  // namespace

extern "C" int vtx_conv3_bwd_fused_supported(int dtype, int M, int K, int N) {
    return g_vtx_sw_conv3_bwd && dtype == VTX_BF16 && K == CB_K && N == CB_N && M % CB_RB == 0 && M >= 4 * CB_RB;
}

// Workgroups (= weight-gradient / statistics partials) a launch over M rows uses: the caller sizes dw_parts with it.
extern "C" int vtx_conv3_bwd_fused_parts(int M) {
    if (g_cb_ncu == 0) {
        int n = 256;                                        // MI355X; one persistent workgroup per CU
#ifndef HIPEMU
        int dev = 0, q = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&q, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && q > 0) n = q;
        (void)hipGetLastError();
#else
        n = 5;                                              // the emulator's grid: workgroups with different round counts
#endif
        g_cb_ncu = n;
    }
    const int nrb = M / CB_RB;
    return nrb < g_cb_ncu ? nrb : g_cb_ncu;
}

extern "C" int vtx_conv3_bwd_fused(int dtype, int M, int K, int N, const void* dz, const void* x3, const float* gamma3,
                                   const float* mean3, const float* rstd3, const float* parts3, int nparts3, float* dgamma3,
                                   float* dbeta3, float* bn_workspace, const void* wt, long ldw, VtxBnBwdFusion* f2, void* dy2,
                                   float* dw_parts, long dw_parts_floats, int* dw_nparts, void* stream) {
    VTX_CHECK(dz && x3 && gamma3 && mean3 && rstd3 && parts3 && dgamma3 && dbeta3 && bn_workspace && wt && f2 && dy2 && dw_parts && dw_nparts,
              VTX_ERR_ARG, "conv3_bwd_fused: null pointer");
    VTX_CHECK(vtx_conv3_bwd_fused_supported(dtype, M, K, N), VTX_ERR_SHAPE,
              "conv3_bwd_fused: takes bf16, K = %d, N = %d, M a multiple of %d (>= %d); got M=%d K=%d N=%d", CB_K, CB_N, CB_RB, 4 * CB_RB, M, K, N);
    VTX_CHECK(f2->x && f2->mean && f2->rstd && f2->gamma && f2->beta && f2->parts && !f2->ymask && !f2->ybits, VTX_ERR_ARG,
              "conv3_bwd_fused: the bn2 fusion needs x, mean, rstd, gamma, beta, parts (mask recomputed from x)");
    VTX_CHECK(ldw % 8 == 0 && !(((uintptr_t)dz | (uintptr_t)x3 | (uintptr_t)wt | (uintptr_t)dy2 | (uintptr_t)f2->x | (uintptr_t)dw_parts) & 15),
              VTX_ERR_SHAPE, "conv3_bwd_fused: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int gx = vtx_conv3_bwd_fused_parts(M);
    VTX_CHECK((long)gx * K * N <= dw_parts_floats, VTX_ERR_WORKSPACE, "conv3_bwd_fused: dw_parts holds %ld floats, %ld needed", dw_parts_floats, (long)gx * K * N);
    VTX_CHECK((long)gx * 2 * N <= f2->parts_cap, VTX_ERR_WORKSPACE, "conv3_bwd_fused: bn2 parts hold %ld floats, %ld needed", f2->parts_cap, (long)gx * 2 * N);
    f2->strips = 0; *dw_nparts = 0;
    const float* coef = nullptr;
    int rc = vtx_bn_bwd_finalize_only(gamma3, rstd3, parts3, nparts3, dgamma3, dbeta3, bn_workspace, M, K, st, &coef);
    if (rc) return rc;
    auto kern = conv3_bwd_fused_kernel;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CB_LDS) != hipSuccess) {
        (void)hipGetLastError();
        vtx_set_error("conv3_bwd_fused: %d bytes of LDS refused", CB_LDS);
        return VTX_ERR_LAUNCH;
    }
    Conv3BwdArgs a{(const bf16_t*)dz, (const bf16_t*)x3, coef, mean3, rstd3, (const bf16_t*)wt, ldw,
                   (const bf16_t*)f2->x, f2->mean, f2->rstd, f2->gamma, f2->beta, (bf16_t*)dy2, f2->parts, dw_parts, M / CB_RB};
    // algorithmic work: two contractions (input + weight gradient); dz, x3, x2 read once, dy2 written once
    VTX_KLAUNCH("conv3_bwd_fused", 4.0 * M * K * N, 2.0 * M * (2.0 * K + 2.0 * N) + 2.0 * K * N + 4.0 * gx * K * N, kern, dim3(gx), dim3(CB_T), CB_LDS, st, a);
    VTX_LAUNCH_CHECK();
    f2->strips = gx; *dw_nparts = gx;
    return VTX_OK;
}

// C[M][N] (fp32) += sum over `nparts` partial matrices ws[p][M][N]: the fold of vtx_conv3_bwd_fused's weight-gradient partials
// (the split-K reduction of the weight-gradient GEMMs, exported so that the caller can put it on its weight-gradient stream).
extern "C" int vtx_partials_reduce_acc(const float* ws, int nparts, int M, int N, float* C, long ldc, void* stream) {
    VTX_CHECK(ws && C && nparts > 0 && M > 0 && N > 0 && N % 4 == 0 && ldc % 4 == 0, VTX_ERR_ARG, "partials_reduce_acc: bad arguments");
    vtx_splitk_reduce_now(ws, nparts, M, N, C, ldc, (hipStream_t)stream);     // caller-owned partials: never deferred into a reduction batch
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
