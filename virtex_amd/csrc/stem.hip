// Streaming kernel for the stem convolution (7x7 / stride 2 / pad 3, 3 -> 64 channels: torchvision ResNet.conv1, reached
// from /root/reference/virtex/modules/visual_backbones.py:68-74) on the PACKED input layout of vtx_image_to_nhwc_halo
// (4-channel pixels, a 3-pixel zero frame, the filter zero-padded to 7x8: a "valid" 7x8 / stride-2 convolution whose
// 16-byte chunks hold two horizontally adjacent pixels), with the BatchNorm statistics of the output in the epilogue:
//      Y[m][ko] = sum_{kh < 7} sum_{j < 32} X[n][2 oh + kh][2 ow .. 2 ow + 7][0..3] (j-th of those 32 values) * Wt[ko][kh][j]
// At bs = 256 it reads 108 MB and writes 411 MB -- the same write-heavy shape as the Bottleneck's 'expand' convolutions
// (expand1x1.hip), and the same structure serves it: NOTHING is tiled over K.
//   * the 64 x 224 filter (28 KiB) sits in LDS for the whole life of the workgroup, XOR-swizzled for ds_read_b128;
//   * a wave walks 16-pixel strips of the output: the seven A fragments of a strip (one per filter row: 16 contiguous
//     bytes = the taps (kw, kw+1) of one pixel pair, at input row 2 oh + kh) come straight from global memory into
//     registers, the next strip's fragments are in flight under the current strip's 28 MFMAs; no barriers after the
//     filter load;
//   * the 16 x 64 results leave through a wave-private LDS strip as 16-byte non-temporal stores -- a strip is 2 KiB of
//     CONTIGUOUS output -- and the lane that drains a chunk always drains the same 8 channels, so the BatchNorm sums
//     sum(y - shift), sum((y - shift)^2) of the stored (rounded) values are 2 x 8 registers per lane;
//   * one statistics partial per workgroup (<= 512 strips: the BatchNorm finalize takes them without compaction).
// The tiled contraction kernel needs 261 us for this layer (M = 3.2 M rows, N = 64, K = 224: 12 544 blocks whose whole
// life is a prologue and an epilogue) and leaves the statistics to a stand-alone pass over the 411 MB (96 us).
// Entry: vtx_conv2d_fwd routes here (bf16, C = 4, 7x8 / s2 / p0, KO = 64, statistics requested) unless
// VIRTEX_AMD_STEM_STREAM=0.
#include <stdlib.h>

#include "vtx_common.h"

extern int g_vtx_sw_stem_stream;

namespace {

constexpr int ST_WAVES = 8, ST_N = 64, ST_KH = 7;
constexpr int ST_ROWB = ST_N * 2 + 16;                   // strip row: 64 channels (128 B) + pad

// 16-byte slot (row n, k-slot s of filter row kh): rows of 32 k (64 B), the four slots of a row XOR-permuted per row
// quad so that every 16-lane group of a ds_read_b128 (rows r..r+15 at one or two logical slots) covers all 64 banks
// (the permutation of the contraction kernel's 32-deep images: gemm_kernel.h swz_slot)
__device__ __forceinline__ int st_wslot(int kh, int n, int s) { return (kh * ST_N + n) * 4 + (s ^ ((0x78 >> (2 * ((n >> 2) & 3))) & 3)); }

__device__ __forceinline__ int st_qdiv(int n, int d) { return vtx_fdiv30(n, d, __builtin_amdgcn_rcpf((float)d)); }   // exact n / d for 0 <= n < 2^30, d > 0

__global__ __launch_bounds__(64 * ST_WAVES, 4) void stem_stream_fwd_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wt, bf16_t* __restrict__ Y, const float* __restrict__ shift,
    float* __restrict__ parts, int M, int H, int W, int OH, int OW, int nt_store, int xcd_major) {
    HIP_DYNAMIC_SHARED(char, smem)
    bf16_t* wimg = reinterpret_cast<bf16_t*>(smem);                        // [7][64][32] swizzled
    char* strips = smem + (size_t)ST_KH * ST_N * 32 * 2;                   // [ST_WAVES][16][ST_ROWB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- the filter -> LDS (once): Wt[ko][kh][32]
    for (int c = tid; c < ST_N * ST_KH * 4; c += 64 * ST_WAVES) {
        const int n = c / (ST_KH * 4), r = c % (ST_KH * 4), kh = r >> 2, s = r & 3;
        const uint4 v = *reinterpret_cast<const uint4*>(Wt + (long)n * (ST_KH * 32) + kh * 32 + s * 8);
        *reinterpret_cast<uint4*>(wimg + (size_t)st_wslot(kh, n, s) * 8) = v;
    }
    // per-lane statistics: this lane drains column chunk lane & 7 (8 channels) of every row it touches
    float sh[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = s2[e] = 0.f; sh[e] = shift ? shift[(lane & 7) * 8 + e] : 0.f; }
    __syncthreads();

    const int nstrips = (M + 15) / 16;
    const int stride = gridDim.x * ST_WAVES;
    const long rowpitch = (long)W * 4;                                     // elements per input row (4-channel pixels)
    char* strip = strips + (size_t)wave * 16 * ST_ROWB;
    // A fragment of strip s, filter row kh: lane (pixel = lane & 15, slot = lane >> 4) holds the 8 values of the pixel
    // pair (2 ow + 2 slot, 2 ow + 2 slot + 1) of input row 2 oh + kh
    auto load_a = [&](int s, bf16x8_t* f) {
        int m = s * 16 + (lane & 15);
        m = m < M ? m : M - 1;                                             // rows past M: any valid pixel (discarded)
        const int n = st_qdiv(m, OH * OW), rem = m - n * OH * OW;
        const int oh = st_qdiv(rem, OW), ow = rem - oh * OW;
        const bf16_t* p = X + (((long)n * H + 2 * oh) * W + 2 * ow + 2 * (lane >> 4)) * 4;
#pragma unroll
        for (int kh = 0; kh < ST_KH; ++kh) f[kh] = *reinterpret_cast<const bf16x8_t*>(p + kh * rowpitch);
    };
    bf16x8_t fa[ST_KH], fn[ST_KH];
    // Which strips a workgroup walks.  Workgroups are dealt to the eight XCDs round-robin (XCD = blockIdx.x & 7) and every XCD has an
    // L2 of its own: with strip s on workgroup (s / ST_WAVES) % gridDim.x, vertically adjacent strips -- 7 strips apart, sharing five
    // of their seven input rows -- sit on different XCDs and every XCD fetched (almost) the whole input: PMC 370 MB for a 108-MB
    // image tensor (profiles/r05_traffic_ratio.txt, 1.50 x the kernel's algorithmic bytes).  XCD-major order instead: in every round
    // XCD x owns ONE contiguous band of (gridDim.x / 8) * ST_WAVES strips, so an input row is fetched by one L2 (plus band edges).
    int s = blockIdx.x * ST_WAVES + wave;
    if (xcd_major && (gridDim.x & 7) == 0) s = ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * ST_WAVES + wave;
    if (s < nstrips) load_a(s, fa);
    for (; s < nstrips; s += stride) {
        const bool more = s + stride < nstrips;
        if (more) load_a(s + stride, fn);                                  // in flight under this strip's work
        f32x4_t acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < ST_KH; ++kh) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(wimg + (size_t)st_wslot(kh, j * 16 + (lane & 15), lane >> 4) * 8);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[kh], acc[j], 0, 0, 0);
            }
        }
        // lane holds pixel lane & 15, channels j*16 + 4*(lane>>4) .. +3
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<uint2*>(strip + (lane & 15) * ST_ROWB + (j * 16 + 4 * (lane >> 4)) * 2) =
                make_uint2(f2bf2(acc[j][0], acc[j][1]), f2bf2(acc[j][2], acc[j][3]));
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 2; ++q) {                                      // 16 pixels x 8 chunks = 128 chunks / 64 lanes
            const int r = (lane >> 3) + 8 * q, ch = lane & 7;
            const uint4 w = *reinterpret_cast<const uint4*>(strip + r * ST_ROWB + ch * 16);
            const int m = s * 16 + r;
            if (m < M) {
                const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = __uint_as_float(u[e] << 16) - sh[2 * e];
                    const float d1 = __uint_as_float(u[e] & 0xffff0000u) - sh[2 * e + 1];
                    s1[2 * e] += d0; s2[2 * e] += d0 * d0;
                    s1[2 * e + 1] += d1; s2[2 * e + 1] += d1 * d1;
                }
                bf16_t* dst = Y + (long)m * ST_N + ch * 8;
                if (nt_store) st16_nt(dst, u32x4_t{w.x, w.y, w.z, w.w});
                else *reinterpret_cast<uint4*>(dst) = w;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (more) {
#pragma unroll
            for (int kh = 0; kh < ST_KH; ++kh) fa[kh] = fn[kh];
        }
    }

    // ---- statistics: lanes that drained the same column chunk (they differ in lane >> 3), then the waves
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int msk = 8; msk < 64; msk <<= 1) { s1[e] += __shfl_xor(s1[e], msk, 64); s2[e] += __shfl_xor(s2[e], msk, 64); }
    }
    __syncthreads();                                                       // every wave is done with its strip
    float* red = reinterpret_cast<float*>(strips);                         // [ST_WAVES][2][64]
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(wave * 2 + 0) * ST_N + lane * 8 + e] = s1[e];
            red[(wave * 2 + 1) * ST_N + lane * 8 + e] = s2[e];
        }
    }
    __syncthreads();
    float* dst = parts + (size_t)blockIdx.x * 2 * ST_N;
    for (int t = tid; t < 2 * ST_N; t += 64 * ST_WAVES) {
        const int which = t / ST_N, c = t % ST_N;
        float a = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < ST_WAVES; ++w2) a += red[(w2 * 2 + which) * ST_N + c];
        dst[(size_t)which * ST_N + c] = a;
    }
}

}  // namespace

// Returns the number of statistics strips written (> 0) when the problem was taken, 0 when it is not this kernel's
// (the caller then uses the tiled contraction kernel), < 0 on a launch error.  x: [N][H][W][4] (frame included),
// w: [64][7][8][4], y: [N][OH][OW][64], OH = (H - 7) / 2 + 1, OW = (W - 8) / 2 + 1.
int vtx_stem_stream_try(int N, int H, int W, int C, int KO, int R, int S, int stride, int pad, const void* x, const void* w,
                        void* y, const float* shift, float* parts, hipStream_t st) {
    if (!g_vtx_sw_stem_stream || !parts || C != 4 || KO != ST_N || R != ST_KH || S != 8 || stride != 2 || pad != 0 || (W & 1)) return 0;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15)) return 0;
    const int OH = (H - R) / 2 + 1, OW = (W - S) / 2 + 1;
    if (OH <= 0 || OW <= 0) return 0;
    const long Ml = (long)N * OH * OW;
    if (Ml < 256 || Ml >= VTX_PIXEL_LIMIT) return 0;
    const int M = (int)Ml;
    const size_t lds = (size_t)ST_KH * ST_N * 32 * 2 + (size_t)ST_WAVES * 16 * ST_ROWB;    // 28 + 18 KiB
    const int nstrips = (M + 15) / 16;
    int gx = 512;                                                          // two workgroups per CU
    if (gx * ST_WAVES > nstrips) gx = (nstrips + ST_WAVES - 1) / ST_WAVES;
    if (gx > M / 64) gx = M / 64;                                          // the caller's partial buffer holds ceil(M/64)+4 strips
    const int nt = (double)M * ST_N * 2 >= 200e6;
    dim3 grid(gx), block(64 * ST_WAVES);
    VTX_KLAUNCH("stem_stream_fwd", 2.0 * M * ST_N * (ST_KH * 32), 2.0 * ((double)N * H * W * 4 + (double)ST_N * ST_KH * 32 + (double)M * ST_N),
                stem_stream_fwd_kernel, grid, block, lds, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, shift, parts, M, H, W, OH, OW, nt,
                g_vtx_sw_stem_stream != 2 /* vtx_set_switch("stem_stream", 2): the plain strip order (A/B) */);
    if (hipGetLastError() != hipSuccess) { vtx_set_error("stem_stream_fwd: launch failed"); return VTX_ERR_LAUNCH; }
    return gx;
}
