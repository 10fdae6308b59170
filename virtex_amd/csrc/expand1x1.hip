// Streaming kernel for the WRITE-HEAVY 1x1 "expand" convolutions of the Bottleneck (conv3: width -> 4*width channels,
// /root/reference/virtex/modules/visual_backbones.py:68-74 through torchvision's Bottleneck) with the BatchNorm
// statistics of the output in the epilogue:
//      Y[M][N] = A[M][K] * W[N][K]^T      K = 64 or 128,  N a multiple of 256,  M = pixels of the batch
// At bs = 256, 64 -> 256 @ 56x56 reads 103 MB and writes 411 MB.  tools/probes/write_probe.hip: the part moves exactly
// that traffic (1:4 read:write, non-temporal stores) in 73-77 us; the tiled contraction kernel needs 155 us for it,
// because every 128-row block is a chain of latencies (kernel arguments, first DMA, two K steps, epilogue, statistics
// fold) that only other resident blocks hide.  Here nothing is tiled over K at all:
//   * the weights of a 256-column block (256 x K bf16 = 32 / 64 KiB) are loaded into LDS ONCE per workgroup;
//   * a wave walks 16-row strips of A: its A fragments come straight from global memory into registers (a lane's
//     MFMA fragment is 16 contiguous bytes of one row), the NEXT strip's fragments are in flight while the current
//     strip is multiplied -- no barriers, no stages: waves are independent after the weight load;
//   * 16 rows x 256 columns leave through a wave-private LDS strip in two halves as 16-byte non-temporal stores of
//     whole 256-byte row segments; the lane that drains a chunk always drains the same 8 columns, so the BatchNorm
//     sums sum(y - shift), sum((y - shift)^2) of the STORED (rounded) values are 2 x 2 x 8 registers per lane;
//   * one statistics partial per workgroup (folded through LDS at the end): at most 512 strips, which the BatchNorm
//     finalize takes without a compaction launch.
// Entry: vtx_gemm_nt routes here (bf16, statistics requested, no bias / residual / activation, K in {64, 128},
// N % 256 == 0, N >= 4 K, M >= 4096) unless VIRTEX_AMD_EXPAND1X1=0.  Measured (profiles/r02_ab_expand1x1.txt):
// 64 -> 256 @ 56x56 163 -> 91 us (5.67 TB/s), 128 -> 512 @ 28x28 94 -> 67 us, the step 28.24 -> 27.93 ms.
#include <stdlib.h>

#include "vtx_common.h"

extern int g_vtx_sw_expand1x1;

namespace {

constexpr int EX_WAVES = 8, EX_NB = 256;                 // waves per workgroup, output columns per workgroup
constexpr int EX_ROWB = 128 * 2 + 16;                    // strip row: half of the 256 columns (256 B) + pad

// 16-byte slot of (row n, k-slot s) inside the weight image: rows of 64 k (128 B), the eight slots of a row XOR-permuted
// by (n >> 1) & 7 -- every 16-lane group of a ds_read_b128 (16 rows at one logical slot) covers all 64 banks.
__device__ __forceinline__ int wslot(int n, int s) { return n * 8 + (s ^ ((n >> 1) & 7)); }

template <int K>
__global__ __launch_bounds__(64 * EX_WAVES, 2) void expand1x1_fwd_kernel(
    const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw, bf16_t* __restrict__ Y, long ldy,
    const float* __restrict__ shift, float* __restrict__ parts, int M, int N, int nt_store) {
    constexpr int KS = K / 32;                           // MFMA k-steps per strip
    constexpr int KH = K / 64;                           // 64-wide halves of a weight row
    HIP_DYNAMIC_SHARED(char, smem)
    bf16_t* wimg = reinterpret_cast<bf16_t*>(smem);                        // [KH][256][64] swizzled
    char* strips = smem + (size_t)EX_NB * K * 2;                           // [EX_WAVES][16][EX_ROWB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.y * EX_NB;

    // ---- the workgroup's weights -> LDS (once)
    for (int c = tid; c < EX_NB * (K / 8); c += 64 * EX_WAVES) {
        const int n = c / (K / 8), s = c % (K / 8);                        // row, 16-byte k-slot
        const uint4 v = *reinterpret_cast<const uint4*>(W + (long)(n0 + n) * ldw + s * 8);
        *reinterpret_cast<uint4*>(wimg + (size_t)(s >> 3) * EX_NB * 64 + wslot(n, s & 7) * 8) = v;
    }
    // per-lane statistics: this lane drains column chunk (lane % 16) of each half: 8 columns, two halves
    float sh[2][8], s1[2][8], s2[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s1[h][e] = s2[h][e] = 0.f;
            sh[h][e] = shift ? shift[n0 + h * 128 + (lane & 15) * 8 + e] : 0.f;
        }
    __syncthreads();

    const int nstrips = (M + 15) / 16;
    const int stride = gridDim.x * EX_WAVES;
    char* strip = strips + (size_t)wave * 16 * EX_ROWB;
    // A fragment of strip s, k-step ks: lane (row = lane & 15, slot = lane >> 4) holds k = ks*32 + slot*8 .. +7
    auto load_a = [&](int s, bf16x8_t* f) {
        int row = s * 16 + (lane & 15);
        row = row < M ? row : M - 1;                                       // rows past M: any valid row (discarded)
        const bf16_t* p = A + (long)row * lda + (lane >> 4) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) f[ks] = *reinterpret_cast<const bf16x8_t*>(p + ks * 32);
    };
    // (Round 3, session 12: the drain as straight-line code -- compile-time FULL / NT variants, so that hipcc counts the stores
    //  instead of guarding the prefetched fragments with vmcnt(0) -- plus a pinned hand-over `fa = fn` at the end of the trip
    //  was built and measured: this kernel unchanged, the stem's twin 140 -> 169 us.  These kernels are throughput-bound with
    //  16 waves per CU; a wave waiting for its own stores costs nothing.  Reverted.)
    bf16x8_t fa[KS], fn[KS];
    int s = blockIdx.x * EX_WAVES + wave;
    if (s < nstrips) load_a(s, fa);
    for (; s < nstrips; s += stride) {
        const bool more = s + stride < nstrips;
        if (more) load_a(s + stride, fn);                                  // in flight under this strip's work
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                      // 128 output columns at a time
            f32x4_t acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = h * 128 + j * 16 + (lane & 15);
                    const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(
                        wimg + (size_t)(ks >> 1) * EX_NB * 64 + wslot(n, (ks & 1) * 4 + (lane >> 4)) * 8);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[ks], acc[j], 0, 0, 0);
                }
            }
            // lane holds rows m = lane & 15, columns j*16 + 4*(lane>>4) .. +3 of this half
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<uint2*>(strip + (lane & 15) * EX_ROWB + (j * 16 + 4 * (lane >> 4)) * 2) =
                    make_uint2(f2bf2(acc[j][0], acc[j][1]), f2bf2(acc[j][2], acc[j][3]));
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                  // 16 rows x 16 chunks = 256 chunks / 64 lanes
                const int r = (lane >> 4) + 4 * q, ch = lane & 15;
                const uint4 w = *reinterpret_cast<const uint4*>(strip + r * EX_ROWB + ch * 16);
                const int m = s * 16 + r;
                if (m < M) {
                    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d0 = __uint_as_float(u[e] << 16) - sh[h][2 * e];
                        const float d1 = __uint_as_float(u[e] & 0xffff0000u) - sh[h][2 * e + 1];
                        s1[h][2 * e] += d0; s2[h][2 * e] += d0 * d0;
                        s1[h][2 * e + 1] += d1; s2[h][2 * e + 1] += d1 * d1;
                    }
                    bf16_t* dst = Y + (long)m * ldy + n0 + h * 128 + ch * 8;
                    if (nt_store) st16_nt(dst, u32x4_t{w.x, w.y, w.z, w.w});
                    else *reinterpret_cast<uint4*>(dst) = w;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (more) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fa[ks] = fn[ks];
        }
    }

    // ---- statistics: lanes that drained the same column chunk (they differ in lane >> 4), then the waves
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s1[h][e] += __shfl_xor(s1[h][e], 16, 64); s1[h][e] += __shfl_xor(s1[h][e], 32, 64);
            s2[h][e] += __shfl_xor(s2[h][e], 16, 64); s2[h][e] += __shfl_xor(s2[h][e], 32, 64);
        }
    __syncthreads();                                                       // every wave is done with its strip
    float* red = reinterpret_cast<float*>(strips);                         // [EX_WAVES][2][256]
    if (lane < 16) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 2 + 0) * EX_NB + h * 128 + lane * 8 + e] = s1[h][e];
                red[(wave * 2 + 1) * EX_NB + h * 128 + lane * 8 + e] = s2[h][e];
            }
    }
    __syncthreads();
    float* dst = parts + (size_t)blockIdx.x * 2 * N;
    for (int t = tid; t < 2 * EX_NB; t += 64 * EX_WAVES) {
        const int which = t / EX_NB, c = t % EX_NB;
        float a = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < EX_WAVES; ++w2) a += red[(w2 * 2 + which) * EX_NB + c];
        dst[(size_t)which * N + n0 + c] = a;
    }
}

template <int K>
int launch_expand(const void* A, long lda, const void* W, long ldw, void* Y, long ldy, const float* shift, float* parts,
                  int M, int N, int nt_store, hipStream_t st) {
    const size_t lds = (size_t)EX_NB * K * 2 + (size_t)EX_WAVES * 16 * EX_ROWB;       // 66 / 98 KiB
    auto kern = expand1x1_fwd_kernel<K>;
    // per device, and cheap: set on every call (a process may drive several GPUs); failure -> the tiled kernel takes the problem
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const int nstrips = (M + 15) / 16;
    int gx = 256 * (K == 64 ? 2 : 1);                                     // one workgroup per resident slot of the chip
    if (gx * EX_WAVES > nstrips) gx = (nstrips + EX_WAVES - 1) / EX_WAVES;
    if (gx > M / 64) gx = M / 64;                                         // the caller's partial buffer holds ceil(M/64)+4 strips
    dim3 grid(gx, N / EX_NB), block(64 * EX_WAVES);
    VTX_KLAUNCH("expand1x1_fwd", 2.0 * M * N * K, 2.0 * ((double)M * K * (N / EX_NB) + (double)N * K + (double)M * N), kern, grid,
                block, lds, st, (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (bf16_t*)Y, ldy, shift, parts, M, N, nt_store);
    return gx;
}

}  // namespace

// Returns the number of statistics strips written (> 0) when the problem was taken, 0 when it is not this kernel's
// (the caller then uses the tiled contraction kernel), < 0 on a launch error.
int vtx_expand1x1_try(int M, int N, int K, const void* A, long lda, const void* W, long ldw, void* Y, long ldy,
                      const float* shift, float* parts, hipStream_t st) {
    if (!g_vtx_sw_expand1x1 || !parts || (K != 64 && K != 128) || N % EX_NB != 0 || N < 4 * K || M < 4096 || ldy != N) return 0;
    if ((lda % 8) || (ldw % 8) || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)Y & 15)) return 0;
    const int nt = (double)M * N * 2 >= 200e6;
    const int strips = K == 64 ? launch_expand<64>(A, lda, W, ldw, Y, ldy, shift, parts, M, N, nt, st)
                               : launch_expand<128>(A, lda, W, ldw, Y, ldy, shift, parts, M, N, nt, st);
    if (hipGetLastError() != hipSuccess) { vtx_set_error("expand1x1: launch failed"); return VTX_ERR_LAUNCH; }
    return strips;
}
