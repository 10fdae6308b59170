// Streaming weight gradient of the 3x3 / stride-1 / pad-1 convolutions (conv2 of every stride-1 Bottleneck: 13 of the 16
// 3x3 convolutions of ResNet-50, /root/reference/virtex/modules/visual_backbones.py:68-74 through torchvision):
//      dw[ko][r][s][c] += sum_{n,oh,ow} dy[n][oh][ow][ko] * x[n][oh + r - 1][ow + s - 1][c]
// The implicit-GEMM formulation (conv_wgrad.hip: M = KO, N = 9 C, K = pixels) re-reads x once per filter tap and dy once
// per 128-column tile of (tap, c) through L2 -- 64 -> 64 @ 56x56 moves 1.4 GB from L2 to LDS for 206 MB of operands and
// runs at 308 TFLOP/s.  Here the contraction index is a PADDED-LINEAR pixel index
//      p' = (n * (H + 2) + ih + 1) * (W + 2) + iw + 1          (one zero pixel around every image)
// in which every filter tap is a CONSTANT shift:  dw[ko][r][s][c] = sum_{p'} dyp[p'][ko] * xp[p' + (r-1)(W+2) + (s-1)][c]
// (dyp / xp = the tensors with zeros at the padding positions; shifts never leave an image's padded block).  So
//   * a workgroup owns a (64 ko) x (9 taps) x (64 c) block of dw in registers (72 fp32 accumulators per lane, 8 waves)
//     and STREAMS a contiguous range of p' through LDS: 32 positions of dyp and 32 of xp per step arrive by LDS-DMA
//     (buffer_load ... lds, one 1-KiB instruction per wave and step; padding positions are out-of-range lanes: the
//     hardware writes the zeros) into two rings -- every operand byte is fetched ONCE per (ko-chunk, c-chunk) pair;
//   * the nine taps read their B fragments from the SAME xp ring at nine constant row offsets (ds_read_b64_tr_b16:
//     the rings are k-major exactly as in HBM, no transposition work), the A fragments (dyp) are shared by the taps:
//     22 transposing reads for 18 MFMAs per wave and step, two address VALU per read (the swizzle of a ring row
//     depends on bits 1 and 3 of its index, which a step of 32 rows does not change: per-tap lane constants);
//   * loads run W3_PF = 8 steps (64 KiB per CU) ahead of the MFMAs behind a counted s_waitcnt and ONE barrier per TWO steps;
//     a loading lane tracks its padded-linear position with carries (no division in the loop);
//   * the pixel ranges' partial blocks go to the split-K workspace as [range][KO][9][C] and the existing reduce adds
//     them into dw.
// MFMA work grows by (H+2)(W+2)/(HW) (7 % at 56x56, 15 % at 28x28, 31 % at 14x14, 65 % at 7x7: the padding positions
// are multiplied as zeros).
// Entry: vtx_conv2d_wgrad routes here (bf16, 3x3 / s1 / p1, C and KO multiples of 64, 7 <= W <= 61, by default H, W >= 28) unless
// VIRTEX_AMD_WGRAD3X3=0.
#include <stdlib.h>

#include "vtx_common.h"

void vtx_splitk_reduce(const float* ws, int S, int M, int N, float* C, long ldc, hipStream_t st);
extern int g_vtx_sw_wgrad3x3;

namespace {

constexpr int W3_WAVES = 8;
constexpr int W3_RING = 512;             // xp ring: positions (rows of 64 channels = 128 B)  -> 64 KiB at LDS offset 0
constexpr int W3_DS = 16;                // dyp ring: slots of 32 positions (4 KiB each)       -> 64 KiB behind it
constexpr int W3_PF = 8;                 // steps of prefetch
constexpr uint32_t W3_OOB = 0x80000000u;

// 16-byte chunk permutation of a k-major row of 64 elements (gemm_kernel.h swz_mc<64>): keyed on bits 1 and 3 of the row index
__device__ __forceinline__ int w3_swz(int chunk, int k) { return chunk ^ ((((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2)); }
__device__ __forceinline__ int w3_qdiv(int n, int d, float inv) { return vtx_fdiv30(n, d, inv); }   // exact floor(n / d) for 0 <= n < 2^30

struct W3Geo { int N, H, W, C, KO, Hp, Wp, P; float inv_img, inv_wp; };

// byte offset (or OOB) of 16 bytes of padded-linear position q, channel chunk `el0` (first element), in a [N][H][W][CH] tensor
__device__ __forceinline__ uint32_t w3_voff(const W3Geo& g, int q, int CH, int el0) {
    if (q < 0 || q >= g.P) return W3_OOB;
    const int n = w3_qdiv(q, g.Hp * g.Wp, g.inv_img), rem = q - n * g.Hp * g.Wp;
    const int ihp = w3_qdiv(rem, g.Wp, g.inv_wp), iwp = rem - ihp * g.Wp;
    if (ihp < 1 || ihp > g.H || iwp < 1 || iwp > g.W) return W3_OOB;
    return (uint32_t)((((n * g.H + ihp - 1) * g.W + iwp - 1) * CH + el0) * 2);
}

template <int T>
__device__ __forceinline__ void w3_taps(vtx_v4s_t (&ra)[2][2], vtx_v4s_t (&rb)[9][2], bf16x8_t (&fa)[2], f32x4_t (&acc)[2][9]) {
    if constexpr (T < 9) {
        vtx_ds_tr_wait_n<(16 - 2 * T > 15 ? 15 : 16 - 2 * T)>();
        if constexpr (T == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = __builtin_shufflevector(ra[i][0], ra[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
        }
        const bf16x8_t fb = __builtin_shufflevector(rb[T][0], rb[T][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[i], acc[i][T], 0, 0, 0);
        w3_taps<T + 1>(ra, rb, fa, acc);
    }
}

__global__ __launch_bounds__(64 * W3_WAVES, 2) void conv3x3_wgrad_stream_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ DY, float* __restrict__ WS, W3Geo g, int lead, int nl,
    int steps_per_range, int total_steps, int npairs) {
    HIP_DYNAMIC_SHARED(char, smem)
#ifndef HIPEMU
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();   // the rings' offsets ARE their LDS addresses
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: the (ko, c) chunk pairs of one pixel range share their operands -> contiguous on one XCD
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int range = wid / npairs, pair = wid - range * npairs;
    const int cchunks = g.C >> 6;
    const int ko0 = (pair / cchunks) * 64, c0 = (pair % cchunks) * 64;
    const int step0 = range * steps_per_range;
    int nsteps = total_steps - step0;
    nsteps = nsteps < steps_per_range ? nsteps : steps_per_range;
    const int p0 = step0 * 32;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), (short)0, (int)((long)g.N * g.H * g.W * g.C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(DY), (short)0, (int)((long)g.N * g.H * g.W * g.KO * 2), 0x00020000);

    // ---- loader role: waves 0-3 stage xp (8 positions each per step), waves 4-7 dyp.  A lane stays on ONE position of
    // the 32-position step and one 16-byte chunk; its padded-linear position advances by 32 per step, tracked as
    // (image, padded row, padded column) with carries -- no division in the loop.
    const bool xrole = wave < 4;
    const int sub = wave & 3;
    const int lpos = 8 * sub + (lane >> 3);                                 // position within the 32-position step
    const int src_chunk = w3_swz(lane & 7, lpos);                           // bits 1, 3 of the ring row = those of lpos
    const int CH = xrole ? g.C : g.KO, el0 = (xrole ? c0 : ko0) + 8 * src_chunk;
    int q = (xrole ? p0 - lead : p0) + lpos;                                // position of this lane's next piece
    int qn, qh, qw;                                                         // its (image, padded row, padded column); q < 0: unused
    {
        const int qq = q < 0 ? 0 : q;
        qn = w3_qdiv(qq, g.Hp * g.Wp, g.inv_img);
        const int rem = qq - qn * g.Hp * g.Wp;
        qh = w3_qdiv(rem, g.Wp, g.inv_wp); qw = rem - qh * g.Wp;
    }
    const uint32_t dst0 = xrole ? (uint32_t)(8 * sub) << 7 : (uint32_t)(W3_RING * 128 + (sub << 10));
    int sissue = 0;                                                         // steps issued so far
    auto issue = [&]() {
        const bool ok = q >= 0 && q < g.P && qh >= 1 && qh <= g.H && qw >= 1 && qw <= g.W;
        const uint32_t vo = ok ? (uint32_t)((((qn * g.H + qh - 1) * g.W + qw - 1) * CH + el0) * 2) : W3_OOB;
        const uint32_t dst = xrole ? dst0 + ((uint32_t)((32 * sissue) & (W3_RING - 1)) << 7) : dst0 + ((uint32_t)(sissue & (W3_DS - 1)) << 12);
        if (xrole) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(smem + dst), 16, (int)vo, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (__attribute__((address_space(3))) void*)(smem + dst), 16, (int)vo, 0, 0, 0);
        ++sissue;
        const bool was_neg = q < 0;
        q += 32;
        if (was_neg) {                                                      // the lead-in of the first range: re-derive once q >= 0
            if (q >= 0) {
                qn = w3_qdiv(q, g.Hp * g.Wp, g.inv_img);
                const int rem = q - qn * g.Hp * g.Wp;
                qh = w3_qdiv(rem, g.Wp, g.inv_wp); qw = rem - qh * g.Wp;
            }
        } else {
            qw += 32;                                                       // Wp >= 9: at most four row carries per step
#pragma unroll
            for (int i = 0; i < 4; ++i) { const bool cy = qw >= g.Wp; qw -= cy ? g.Wp : 0; qh += cy ? 1 : 0; }
            const bool ci = qh >= g.Hp; qh -= ci ? g.Hp : 0; qn += ci ? 1 : 0;
        }
    };

    // ---- compute role: wave = (ko half, c fragment): 2 ko fragments x 1 c fragment x 9 taps
    const int koh = wave >> 2, cw = wave & 3;
    const int w = lane & 15, gq = lane >> 4;
    const int kk = 8 * gq + (w >> 2);                                       // k row (within a 32-step) this lane addresses
    // A (dyp) fragment addresses inside a slot: [k][64 ko] rows of 128 B
    uint32_t offA[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = 16 * (2 * koh + i) + 4 * (w & 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = kk + 4 * h;
            offA[i][h] = (uint32_t)(W3_RING * 128 + k * 128 + w3_swz(rr >> 3, k) * 16 + (rr & 7) * 2);
        }
    }
    // B (xp) fragment addresses: ring row (lead + 32 t + delta_tap + kk [+4]) -- per-tap lane constants
    uint32_t offB[9][2];
    {
        const int rr = 16 * cw + 4 * (w & 3);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int delta = (tap / 3 - 1) * g.Wp + (tap % 3 - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = (lead + delta + kk + 4 * h) & (W3_RING - 1);
                offB[tap][h] = (uint32_t)(k * 128 + w3_swz(rr >> 3, k) * 16 + (rr & 7) * 2);
            }
        }
    }

    f32x4_t acc[2][9];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) acc[i][tap] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    constexpr int WAIT_PF = (W3_PF & 0xF) | ((W3_PF >> 4) << 14) | (0x7 << 4) | (0xF << 8);
    const int xahead = nl + W3_PF;
    for (int s = 0; s < (xrole ? xahead : W3_PF); ++s) issue();
    // TWO steps per barrier (the ring margins are sized for it: host check): per pair two loads per wave are issued, the
    // wait leaves the W3_PF youngest in flight (x-steps <= t + 1 + nl and dy-steps <= t + 1 have landed), one barrier,
    // then 2 x (22 transposing reads + 18 MFMAs)
    for (int t = 0; t < nsteps; t += 2) {
        issue(); issue();                           // beyond the range's needs: out-of-range lanes or never-read slots
        __builtin_amdgcn_s_waitcnt(WAIT_PF);
        __builtin_amdgcn_s_barrier();               // everybody's pieces have landed; all waves are done with the previous pair
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && t + 1 >= nsteps) break;   // odd tail: the positions of step t + 1 belong to the next range
            const uint32_t slot = (uint32_t)((t + u) & (W3_DS - 1)) << 12;
            const uint32_t rb = (uint32_t)((32 * (t + u)) & (W3_RING - 1)) << 7;
            // all 22 transposing reads of the step first (asm: see vtx_ds_read_tr16), then tap by tap: tap T multiplies as
            // soon as ITS two reads have returned (LDS returns in order: all but the 16 - 2 T youngest; lgkmcnt holds at
            // most 15) -- the later taps' reads land under the earlier taps' MFMAs.  The xp ring is 64 KiB at LDS address
            // 0: a 16-bit add wraps by itself.
            vtx_v4s_t ra[2][2], rb2[9][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ra[i][0] = vtx_ds_read_tr16_at(smem, offA[i][0] + slot);
                ra[i][1] = vtx_ds_read_tr16_at(smem, offA[i][1] + slot);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                rb2[tap][0] = vtx_ds_read_tr16_at(smem, (uint32_t)(uint16_t)((uint16_t)offB[tap][0] + (uint16_t)rb));
                rb2[tap][1] = vtx_ds_read_tr16_at(smem, (uint32_t)(uint16_t)((uint16_t)offB[tap][1] + (uint16_t)rb));
            }
            bf16x8_t fa[2];
            w3_taps<0>(ra, rb2, fa, acc);
        }
    }
    // lane holds dw[ko = .. + (lane & 15)][tap][c = .. + 4 (lane >> 4) + 0..3]
    float* out = WS + (size_t)range * g.KO * 9 * g.C;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ko = ko0 + 16 * (2 * koh + i) + (lane & 15);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            *reinterpret_cast<float4*>(out + ((size_t)ko * 9 + tap) * g.C + c0 + 16 * cw + 4 * (lane >> 4)) =
                make_float4(acc[i][tap][0], acc[i][tap][1], acc[i][tap][2], acc[i][tap][3]);
    }
}

}  // namespace

// Returns 1 when the problem was taken (dw += the gradient, partial sums through `ws`), 0 when it is not this kernel's
// (the caller then uses the implicit-GEMM kernel), < 0 on a launch error.
int vtx_conv3x3_wgrad_try(int N, int H, int W, int C, int KO, int R, int S, int stride, int pad, const void* x, const void* dy,
                          float* dw, float* ws, long ws_floats, hipStream_t st) {
    const int on = g_vtx_sw_wgrad3x3;
    if (!on || R != 3 || S != 3 || stride != 1 || pad != 1 || (C & 63) || (KO & 63) || !ws) return 0;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dw & 15) || ((uintptr_t)ws & 15)) return 0;
    const int Hp = H + 2, Wp = W + 2;
    const long P = (long)N * Hp * Wp;
    const int lead = ((Wp + 1 + 31) / 32) * 32, nl = 2 * lead / 32;
    if (P >= VTX_PIXEL_LIMIT || W3_RING / 32 < nl + W3_PF + 4 || Wp < 9 || Hp < 5) return 0;   // ring margins for two steps per barrier (W <= 61);
                                                                                         // the loader carries at most ONE image per 32-position step (<= 4 rows of >= 9): Hp >= 5
    // by image size (switch value 1): the padding positions are multiplied as zeros -- +7 % MFMA work at 56x56, +15 % at
    // 28x28, +31 % at 14x14, +65 % at 7x7, where the implicit-GEMM kernel (whose operands then fit the caches) wins
    if (on == 1 && (H < 28 || W < 28)) return 0;
    if ((double)N * H * W * C * 2 >= 2.0e9 || (double)N * H * W * KO * 2 >= 2.0e9) return 0;
    const int npairs = (KO / 64) * (C / 64);
    const int total_steps = (int)((P + 31) / 32);
    int nranges = 256 / npairs;                                           // one workgroup per CU
    if (nranges < 1) nranges = 1;
    if (nranges > total_steps / 8) nranges = total_steps / 8 > 0 ? total_steps / 8 : 1;     // at least 8 steps per range
    const int spr = (total_steps + nranges - 1) / nranges;
    nranges = (total_steps + spr - 1) / spr;
    const long slice = (long)KO * 9 * C;
    if ((long)nranges * slice > ws_floats) return 0;
    W3Geo g{N, H, W, C, KO, Hp, Wp, (int)P, 1.0f / (float)(Hp * Wp), 1.0f / (float)Wp};
    const size_t lds = (size_t)W3_RING * 128 + (size_t)W3_DS * 4096;       // 128 KiB
    auto kern = conv3x3_wgrad_stream_kernel;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    dim3 grid(nranges * npairs), block(64 * W3_WAVES);
    const double pixels = (double)N * H * W;
    VTX_KLAUNCH("conv3x3_wgrad_stream", 2.0 * pixels * 9 * C * KO, 2.0 * pixels * (C + KO) + 4.0 * nranges * slice, kern, grid, block, lds, st,
                (const bf16_t*)x, (const bf16_t*)dy, ws, g, lead, nl, spr, total_steps, npairs);
    if (hipGetLastError() != hipSuccess) { vtx_set_error("conv3x3_wgrad_stream: launch failed"); return VTX_ERR_LAUNCH; }
    vtx_splitk_reduce(ws, nranges, KO, 9 * C, dw, 9 * (long)C, st);
    return 1;
}
