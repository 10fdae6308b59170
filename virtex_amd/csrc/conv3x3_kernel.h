// 3x3 / stride-1 / pad-1 convolution (forward, and -- with the filter taps flipped -- the input gradient) as an implicit
// GEMM whose A operand is SHARED by the three taps of a filter row.
//
// The generic contraction kernel (gemm_kernel.h, loaders ConvFwdA / ConvDgradA) walks K = (kh, kw, c) tap by tap: every K
// step DMAs a fresh BM x 32 A tile, although the tiles of (kh, kw-1), (kh, kw), (kh, kw+1) hold the same pixels shifted by
// one row of the tile -- the kernel's K loop is bound by operand delivery (L2 -> LDS, HISTORY.md), and two thirds of the A
// traffic of a 3x3 convolution is that repetition.  Here a K "super-step" is (kh, 32 channels): ONE A tile of BM pixels
// [m0 - 1, m0 + BM - 1) in the image's linear NHWC pixel order, shifted by (kh - 1) image rows, serves three MFMA steps that
// read their fragments at row offsets 0, 1, 2 -- the tile yields BM - 2 output pixels; the B tile (BN filters x 32
// channels of one tap) still changes every step.  Per three steps the DMA moves BM*64 + 3*BN*64 bytes instead of
// 3*(BM + BN)*64: 40 KB instead of 72 KB for 256 x 128.
//   * pixels are linear, so the neighbour of a pixel at x = 0 / x = W-1 in the tile is the end / start of the adjacent image
//     row: the A fragments of those rows are zeroed in registers for the tap that would reach across (one v_cndmask per
//     fragment register, taps kw = 0 and kw = 2 only);
//   * image rows above / below the image (kh = 0 at y = 0, kh = 2 at y = H-1) and pixels outside [0, M) are zero-filled by
//     the buffer load itself (voffset 0x80000000), one validity bit per (lane, kh) computed once;
//   * LDS: an A ring of 2 tiles (the next super-step's tile lands under the three steps of the current one) + a B ring of
//     3 tiles: 56 KB for 256 x 128 (72 KB in the generic kernel), counted s_waitcnt vmcnt as there;
//   * the epilogue is the generic one (tile_epilogue: bias / statistics / fused BatchNorm backward), with the block's row
//     limit lowered to m0 + BM - 2: the last two rows of a tile belong to the next one.
#pragma once
#include "gemm_kernel.h"

namespace vtxg {

struct Conv3x3Geo { int N, H, W, CH, M; };   // CH = channels of the A tensor (C forward, KO input gradient); M = N*H*W

template <int N> __device__ __forceinline__ void vtx_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
}

template <int BM, int BN, int WM, int WN, class BL, class EP>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 8) ? 4 : (BM * BN == 128 * 128 ? 3 : 4))
void conv3x3_shared_kernel(const bf16_t* __restrict__ x, Conv3x3Geo g, BL bl, EP ep, int flip, int tiles_n) {
    constexpr int NW = WM * WN, WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int BK = 32, ROWS_OUT = BM - 2;
    constexpr int ATILE = BM * BK, BTILE = BN * BK;            // elements
    constexpr int NA = BM / (16 * NW);                          // A wave-instructions (16 rows x 64 B each) per wave and tile
    typedef DmaStager<BN, NW, BL, BK> SB;
    constexpr int NB = SB::NI;
    static_assert(NA >= 1 && NA * 16 * NW == BM, "A tile: whole wave-instructions");
    constexpr int LDS_BYTES = (2 * ATILE + 3 * BTILE) * 2;
    HIP_DYNAMIC_SHARED(bf16_t, lds)
    bf16_t* abuf = lds;                       // [2][BM][32]
    bf16_t* bbuf = lds + 2 * ATILE;           // [3][BN][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-contiguous tile ranges, as in contraction_v2_kernel
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * ROWS_OUT, n0 = tile_n * BN;
    ep.M = ep.M < m0 + ROWS_OUT ? ep.M : m0 + ROWS_OUT;        // rows BM-2, BM-1 of this tile are the next tile's

    // ---- A stager: tile row r = pixel m0 - 1 + r; lane (row = 16 q + lane / 4, physical slot = lane % 4) fetches the logical
    // 16-byte channel chunk swz_slot(slot, row) of its row (source-side swizzle, as DmaStager)
    const int CH = g.CH, W = g.W, H = g.H;
    const long bias_el = (long)(W + 1) * CH;                    // descriptor base = x - bias: offsets stay non-negative
    __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(x) - bias_el, (short)0, (int)(((long)g.M * CH + 2 * bias_el) * 2), 0x00020000);
    uint32_t aoff[NA]; uint32_t avalid[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = 16 * (wave + NW * i) + (lane >> 2), phys = lane & 3;
        const int p = m0 - 1 + row;                             // linear pixel of this tile row
        const bool in = p >= 0 && p < g.M;
        const int pp = in ? p : 0;
        const int y = qdiv(pp, W) % H;                          // (n*H + y) = pp / W
        aoff[i] = (uint32_t)((((long)pp + 1) * CH + 8 * swz_slot(phys, row)) * 2);
        avalid[i] = in ? ((y >= 1 ? 1u : 0u) | 2u | (y + 1 < H ? 4u : 0u)) : 0u;
    }
    auto issue_a = [&](int kh, int c0, bf16_t* dst) {
        const uint32_t so = (uint32_t)(((long)kh * W * CH + c0) * 2);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const uint32_t vo = (avalid[i] >> kh) & 1u ? aoff[i] : VTX_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (__attribute__((address_space(3))) void*)(dst + (wave + NW * i) * 512), 16,
                                                     (int)vo, (int)so, 0, 0);
        }
    };
    SB sb;
    sb.init(bl, n0, wave, lane, 0);

    // ---- x-border masks of this lane's output rows (one per 16-row fragment): bit 0: x == 0, bit 1: x == W - 1
    uint32_t xb[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * WTM + i * 16 + (lane & 15);
        const int mm = m < g.M ? m : 0;
        const int xx = mm - qdiv(mm, W) * W;
        xb[i] = (xx == 0 ? 1u : 0u) | (xx == W - 1 ? 2u : 0u);
    }

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int PPT = EpiShape<BN, NW, EP>::PPT;
    float pre[PPT][4];
    epi_prefetch<BN, NW>(ep, pre, tid, n0);

    // ---- K loop.  step t = 3 u + kw, super-step u = kh * (CH / 32) + cc; all indices advance by counters (no divisions)
    const int ccs = CH >> 5, U = 3 * ccs, T = 3 * U;
    auto bk0 = [&](int kh, int cc, int kw) {                    // first k of a step in the filter's [tap][channel] order
        const int tap = flip ? (2 - kh) * 3 + (2 - kw) : kh * 3 + kw;
        return tap * CH + cc * 32;
    };
    int pkh = 0, pcc = 0, pkw = 0;                              // (kh, cc, kw) of the next B tile to request
    auto next_b = [&](bf16_t* dst) {
        sb.issue(bl, bk0(pkh, pcc, pkw), dst, wave);
        if (++pkw == 3) { pkw = 0; if (++pcc == ccs) { pcc = 0; ++pkh; } }
    };
    issue_a(0, 0, abuf);
    next_b(bbuf);
    if (T > 1) next_b(bbuf + BTILE);
    int akh = 0, acc_ = 0;                                      // (kh, cc) of the A tile requested last
    int u = 0, kw = 0, bs = 0;                                  // super-step, tap column, B stage of step t
    for (int t = 0; t < T; ++t) {
        // this wave's pieces of B_t (and of A_u when kw == 0) have landed; later requests may still be in flight
        if (t + 2 >= T) vtx_wait_vm<0>();
        else if (kw == 0) vtx_wait_vm<NB>();                    // younger: B_{t+1}
        else if (u + 1 < U) vtx_wait_vm<NB + NA>();             // younger: B_{t+1} and A_{u+1}
        else vtx_wait_vm<NB>();
        __builtin_amdgcn_s_barrier();
        if (t + 2 < T) next_b(bbuf + (bs + 2 >= 3 ? bs - 1 : bs + 2) * BTILE);
        if (kw == 0 && u + 1 < U) {
            if (++acc_ == ccs) { acc_ = 0; ++akh; }
            issue_a(akh, acc_ * 32, abuf + (((u + 1) & 1) ? ATILE : 0));
        }
        const bf16_t* ca = abuf + ((u & 1) ? ATILE : 0);
        const bf16_t* cb = bbuf + bs * BTILE;
        bf16x8_t fa[MT], fb[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = wm * WTM + i * 16 + (lane & 15) + kw;
            fa[i] = *reinterpret_cast<const bf16x8_t*>(ca + row * 32 + swz_slot(lane >> 4, row) * 8);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = SB::frag(cb, wn * WTN + j * 16, lane, 0);
        if (kw != 1) {                                          // wave-uniform: taps reaching across the image's left / right edge
            const uint32_t bit = kw == 0 ? 1u : 2u;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if (xb[i] & bit) fa[i] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        bs = bs + 1 >= 3 ? 0 : bs + 1;
        if (++kw == 3) { kw = 0; ++u; }
    }
    tile_epilogue<BM, BN, WM, WN, LDS_BYTES>(ep, acc, pre, lds, m0, n0, tile_m, tile_n, tid, lane, wave);
}

extern int g_vtx_sw_conv3x3_shared;     // vtx_set_switch("conv3x3_shared"): 0 off, 1 on (M >= 100 000 pixels), 2 every size (tests)

// profiler class name in the contraction kernel's "[BM = ..., AL = ..., EP = ...]" form (bench.py parses it)
template <class T, int FLIP> struct Conv3x3SharedA {};
template <int BM, int BN, int WM, int WN, class AL, class BL, class EP> inline const char* conv3x3_class_name() { return __PRETTY_FUNCTION__; }

// Launch when the problem is this kernel's: bf16, 3x3, stride 1, pad 1 (same-size output), CH % 32 == 0, buffer-addressable.
// a_ptr / CH: the tensor the taps slide over (x forward, dy input gradient); bl: the filter as a [rows = N][K = 9*CH] row-major
// operand ([KO][3][3][C] forward, [C][3][3][KO] input gradient); flip: the input gradient's tap order.
// Returns the number of statistics strips (tiles of BM - 2 rows) or 0 when the problem is not taken.
template <class EP, class FB>
inline int conv3x3_shared_try(const void* a_ptr, int Nimg, int H, int W, int CH, int Nout, FB make_b, const EP& ep_in, int flip,
                              hipStream_t st) {
    if (!g_vtx_sw_conv3x3_shared || CH % 32 != 0 || Nout % 8 != 0) return 0;
    const long Ml = (long)Nimg * H * W;
    // measured per layer at bs 256 (profiles/r03_conv3x3_shared.txt): 64->64 @56x56 118 -> 101 us forward, 114 -> 107 input gradient;
    // 128->128 @28x28 81 -> 79 / 79 -> 78; 256->256 @14x14 68 -> 75 and 512->512 @7x7 67 -> 96 (few tiles, short pipelines):
    // the kernel takes the large images only
    if (Ml >= VTX_PIXEL_LIMIT || Ml < (g_vtx_sw_conv3x3_shared >= 2 ? 1 : 100000) || ((double)Ml * CH + 2.0 * (W + 1) * CH) * 2 >= VTX_BUF_LIMIT) return 0;
    const int M = (int)Ml;
    Conv3x3Geo g{Nimg, H, W, CH, M};
    EP ep = ep_in;
    if constexpr (EP::STAGED) ep.nt = vtx_nt_policy((double)M * Nout * sizeof(typename EP::Out));
#define VTX_C3(BM_, BN_, WM_, WN_, SB_)                                                                                   \
    {                                                                                                                     \
        PlainKC<bf16_t, SB_> b; make_b(b);                                                                                \
        if (!b.buf_ok(32)) return 0;                                                                                      \
        g_vtx_last_generation = 2;                      /* an LDS-DMA kernel: what vtx_last_contraction_generation reports */ \
        g_vtx_generation_count[2].fetch_add(1, std::memory_order_relaxed);                                                \
        const int tiles_m = vtx_cdiv(M, BM_ - 2), tiles_n = vtx_cdiv(Nout, BN_);                                          \
        constexpr size_t lds_bytes = (2 * (size_t)BM_ * 32 + 3 * (size_t)BN_ * 32) * 2;                                  \
        auto kern = conv3x3_shared_kernel<BM_, BN_, WM_, WN_, PlainKC<bf16_t, SB_>, EP>;                                  \
        bool prof = g_vtx_prof_on != 0;                                                                                   \
        if (prof) {                                                                                                       \
            static const int cls0 = vtx_prof_register(conv3x3_class_name<BM_, BN_, WM_, WN_, Conv3x3SharedA<bf16_t, 0>, PlainKC<bf16_t, SB_>, EP>()); \
            static const int cls1 = vtx_prof_register(conv3x3_class_name<BM_, BN_, WM_, WN_, Conv3x3SharedA<bf16_t, 1>, PlainKC<bf16_t, SB_>, EP>()); \
            const int cls = flip ? cls1 : cls0;                                                                           \
            prof = g_vtx_prof_only < 0 || g_vtx_prof_only == cls;                                                         \
            if (prof) {                                                                                                   \
                hipEvent_t e0, e1;                                                                                        \
                vtx_prof_events(cls, 2.0 * M * Nout * 9.0 * CH, 2.0 * ((double)M * CH + 9.0 * CH * Nout) + epi_bytes(ep, (double)M * Nout, 1), &e0, &e1); \
                hipExtLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(64 * WM_ * WN_), (uint32_t)lds_bytes, st, e0, e1, 0, (const bf16_t*)a_ptr, g, b, ep, flip, tiles_n); \
                return tiles_m;                                                                                           \
            }                                                                                                             \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(64 * WM_ * WN_), lds_bytes, st, (const bf16_t*)a_ptr, g, b, ep, flip, tiles_n); \
        return tiles_m;                                                                                                   \
    }
    const int c = Nout <= 64 ? 3 : pick_tile(M, Nout, 1, true);          // the contraction kernel's cost model chooses the tile
    // 64 filters: four waves on 254-pixel tiles (half the per-tile prologue / epilogue per output of the 126-pixel tile:
    // 64->64 @56x56 forward 99.7 -> 97.3 us, input gradient 108.7 -> 98.8); small problems keep the 126-pixel tile
    if (Nout <= 64 && M >= (g_vtx_sw_conv3x3_shared >= 2 ? 512 : 100000)) VTX_C3(256, 64, 4, 1, 1)
    else if (Nout <= 64 || c == 3 || c == 5) VTX_C3(128, 64, 2, 2, 1)
    else if (c == 2 || c == 4) VTX_C3(128, 128, 2, 2, 2)
    else VTX_C3(256, 128, 4, 2, 1)
#undef VTX_C3
}

}  // namespace vtxg
