// Generation 3 for k-major operand pairs -- the weight gradients  C[m][n] (+)= sum_k A[k][m] B[k][n]  (A = the output gradient
// [pixels][KO], B = the layer input [pixels][C] or its im2col-free gather) -- included by gemm_kernel.h behind gemm_v3.h.
// Same phase schedules, barriers, counted waits and unit bookkeeping as the row-major kernels of gemm_v3.h (read that header
// first); what differs is the LDS image and the fragment path:
//   * a unit is [64 k][128 rows] exactly as in HBM (k-major: 64 rows of 256 bytes), staged by the same two 1-KiB
//     wave-instructions per wave (an instruction = 4 k rows); the 16-byte row chunks of a k row are XOR-permuted on the DMA
//     SOURCE side (swz_mc<128>: by k & 3 and bit 3 of k) so that
//   * fragments are 2 x ds_read_b64_tr_b16 each (the transposing read; vtx_ds_read_tr16_imm: asm, 16-bit immediate offsets from
//     ONE address register per 16-row tile and LDS buffer), bank-conflict free; the reads are invisible to hipcc's waitcnt
//     pass, so the phase's `s_waitcnt lgkmcnt(0)` behind its first barrier is what orders them in front of the MFMAs.
// A phase of the 256x256 kernel issues 24 / 8 / 16 / 0 transposing reads (twice the instructions of the row-major kernel for
// the same bytes); "the four B reads retired in front of the barrier" of the row-major schedule are the first 8 of 24 here:
// lgkmcnt(15) (the counter's ceiling) retires the first 9.
// Split-K slices write fp32 partial tiles (EpiStore<float>: the interior-tile lean path of gemm_v3.h); the caller reduces.
#pragma once

template <int WT, int QR, class L> struct UnitStagerMC {
    static_assert(L::MC, "k-major operands");
    static_assert(WT % QR == 0 && 128 % QR == 0 && QR % 8 == 0, "sub-blocks tile the wave tile and the unit in whole 16-byte chunks");
    static constexpr int NU = WT / QR;
    typename L::BState st;                   // slot 2u + j: wave-instruction j of unit u
    int kl[2];                               // the k row (inside a K tile) this lane stages with instruction j
    __amdgpu_buffer_rsrc_t rsrc;

    __device__ __forceinline__ void init(const L& l, int row0, int wave, int lane, int k_first) {
        const BufView v = l.view();
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(v.base), (short)0, (int)v.bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) kl[j] = 4 * (wave + 8 * j) + (lane >> 4);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = 8 * swz_mc<128>(lane & 15, kl[j]);         // unit row of the chunk this lane's LDS slot must hold
                const int orow = (r / QR) * WT + u * QR + (r % QR);
                l.template binit<64>(st, 2 * u + j, row0 + orow, kl[j], k_first);
            }
    }
    // (position-tracking loaders advance their state in voff: every slot is issued exactly once per K tile, in increasing k)
    template <int U, int J> __device__ __forceinline__ void issue1(const L& l, int k0, bool valid, bf16_t* unit, int wave) {
        const uint32_t so = valid ? l.template soff<64>(k0) : 0u;
        uint32_t vo;
        if constexpr (L::TAILS) vo = l.template voff<false, 64>(st, 2 * U + J, k0, kl[J]);
        else vo = l.template voff<true, 64>(st, 2 * U + J, k0, kl[J]);
        if (!valid) vo = VTX_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(unit + (wave + 8 * J) * 512),
                                                 16, (int)vo, (int)so, 0, 0);
    }
    template <int U> __device__ __forceinline__ void issue(const L& l, int k0, bool valid, bf16_t* unit, int wave) {
        issue1<U, 0>(l, k0, valid, unit, wave);
        issue1<U, 1>(l, k0, valid, unit, wave);
    }
};

// byte offset, inside a unit, of the first transposing read of this lane's fragment of unit rows r0 .. r0+15 (k half 0); the
// second read of the fragment lies 4 k rows (1024 bytes) further, the second k half 32 k rows (8192 bytes)
__device__ __forceinline__ uint32_t v3mc_lane_off(int r0, int lane) {
    const int w = lane & 15, ka = 8 * (lane >> 4) + (w >> 2), rr = r0 + 4 * (w & 3);
    return (uint32_t)((ka * 128 + swz_mc<128>(rr >> 3, ka) * 8 + (rr & 7)) * 2);
}

__device__ __forceinline__ bf16x8_t v3mc_join(vtx_v4s_t a, vtx_v4s_t b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

constexpr int V3_UB = V3_UNIT * 2;            // bytes of a unit

// ------------------------------------------------------------------ 256 x 256 (see contraction_v3_256x256_kernel)
template <class AL, class BL, class EP>
__global__ __launch_bounds__(512, 2) void contraction_v3mc_256x256_kernel(AL al, BL bl, EP ep, int K, int tiles_n, int kt_per_split,
                                                                          int abl, unsigned long long* dbg, int /*vb_tiles: unused*/) {
    constexpr int BM = 256, BN = 256, WM = 2, WN = 4, MT = 8, NT = 4;
    constexpr int BUF = 4 * V3_UNIT;
    typedef UnitStagerMC<128, 64, AL> SA;
    typedef UnitStagerMC<64, 32, BL> SB;
    HIP_DYNAMIC_SHARED(bf16_t, lds)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    V3_STAMP(0);
    int tile, slice;
    v3_block_tile(abl, tile, slice);
    set_slice(ep, slice);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int nkt = (K + 63) >> 6;
    const int kt0 = slice * kt_per_split;
    const int kt1 = kt0 + kt_per_split < nkt ? kt0 + kt_per_split : nkt;

    SA sa; SB sb;
    sa.init(al, m0, wave, lane, kt0 * 64);
    sb.init(bl, n0, wave, lane, kt0 * 64);
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int PPT = EpiShape<BN, 8, EP>::PPT;
    float pre[PPT][4];
    epi_prefetch<BN, 8>(ep, pre, tid, n0);

    bf16_t* const E = lds;
    bf16_t* const O = lds + BUF;
    constexpr int A0 = 0, A1 = V3_UNIT, B0 = 2 * V3_UNIT, B1 = 3 * V3_UNIT;       // element offsets of the units in a buffer
    const char* const ldsb = reinterpret_cast<const char*>(lds);
    // one address register per 16-row tile and buffer; unit, k half and second read are immediates (< 64 KiB)
    uint32_t aE[4], aO[4], bE[2], bO[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { aE[i] = v3mc_lane_off(wm * 64 + i * 16, lane); aO[i] = aE[i] + BUF * 2; }
#pragma unroll
    for (int j = 0; j < 2; ++j) { bE[j] = v3mc_lane_off(wn * 32 + j * 16, lane); bO[j] = bE[j] + BUF * 2; }
    vtx_v4s_t ra[4][2][2], rb[2][2][2][2];     // [tile][k half][read]; rb: [sub-block][tile][k half][read]

#define V3M_READ_A(ADR, U)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
        ra[i][0][0] = vtx_ds_read_tr16_imm<(U) * 2>(ldsb, ADR[i]);                                    \
        ra[i][0][1] = vtx_ds_read_tr16_imm<(U) * 2 + 1024>(ldsb, ADR[i]);                             \
        ra[i][1][0] = vtx_ds_read_tr16_imm<(U) * 2 + 8192>(ldsb, ADR[i]);                             \
        ra[i][1][1] = vtx_ds_read_tr16_imm<(U) * 2 + 8192 + 1024>(ldsb, ADR[i]);                      \
    }
#define V3M_READ_B(ADR, U, Q)                                                                       \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                  \
        rb[Q][j][0][0] = vtx_ds_read_tr16_imm<(U) * 2>(ldsb, ADR[j]);                                 \
        rb[Q][j][0][1] = vtx_ds_read_tr16_imm<(U) * 2 + 1024>(ldsb, ADR[j]);                          \
        rb[Q][j][1][0] = vtx_ds_read_tr16_imm<(U) * 2 + 8192>(ldsb, ADR[j]);                          \
        rb[Q][j][1][1] = vtx_ds_read_tr16_imm<(U) * 2 + 8192 + 1024>(ldsb, ADR[j]);                   \
    }
#define V3M_MMA(UA, UB)                                                                                               \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                acc[(UA) * 4 + i][(UB) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                            \
                    v3mc_join(rb[UB][j][h][0], rb[UB][j][h][1]), v3mc_join(ra[i][h][0], ra[i][h][1]), acc[(UA) * 4 + i][(UB) * 2 + j], 0, 0, 0);
// the fragment reads are asm statements: ALWAYS waited for by hand behind the phase's first barrier
#define V3M_COMPUTE_BEGIN()               \
    VTX3_FENCE();                         \
    __builtin_amdgcn_s_barrier();         \
    VTX3_WAIT_LGKM(0);                    \
    VTX3_FENCE();                         \
    V3_PRIO(1);

    if (kt0 < kt1) {
        sb.template issue<0>(bl, kt0 * 64, true, E + B0, wave);
        sa.template issue<0>(al, kt0 * 64, true, E + A0, wave);
        sb.template issue<1>(bl, kt0 * 64, true, E + B1, wave);
        sa.template issue<1>(al, kt0 * 64, true, E + A1, wave);
        {
            const bool v1 = kt0 + 1 < kt1;
            sb.template issue<0>(bl, (kt0 + 1) * 64, v1, O + B0, wave);
            sa.template issue<0>(al, (kt0 + 1) * 64, v1, O + A0, wave);
            sb.template issue<1>(bl, (kt0 + 1) * 64, v1, O + B1, wave);
        }
        VTX3_WAIT_VM(6);
        __builtin_amdgcn_s_barrier();
        V3_STAGGER(wave >= 4);
        V3_STAMP(1);
        for (int kt = kt0; kt < kt1; kt += 2) {
            const bool v1 = kt + 1 < kt1, v2 = kt + 2 < kt1, v3 = kt + 3 < kt1;
            const int k1 = (kt + 1) * 64, k2 = (kt + 2) * 64, k3 = (kt + 3) * 64;
            // ---- phase 1
            V3M_READ_B(bE, B0, 0)
            VTX3_FENCE();
            V3M_READ_A(aE, A0)
            sa.template issue<1>(al, k1, v1, O + A1, wave);
            VTX3_FENCE();
            VTX3_WAIT_LGKM(15);
            V3M_COMPUTE_BEGIN()
            V3M_MMA(0, 0)
            V3_COMPUTE_END()
            // ---- phase 2
            V3M_READ_B(bE, B1, 1)
            sb.template issue<0>(bl, k2, v2, E + B0, wave);
            V3M_COMPUTE_BEGIN()
            V3M_MMA(0, 1)
            V3_COMPUTE_END()
            // ---- phase 3
            V3M_READ_A(aE, A1)
            sa.template issue<0>(al, k2, v2, E + A0, wave);
            V3M_COMPUTE_BEGIN()
            V3M_MMA(1, 1)
            V3_COMPUTE_END()
            // ---- phase 4
            sb.template issue<1>(bl, k2, v2, E + B1, wave);
            VTX3_FENCE();
            VTX3_WAIT_VM(6);
            V3M_COMPUTE_BEGIN()
            V3M_MMA(1, 0)
            V3_COMPUTE_END()
            // ---- phase 5
            V3M_READ_B(bO, B0, 0)
            VTX3_FENCE();
            V3M_READ_A(aO, A0)
            sa.template issue<1>(al, k2, v2, E + A1, wave);
            VTX3_FENCE();
            VTX3_WAIT_LGKM(15);
            V3M_COMPUTE_BEGIN()
            V3M_MMA(0, 0)
            V3_COMPUTE_END()
            // ---- phase 6
            V3M_READ_B(bO, B1, 1)
            sb.template issue<0>(bl, k3, v3, O + B0, wave);
            V3M_COMPUTE_BEGIN()
            V3M_MMA(0, 1)
            V3_COMPUTE_END()
            // ---- phase 7
            V3M_READ_A(aO, A1)
            sa.template issue<0>(al, k3, v3, O + A0, wave);
            V3M_COMPUTE_BEGIN()
            V3M_MMA(1, 1)
            V3_COMPUTE_END()
            // ---- phase 8
            sb.template issue<1>(bl, k3, v3, O + B1, wave);
            VTX3_FENCE();
            VTX3_WAIT_VM(6);
            V3M_COMPUTE_BEGIN()
            V3M_MMA(1, 0)
            V3_COMPUTE_END()
        }
        VTX3_WAIT_VM(0);
        V3_STAGGER(wave < 4);
        V3_STAMP(2);
    }
#undef V3M_READ_A
#undef V3M_READ_B
#undef V3M_MMA
    if (!v3_lean_epilogue<BM, BN, WM, WN>(ep, acc, lds, m0, n0, lane, wave, (abl & 256) == 0))
        tile_epilogue<BM, BN, WM, WN, 2 * BUF * 2>(ep, acc, pre, lds, m0, n0, tile / tiles_n, tile % tiles_n, tid, lane, wave);
    V3_STAMP(3);
}

// ------------------------------------------------------------------ 256 x 128 (see contraction_v3_256x128_kernel)
template <class AL, class BL, class EP>
__global__ __launch_bounds__(512, 2) void contraction_v3mc_256x128_kernel(AL al, BL bl, EP ep, int K, int tiles_n, int kt_per_split,
                                                                          int abl, unsigned long long* dbg, int /*vb_tiles: unused*/) {
    constexpr int BM = 256, BN = 128, WM = 4, WN = 2, MT = 4, NT = 4;
    constexpr int BUF = 3 * V3_UNIT;
    typedef UnitStagerMC<64, 32, AL> SA;
    typedef UnitStagerMC<64, 64, BL> SB;
    HIP_DYNAMIC_SHARED(bf16_t, lds)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    V3_STAMP(0);
    int tile, slice;
    v3_block_tile(abl, tile, slice);
    set_slice(ep, slice);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int nkt = (K + 63) >> 6;
    const int kt0 = slice * kt_per_split;
    const int kt1 = kt0 + kt_per_split < nkt ? kt0 + kt_per_split : nkt;

    SA sa; SB sb;
    sa.init(al, m0, wave, lane, kt0 * 64);
    sb.init(bl, n0, wave, lane, kt0 * 64);
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int PPT = EpiShape<BN, 8, EP>::PPT;
    float pre[PPT][4];
    epi_prefetch<BN, 8>(ep, pre, tid, n0);

    constexpr int A0 = 0, A1 = V3_UNIT, B = 2 * V3_UNIT;
    const char* const ldsb = reinterpret_cast<const char*>(lds);
    uint32_t aX[3][2], bX[3][4];               // [buffer][16-row tile]
#pragma unroll
    for (int x = 0; x < 3; ++x) {
#pragma unroll
        for (int i = 0; i < 2; ++i) aX[x][i] = v3mc_lane_off(wm * 32 + i * 16, lane) + x * BUF * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) bX[x][j] = v3mc_lane_off(wn * 64 + j * 16, lane) + x * BUF * 2;
    }
    vtx_v4s_t ra[2][2][2], rb[4][2][2];

#define V3M_READ_A(X, U)                                                                            \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
        ra[i][0][0] = vtx_ds_read_tr16_imm<(U) * 2>(ldsb, aX[X][i]);                                  \
        ra[i][0][1] = vtx_ds_read_tr16_imm<(U) * 2 + 1024>(ldsb, aX[X][i]);                           \
        ra[i][1][0] = vtx_ds_read_tr16_imm<(U) * 2 + 8192>(ldsb, aX[X][i]);                           \
        ra[i][1][1] = vtx_ds_read_tr16_imm<(U) * 2 + 8192 + 1024>(ldsb, aX[X][i]);                    \
    }
#define V3M_READ_B(X)                                                                               \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                  \
        rb[j][0][0] = vtx_ds_read_tr16_imm<B * 2>(ldsb, bX[X][j]);                                    \
        rb[j][0][1] = vtx_ds_read_tr16_imm<B * 2 + 1024>(ldsb, bX[X][j]);                             \
        rb[j][1][0] = vtx_ds_read_tr16_imm<B * 2 + 8192>(ldsb, bX[X][j]);                             \
        rb[j][1][1] = vtx_ds_read_tr16_imm<B * 2 + 8192 + 1024>(ldsb, bX[X][j]);                      \
    }
#define V3M_MMA(UA)                                                                                                   \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                 \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
                acc[(UA) * 2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                       \
                    v3mc_join(rb[j][h][0], rb[j][h][1]), v3mc_join(ra[i][h][0], ra[i][h][1]), acc[(UA) * 2 + i][j], 0, 0, 0);
// one K tile: X = index of its buffer, Y = pointer of the buffer tile +2 is staged into, V = tile +2 exists
#define V3M_TILE(X, Y, K2, V)                                                  \
    V3M_READ_B(X)                                                              \
    VTX3_FENCE();                                                              \
    V3M_READ_A(X, A0)                                                          \
    sb.template issue<0>(bl, K2, V, (Y) + B, wave);                            \
    sa.template issue1<0, 0>(al, K2, V, (Y) + A0, wave);                       \
    V3M_COMPUTE_BEGIN()                                                        \
    V3M_MMA(0)                                                                 \
    V3_COMPUTE_END()                                                           \
    V3M_READ_A(X, A1)                                                          \
    sa.template issue1<0, 1>(al, K2, V, (Y) + A0, wave);                       \
    sa.template issue<1>(al, K2, V, (Y) + A1, wave);                           \
    VTX3_FENCE();                                                              \
    VTX3_WAIT_VM(6);                                                           \
    V3M_COMPUTE_BEGIN()                                                        \
    V3M_MMA(1)                                                                 \
    V3_COMPUTE_END()

    if (kt0 < kt1) {
        bf16_t* const X0 = lds;
        bf16_t* const X1 = lds + BUF;
        bf16_t* const X2 = lds + 2 * BUF;
        sb.template issue<0>(bl, kt0 * 64, true, X0 + B, wave);
        sa.template issue<0>(al, kt0 * 64, true, X0 + A0, wave);
        sa.template issue<1>(al, kt0 * 64, true, X0 + A1, wave);
        {
            const bool v1 = kt0 + 1 < kt1;
            sb.template issue<0>(bl, (kt0 + 1) * 64, v1, X1 + B, wave);
            sa.template issue<0>(al, (kt0 + 1) * 64, v1, X1 + A0, wave);
            sa.template issue<1>(al, (kt0 + 1) * 64, v1, X1 + A1, wave);
        }
        VTX3_WAIT_VM(6);
        __builtin_amdgcn_s_barrier();
        V3_STAGGER(wave >= 4);
        V3_STAMP(1);
        for (int kt = kt0; kt < kt1; kt += 3) {
            V3M_TILE(0, X2, (kt + 2) * 64, kt + 2 < kt1)
            V3M_TILE(1, X0, (kt + 3) * 64, kt + 3 < kt1)
            V3M_TILE(2, X1, (kt + 4) * 64, kt + 4 < kt1)
        }
        VTX3_WAIT_VM(0);
        V3_STAGGER(wave < 4);
        V3_STAMP(2);
    }
#undef V3M_READ_A
#undef V3M_READ_B
#undef V3M_MMA
#undef V3M_TILE
#undef V3M_COMPUTE_BEGIN
    if (!v3_lean_epilogue<BM, BN, WM, WN>(ep, acc, lds, m0, n0, lane, wave, (abl & 256) == 0))
        tile_epilogue<BM, BN, WM, WN, 3 * BUF * 2>(ep, acc, pre, lds, m0, n0, tile / tiles_n, tile % tiles_n, tid, lane, wave);
    V3_STAMP(3);
}
