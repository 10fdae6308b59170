// Generation 3 of the contraction kernel (included by gemm_kernel.h, namespace vtxg): a deep-pipelined K loop for the
// MFMA-leaning classes -- text-head GEMMs, 3x3 / strided convolutions and 1x1 convolutions of the 14x14 / 7x7 stages,
// forward and input gradient (row-major "KC" operand pairs).  Same loaders, same epilogues, same C = epi(A B^T) contract
// as generation 2; what changes is the shape of the K loop:
//
//   * 8 waves, ONE block per CU, 64-deep K tiles staged by LDS-DMA in UNITS of 128 rows x 64 k (16 KiB = two 1-KiB
//     wave-instructions per wave).  A unit is not a contiguous row range of the tile: it holds, for every wave along the
//     operand's axis, the rows of ONE sub-block of that wave's tile (unit row r <-> operand row (r / QR) * WT + u * QR +
//     r % QR), so that a unit is consumed -- by all of its readers -- in exactly one phase and can be re-staged right after.
//   * a K tile is worked off in PHASES of 16 MFMAs per wave; every phase is
//         fragment reads of the phase (ds_read_b128) | stage one or two units (LDS-DMA) | [counted waits] | s_barrier |
//         lgkmcnt(0) | 16 MFMAs at raised priority | s_barrier
//     and the two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run staggered by ONE barrier, so that on every
//     SIMD one wave is in its MFMA cluster while the other issues its reads and DMAs.
//   * the DMA queue is never drained inside the loop: `s_waitcnt vmcnt(6)` once per K tile leaves three units (one and a
//     half K-tile halves) in flight across the barriers.
//
// Hazard rules the schedules below are built on (cdna_hip_programming.md section 5, "The 256^2 8-phase template"):
//   RAW  a staged unit is read one phase AFTER the phase whose counted vmcnt retires it (the wait sits in front of that
//        phase's first barrier; with the groups staggered by a barrier, a reader has then passed a barrier behind EVERY
//        wave's wait);
//   WAR  a unit is re-staged two phases after the phase that issued its last reads -- or one phase after, when an lgkmcnt
//        in front of the reading phase's first barrier retired those reads (the four B reads of a 12-read phase, issued
//        first: lgkmcnt(8)).
// The CPU emulator checks RAW in its late-landing DMA mode (a DMA lands at the issuing lane's covering vmcnt) and the
// issue-order part of WAR in its default mode (a DMA lands at issue): tests/test_kernels.py runs the v3 cases in both.
#pragma once

#ifdef HIPEMU
#define VTX3_WAIT_VM(N) hipemu::dma_retire(N)
#define VTX3_WAIT_LGKM(N) ((void)0)
#define VTX3_FENCE() ((void)0)
#else
#define VTX3_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define VTX3_WAIT_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory")
#define VTX3_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// measurement builds (-DVTX_ABLATE, tools/ablate_gen3.py): bit 0 no MFMA, bit 1 no fragment reads, bit 2 no DMA inside the loop,
// bit 6 (64) no K loop at all, bit 7 (128) no epilogue
#ifdef VTX_ABLATE
#define V3_ABL(bits) (abl & (bits))
// time stamps of wave `wave` of block blockIdx.x (shader clock): dbg[(block * 8 + wave) * 4 + slot], slots: 0 kernel entry,
// 1 first tile landed (K loop starts), 2 K loop done, 3 epilogue done (vtx_set_debug_buffer)
#define V3_STAMP(slot)                                                                                          \
    do {                                                                                                        \
        if (dbg && lane == 0) dbg[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 4 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define V3_ABL(bits) false
#define V3_STAMP(slot) ((void)0)
#endif

constexpr int V3_UNIT = 128 * 64;          // elements of a staged unit (128 rows x 64 k, 16 KiB)

// Stages the units of one operand.  WT = rows of the operand one wave's tile spans, QR = rows of one sub-block.
template <int WT, int QR, class L> struct UnitStager {
    static_assert(!L::MC, "generation 3 takes row-major (k-contiguous) operands");
    static_assert(WT % QR == 0 && 128 % QR == 0 && QR % 16 == 0, "sub-blocks tile the wave tile and the unit");
    static constexpr int NU = WT / QR;       // units per K tile
    typename L::BState st;                   // slot 2u + j: wave-instruction j of unit u
    __amdgpu_buffer_rsrc_t rsrc;

    __device__ __forceinline__ void init(const L& l, int row0, int wave, int lane) {
        const BufView v = l.view();
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(v.base), (short)0, (int)v.bytes, 0x00020000);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = 8 * (wave + 8 * j) + (lane >> 3);          // unit row this lane stages (128-byte rows: 8 lanes each)
                const int orow = (r / QR) * WT + u * QR + (r % QR);      // operand row inside the block tile
                l.template binit<64>(st, 2 * u + j, row0 + orow, 8 * swz_slot64(lane & 7, r));
            }
    }
    // wave-instruction J of unit U of the K tile starting at k0; valid == false (wave-uniform: the tile lies beyond this
    // block's K range) delivers zeros -- the DMA count per phase stays what the vmcnt immediates assume
    template <int U, int J> __device__ __forceinline__ void issue1(const L& l, int k0, bool valid, bf16_t* unit, int wave) {
        const uint32_t so = valid ? l.template soff<64>(k0) : 0u;
        uint32_t vo;
        if constexpr (L::TAILS) vo = l.template voff<false, 64>(st, 2 * U + J, k0);
        else vo = l.template voff<true, 64>(st, 2 * U + J, k0);
        if (!valid) vo = VTX_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(unit + (wave + 8 * J) * 512),
                                                 16, (int)vo, (int)so, 0, 0);
    }
    template <int U> __device__ __forceinline__ void issue(const L& l, int k0, bool valid, bf16_t* unit, int wave) {
        issue1<U, 0>(l, k0, valid, unit, wave);
        issue1<U, 1>(l, k0, valid, unit, wave);
    }
};

__device__ __forceinline__ bf16x8_t v3_ld(const bf16_t* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

// element offset, inside a unit, of this lane's fragment of unit rows r0 .. r0+15 (r0 % 16 == 0) for the k half h
__device__ __forceinline__ int v3_lane_off(int r0, int lane, int h) {
    return (r0 + (lane & 15)) * 64 + (((4 * h + (lane >> 4)) ^ ((lane >> 1) & 7)) * 8);
}

// XCD-aware block -> (tile, slice) map of the contraction kernels (see contraction_v2_kernel)
__device__ __forceinline__ void v3_block_tile(int abl, int& tile, int& slice) {
    if (abl & 32) { tile = blockIdx.x; slice = blockIdx.y; return; }
    if (gridDim.y == 1 || (abl & 16)) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        slice = blockIdx.y;
    } else {
        const int T = gridDim.x, nwg = T * gridDim.y, L = blockIdx.y * T + blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = L & 7, idx = L >> 3;
        const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        slice = w / T; tile = w - slice * T;
    }
}

// the same map for the L-th VIRTUAL block of a persistent launch (T tiles, no split-K): physical block p works off the virtual
// blocks p, p + G, p + 2G, ... with G a multiple of 8, so every virtual block of a physical block lies in the tile range of ITS XCD
__device__ __forceinline__ int v3_vb_tile(int abl, int L, int T) {
    if (abl & 32) return L;
    const int q = T >> 3, r = T & 7, xcd = L & 7, idx = L >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// A/B builds (python -m virtex_amd.build --variant X --define ...): VTX3_NO_LGKM0 leaves the fragment waits to hipcc's own
// lgkmcnt ladder inside the MFMA cluster, VTX3_NO_PRIO drops the priority flips, VTX3_NO_STAGGER runs the two wave groups in
// lockstep (what the stagger is worth)
#ifdef VTX3_NO_LGKM0
#define V3_WAIT_FRAGS() ((void)0)
#else
#define V3_WAIT_FRAGS() VTX3_WAIT_LGKM(0)
#endif
#ifdef VTX3_NO_PRIO
#define V3_PRIO(x) ((void)0)
#else
#define V3_PRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
#ifdef VTX3_NO_STAGGER
#define V3_STAGGER(cond) ((void)0)
#else
#define V3_STAGGER(cond) do { if (cond) __builtin_amdgcn_s_barrier(); } while (0)
#endif
// the synchronised half of a phase: everything the phase issued is behind the first barrier; MFMAs between the barriers
#define V3_COMPUTE_BEGIN()                \
    VTX3_FENCE();                         \
    __builtin_amdgcn_s_barrier();         \
    V3_WAIT_FRAGS();                      \
    VTX3_FENCE();                         \
    V3_PRIO(1);
#define V3_COMPUTE_END()                  \
    V3_PRIO(0);                           \
    VTX3_FENCE();                         \
    __builtin_amdgcn_s_barrier();         \
    VTX3_FENCE();


// ------------------------------------------------------------------ the lean epilogue of interior tiles
// With ONE block per CU nothing overlaps a block's epilogue, and the general tile_epilogue (run-time activation / dropout /
// pre-activation / row-map dispatch and bounds checks around every 16x16 tile, fully unrolled over the 8 row steps of a
// 128-row wave tile) measured 37 000 cycles on the 256x256 kernel -- 13 K tiles' worth (profiles/r04_gen3_ablation.txt).
// Blocks whose tile lies inside the matrix and whose epilogue is {alpha, bias, none / GELU / ReLU, residual, store} -- every
// text-head GEMM of the step -- take this path instead: ACT is a compile-time constant, no bounds logic, two wave-private
// strips used alternately (the LDS round trip of step i+1 overlaps the stores of step i), 16-byte stores.  Same arithmetic,
// same rounding points as EpiStore::transform / finish.
template <class EP> struct V3Lean { static constexpr bool OK = false; };
template <class T> struct V3Lean<EpiStore<T, STATS_NONE>> { static constexpr bool OK = true; };

// PD: the pre-activation copy and the dropout of the feed-forward layer's first GEMM (transdec: gelu -> dropout, with the
// pre-activation kept for the backward pass) -- the pre-activation values take a second strip of the same step.
template <int ACT, int WTM, int WTN, class T, bool PD = false>
__device__ __forceinline__ void v3_lean_store(const EpiStore<T, STATS_NONE>& ep, f32x4_t (&acc)[WTM / 16][WTN / 16], bf16_t* lds,
                                              int mw, int nw, int lane, int wave) {
    constexpr int MT = WTM / 16, NT = WTN / 16;
    constexpr int ROWB = WTN * (int)sizeof(T) + 16;            // padded strip row (bytes)
    constexpr int CPR = WTN * (int)sizeof(T) / 16;             // 16-byte chunks per row
    constexpr int EPV = 16 / (int)sizeof(T);
    constexpr int NCH = 16 * CPR / 64;                         // chunks a lane moves per 16-row step
    static_assert(16 * CPR % 64 == 0, "whole wave-instructions per step");
    char* const strip0 = reinterpret_cast<char*>(lds) + wave * (2 * 16 * ROWB);
    float4 bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = ep.bias ? *reinterpret_cast<const float4*>(ep.bias + nw + j * 16 + 4 * (lane >> 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float alpha = ep.alpha;
    const bool has_res = ep.residual != nullptr;
    // ACT_SOFTMAX_GRAD (the tied projection recomputed in the cross-entropy backward): target, scale and log-sum-exp of the lane's
    // row in every 16-row step, all requested here -- the general epilogue this mode used to take costs 37 000 cycles per tile
    typename EpiStore<T, STATS_NONE>::RowData rd[ACT == ACT_SOFTMAX_GRAD ? MT : 1];
    if constexpr (ACT == ACT_SOFTMAX_GRAD) {
#pragma unroll
        for (int i = 0; i < MT; ++i) rd[i] = ep.row_data(mw + i * 16 + (lane & 15));
        vtx_loads_issued();
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        char* const strip = strip0 + (PD ? 1 : (i & 1)) * (16 * ROWB);
        char* const pstrip = strip0;                           // PD: pre-activation values of the step
        uint4 res[NCH];
        if (has_res) {                                         // requested before the step's trip through the strip
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int c = lane + 64 * q;
                res[q] = *reinterpret_cast<const uint4*>(ep.residual + (long)(mw + i * 16 + c / CPR) * ep.ldr + nw + (c % CPR) * EPV);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4_t v = acc[i][j] * alpha;
            v[0] += bv[j].x; v[1] += bv[j].y; v[2] += bv[j].z; v[3] += bv[j].w;
            if constexpr (PD) {
                if (ep.preact) st4v<T>(reinterpret_cast<T*>(pstrip + (lane & 15) * ROWB) + j * 16 + 4 * (lane >> 4), v);
            }
            if constexpr (ACT == ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            } else if constexpr (ACT == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if constexpr (ACT == ACT_SOFTMAX_GRAD) {
                const int n = nw + j * 16 + 4 * (lane >> 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rd[i].g * (__expf(v[e] - rd[i].l) - ((long long)(n + e) == rd[i].t ? 1.f : 0.f));
            }
            if constexpr (PD) {
                if (ep.drop.thresh) {
                    const long o = (long)(mw + i * 16 + (lane & 15)) * ep.ldc + nw + j * 16 + 4 * (lane >> 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ep.drop.apply(v[e], (uint64_t)(o + e));
                }
            }
            st4v<T>(reinterpret_cast<T*>(strip + (lane & 15) * ROWB) + j * 16 + 4 * (lane >> 4), v);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int c = lane + 64 * q, r = c / CPR, ch = c % CPR;
            uint4 w = *reinterpret_cast<const uint4*>(strip + r * ROWB + ch * 16);
            if constexpr (PD) {
                if (ep.preact)
                    *reinterpret_cast<uint4*>(ep.preact + (long)(mw + i * 16 + r) * ep.ldc + nw + ch * EPV) =
                        *reinterpret_cast<const uint4*>(pstrip + r * ROWB + ch * 16);
            }
            if (has_res) w = add16<T>(w, res[q]);
            T* dst = ep.out + (long)(mw + i * 16 + r) * ep.ldc + nw + ch * EPV + ep.split_off;
            if (ep.nt) st16_nt(dst, u32x4_t{w.x, w.y, w.z, w.w});
            else *reinterpret_cast<uint4*>(dst) = w;
        }
        if constexpr (PD) __builtin_amdgcn_wave_barrier();     // both strips are re-written by the next step
        // (the other strip is written next: this one is re-written two steps from now, behind the next wave barrier)
    }
}

// The same epilogue without LDS (round 4, second half): the four 16x16 tiles of a wave row are TRANSPOSED ACROSS THE FOUR LANE ROWS
// of the wave in registers -- gfx950's v_permlane16_swap / v_permlane32_swap, two stages -- so that lane (r, g) ends up with all
// 16 columns of tile g for matrix row r: 32 contiguous bytes of bf16 (64 of fp32) per lane, 128 (256) per matrix row and wave,
// stored straight from registers.  Per 16-row step: 4 W swaps (W = dwords per 4-column group) and 2 (4) 16-byte stores instead of
// 4 strip writes, a wave barrier, 2 (4) strip reads and their LDS latency.  Element (j, g) = tile j, column group g:
//   stage 1  permlane16_swap(T0, T1) -> rows {E00 E10 E02 E12}, {E01 E11 E03 E13};  (T2, T3) alike
//   stage 2  permlane32_swap(T0', T2') -> {E00 E10 E20 E30} = E[g][0], {E02 E12 E22 E32} = E[g][2];  (T1', T3') -> E[g][1], E[g][3]
// (semantics probed on the part: tools/probes/permlane_probe.hip; the CPU emulator implements the same maps).
// MEASURED SLOWER than the strips and off by default (vtx_set_switch("epi_regs", 1)): ffn1 forward 59.1 -> 65.6 us, vocabulary
// projection 152 -> 161, step 23.59 -> 23.70 ms (profiles/r04_epilogue_register_transpose.txt).  With lane & 15 = matrix row a
// store instruction touches 16 rows and half of each 128-byte line (the other half follows with the next instruction); a strip
// drain writes 8 rows of whole lines per instruction.  What bounds this epilogue is the write path, not the LDS round trip.
template <int ACT, int WTM, int WTN, class T>
__device__ __forceinline__ void v3_lean_store_regs(const EpiStore<T, STATS_NONE>& ep, f32x4_t (&acc)[WTM / 16][WTN / 16], int mw, int nw,
                                                   int lane) {
    static_assert(WTN == 64, "four 16-column tiles per wave row: one per lane row after the transposition");
    constexpr int MT = WTM / 16;
    constexpr int W = sizeof(T) == 2 ? 2 : 4;                  // dwords per (tile, 4-column group)
    constexpr int NCH = 4 * W / 4;                             // 16-byte chunks per lane and step
    constexpr int EPV = 16 / (int)sizeof(T);
    const int g = lane >> 4, r = lane & 15;
    float4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = ep.bias ? *reinterpret_cast<const float4*>(ep.bias + nw + j * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float alpha = ep.alpha;
    const bool has_res = ep.residual != nullptr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const long row = mw + i * 16 + r;
        uint4 res[NCH];
        if (has_res) {
#pragma unroll
            for (int q = 0; q < NCH; ++q) res[q] = *reinterpret_cast<const uint4*>(ep.residual + row * ep.ldr + nw + 16 * g + q * EPV);
        }
        uint32_t t[4][W];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4_t v = acc[i][j] * alpha;
            v[0] += bv[j].x; v[1] += bv[j].y; v[2] += bv[j].z; v[3] += bv[j].w;
            if constexpr (ACT == ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            } else if constexpr (ACT == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if constexpr (sizeof(T) == 2) { t[j][0] = f2bf2(v[0], v[1]); t[j][1] = f2bf2(v[2], v[3]); }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) t[j][e] = __float_as_uint(v[e]);
            }
        }
        uint32_t u[4][W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const auto s01 = __builtin_amdgcn_permlane16_swap(t[0][w], t[1][w], false, false);
            const auto s23 = __builtin_amdgcn_permlane16_swap(t[2][w], t[3][w], false, false);
            const auto a = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
            const auto b = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
            u[0][w] = a[0]; u[2][w] = a[1]; u[1][w] = b[0]; u[3][w] = b[1];
        }
        T* const dst = ep.out + row * ep.ldc + nw + 16 * g + ep.split_off;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            uint4 w4;
            if constexpr (sizeof(T) == 2) w4 = make_uint4(u[2 * q][0], u[2 * q][1], u[2 * q + 1][0], u[2 * q + 1][1]);
            else w4 = make_uint4(u[q][0], u[q][1], u[q][2], u[q][3]);
            if (has_res) w4 = add16<T>(w4, res[q]);
            if (ep.nt) st16_nt(dst + q * EPV, u32x4_t{w4.x, w4.y, w4.z, w4.w});
            else *reinterpret_cast<uint4*>(dst + q * EPV) = w4;
        }
    }
}

// does the tile at (m0, n0) take the lean path?  (block-uniform)
template <int BM, int BN, class EP> __device__ __forceinline__ bool v3_lean_ok(const EP& ep, int m0, int n0) {
    if constexpr (V3Lean<EP>::OK) {
        const bool pd = ep.preact || ep.drop.thresh != 0u;      // pre-activation copy / dropout: the GELU instantiation only (the feed-forward layer)
        return m0 + BM <= ep.M && n0 + BN <= ep.N && (!pd || ep.act == ACT_GELU) && !ep.map_on &&
               (ep.act == ACT_NONE || ep.act == ACT_GELU || ep.act == ACT_RELU || ep.act == ACT_SOFTMAX_GRAD) && (ep.N & 3) == 0;
    } else {
        return false;
    }
}

// true when the block took the lean path
template <int BM, int BN, int WM, int WN, class EP>
__device__ __forceinline__ bool v3_lean_epilogue(const EP& ep, f32x4_t (&acc)[BM / WM / 16][BN / WN / 16], bf16_t* lds, int m0, int n0,
                                                 int lane, int wave, bool strips) {
    if constexpr (V3Lean<EP>::OK) {
        const bool lean = v3_lean_ok<BM, BN>(ep, m0, n0);
        if (!lean) return false;
        constexpr int WTM = BM / WM, WTN = BN / WN;
        const int mw = m0 + (wave / WN) * WTM, nw = n0 + (wave % WN) * WTN;
        const bool pd = ep.preact || ep.drop.thresh != 0u;
        if (!strips && ep.act != ACT_SOFTMAX_GRAD && !pd) {           // registers only: no stage memory is touched, no block barrier needed
            if (ep.act == ACT_NONE) v3_lean_store_regs<ACT_NONE, WTM, WTN>(ep, acc, mw, nw, lane);
            else if (ep.act == ACT_GELU) v3_lean_store_regs<ACT_GELU, WTM, WTN>(ep, acc, mw, nw, lane);
            else v3_lean_store_regs<ACT_RELU, WTM, WTN>(ep, acc, mw, nw, lane);
            return true;
        }
        // every wave is done with the stage memory (a raw barrier: __syncthreads() would also drain vmcnt, i.e. wait for the
        // NEXT tile's staged units of a persistent block to land before the first strip is written)
        VTX3_WAIT_LGKM(0);
        __builtin_amdgcn_s_barrier();
        if (ep.act == ACT_NONE) v3_lean_store<ACT_NONE, WTM, WTN>(ep, acc, lds, mw, nw, lane, wave);
        else if (ep.act == ACT_GELU && pd) v3_lean_store<ACT_GELU, WTM, WTN, typename EP::Out, true>(ep, acc, lds, mw, nw, lane, wave);
        else if (ep.act == ACT_GELU) v3_lean_store<ACT_GELU, WTM, WTN>(ep, acc, lds, mw, nw, lane, wave);
        else if (ep.act == ACT_SOFTMAX_GRAD) v3_lean_store<ACT_SOFTMAX_GRAD, WTM, WTN>(ep, acc, lds, mw, nw, lane, wave);
        else v3_lean_store<ACT_RELU, WTM, WTN>(ep, acc, lds, mw, nw, lane, wave);
        return true;
    } else {
        return false;
    }
}

// ------------------------------------------------------------------ 256 x 256: waves 2 (M) x 4 (N), wave tile 128 x 64
// Units per K tile: A0 A1 (sub-blocks of 64 rows), B0 B1 (sub-blocks of 32 rows); two LDS buffers E / O of four units.
// K tile t (buffer E, phases 1-4; t+1: buffer O, phases 5-8), per wave:
//   phase 1  reads b0 (4) then a0 (8)   MFMA a0 x b0     phase 2  reads b1 (4)   MFMA a0 x b1
//   phase 3  reads a1 (8)               MFMA a1 x b1     phase 4  no reads       MFMA a1 x b0
// Staging (one unit per phase), tile indices relative to the iteration's even tile t:
//   1: O.A1 <- t+1   2: E.B0 <- t+2   3: E.A0 <- t+2   4: E.B1 <- t+2   [vmcnt(6): O complete]
//   5: E.A1 <- t+2   6: O.B0 <- t+3   7: O.A0 <- t+3   8: O.B1 <- t+3   [vmcnt(6): E complete]
// WAR: E.B0 last read in phase 1 behind lgkmcnt(8) -> phase 2; E.A0 phase 1 -> 3; E.B1 phase 2 -> 4; E.A1 phase 3 -> 5; O alike.
template <class AL, class BL, class EP, bool LEAN = false, bool PERS = false>
__global__ __launch_bounds__(512, 2) void contraction_v3_256x256_kernel(AL al, BL bl, EP ep, int K, int tiles_n, int kt_per_split,
                                                                        int abl, unsigned long long* dbg, int vb_tiles) {
    constexpr int BM = 256, BN = 256, WM = 2, WN = 4, MT = 8, NT = 4;
    constexpr int BUF = 4 * V3_UNIT;
    typedef UnitStager<128, 64, AL> SA;
    typedef UnitStager<64, 32, BL> SB;
    HIP_DYNAMIC_SHARED(bf16_t, lds)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    V3_STAMP(0);
    // PERS (launch_v3: more tiles than CUs, no split-K, plain epilogue): one block per CU walks the virtual blocks L, L + G, ...;
    // the first K tile (and one unit of the second) of the NEXT tile are staged in front of the current tile's epilogue, whose
    // strips live in units the staging leaves alone -- the DMA latency, the stager set-up and the dispatch of a new block (what
    // the stamps call the prologue: 4 400 cycles, plus the launch of the block itself) disappear behind the epilogue's stores.
    // MEASURED NEUTRAL and off by default (vtx_set_switch("gen3_pers", 256)): the counted `s_waitcnt vmcnt` of the next tile's
    // first phases counts the epilogue's STORES too (one in-order counter for loads and stores on gfx9), so the K loop restarts
    // when the stores have drained -- which is what the epilogue costs (all 256 CUs flush 128 KiB each at the same moment).
    // Hiding that drain needs waves that never wait on vmcnt in the K loop: DMA issued by a producer wave.
    int tile, slice;
    int L = blockIdx.x;
    if constexpr (PERS) { tile = v3_vb_tile(abl, L, vb_tiles); slice = 0; }
    else v3_block_tile(abl, tile, slice);
    set_slice(ep, slice);
    int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int nkt = (K + 63) >> 6;
    const int kt0 = slice * kt_per_split;
    const int kt1 = kt0 + kt_per_split < nkt ? kt0 + kt_per_split : nkt;

    SA sa; SB sb;
    sa.init(al, m0, wave, lane);
    sb.init(bl, n0, wave, lane);
    bool staged = false;                 // the units of the current tile's prologue that do not collide with the strips are in flight
    f32x4_t acc[MT][NT];
    for (;;) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int PPT = EpiShape<BN, 8, EP>::PPT;
    float pre[PPT][4];
    epi_prefetch<BN, 8>(ep, pre, tid, n0);

    bf16_t* const E = lds;
    bf16_t* const O = lds + BUF;
    constexpr int A0 = 0, A1 = V3_UNIT, B0 = 2 * V3_UNIT, B1 = 3 * V3_UNIT;
    const int al0 = v3_lane_off(wm * 64, lane, 0), al1 = v3_lane_off(wm * 64, lane, 1);
    const int bl0 = v3_lane_off(wn * 32, lane, 0), bl1 = v3_lane_off(wn * 32, lane, 1);
    bf16x8_t fa[4][2], fb[2][2][2];
    if (V3_ABL(2)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i][0] = fa[i][1] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[q][j][0] = fb[q][j][1] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    }

#define V3_READ_A(X, U)                                                        \
    if (!V3_ABL(2)) _Pragma("unroll") for (int i = 0; i < 4; ++i) {                             \
        fa[i][0] = v3_ld((X) + (U) + al0 + i * 1024);                           \
        fa[i][1] = v3_ld((X) + (U) + al1 + i * 1024);                           \
    }
#define V3_READ_B(X, U, Q)                                                     \
    if (!V3_ABL(2)) _Pragma("unroll") for (int j = 0; j < 2; ++j) {                             \
        fb[Q][j][0] = v3_ld((X) + (U) + bl0 + j * 1024);                        \
        fb[Q][j][1] = v3_ld((X) + (U) + bl1 + j * 1024);                        \
    }
// (a tile past the end of the block's K range was staged as zeros: its MFMAs run and add nothing -- a branch around them
//  leaves hipcc with fragment reads it believes pending on one path, and it pads every later read with an s_waitcnt)
#define V3_MMA(UA, UB)                                                                                                \
    if (!V3_ABL(1)) _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                acc[(UA) * 4 + i][(UB) * 2 + j] =                                                                     \
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[UB][j][h], fa[i][h], acc[(UA) * 4 + i][(UB) * 2 + j], 0, 0, 0);

    if (kt0 < kt1 && !V3_ABL(64)) {
        // prologue: the whole first tile, three units of the second
        if (!staged) {
            sb.template issue<0>(bl, kt0 * 64, true, E + B0, wave);
            sa.template issue<0>(al, kt0 * 64, true, E + A0, wave);
            sb.template issue<1>(bl, kt0 * 64, true, E + B1, wave);
            sa.template issue<1>(al, kt0 * 64, true, E + A1, wave);
            const bool v1 = kt0 + 1 < kt1;
            sb.template issue<0>(bl, (kt0 + 1) * 64, v1, O + B0, wave);
            sa.template issue<0>(al, (kt0 + 1) * 64, v1, O + A0, wave);
            sb.template issue<1>(bl, (kt0 + 1) * 64, v1, O + B1, wave);
        } else {                                              // E and O.A0 went out in front of the previous tile's epilogue
            const bool v1 = kt0 + 1 < kt1;
            sb.template issue<0>(bl, (kt0 + 1) * 64, v1, O + B0, wave);
            sb.template issue<1>(bl, (kt0 + 1) * 64, v1, O + B1, wave);
        }
        VTX3_WAIT_VM(6);
        __builtin_amdgcn_s_barrier();
        V3_STAGGER(wave >= 4);                                // the second wave group runs one barrier behind
        V3_STAMP(1);
        for (int kt = kt0; kt < kt1; kt += 2) {
            const bool v1 = kt + 1 < kt1, v2 = kt + 2 < kt1, v3 = kt + 3 < kt1;
            const int k1 = (kt + 1) * 64, k2 = (kt + 2) * 64, k3 = (kt + 3) * 64;
            // ---- phase 1
            V3_READ_B(E, B0, 0)
            VTX3_FENCE();
            V3_READ_A(E, A0)
            if (!V3_ABL(4)) sa.template issue<1>(al, k1, v1, O + A1, wave);
            VTX3_FENCE();
            VTX3_WAIT_LGKM(8);
            V3_COMPUTE_BEGIN()
            V3_MMA(0, 0)
            V3_COMPUTE_END()
            // ---- phase 2
            V3_READ_B(E, B1, 1)
            if (!V3_ABL(4)) sb.template issue<0>(bl, k2, v2, E + B0, wave);
            V3_COMPUTE_BEGIN()
            V3_MMA(0, 1)
            V3_COMPUTE_END()
            // ---- phase 3
            V3_READ_A(E, A1)
            if (!V3_ABL(4)) sa.template issue<0>(al, k2, v2, E + A0, wave);
            V3_COMPUTE_BEGIN()
            V3_MMA(1, 1)
            V3_COMPUTE_END()
            // ---- phase 4
            if (!V3_ABL(4)) sb.template issue<1>(bl, k2, v2, E + B1, wave);
            VTX3_FENCE();
            VTX3_WAIT_VM(6);
            V3_COMPUTE_BEGIN()
            V3_MMA(1, 0)
            V3_COMPUTE_END()
            // ---- phase 5
            V3_READ_B(O, B0, 0)
            VTX3_FENCE();
            V3_READ_A(O, A0)
            if (!V3_ABL(4)) sa.template issue<1>(al, k2, v2, E + A1, wave);
            VTX3_FENCE();
            VTX3_WAIT_LGKM(8);
            V3_COMPUTE_BEGIN()
            V3_MMA(0, 0)
            V3_COMPUTE_END()
            // ---- phase 6
            V3_READ_B(O, B1, 1)
            if (!V3_ABL(4)) sb.template issue<0>(bl, k3, v3, O + B0, wave);
            V3_COMPUTE_BEGIN()
            V3_MMA(0, 1)
            V3_COMPUTE_END()
            // ---- phase 7
            V3_READ_A(O, A1)
            if (!V3_ABL(4)) sa.template issue<0>(al, k3, v3, O + A0, wave);
            V3_COMPUTE_BEGIN()
            V3_MMA(1, 1)
            V3_COMPUTE_END()
            // ---- phase 8
            if (!V3_ABL(4)) sb.template issue<1>(bl, k3, v3, O + B1, wave);
            VTX3_FENCE();
            VTX3_WAIT_VM(6);
            V3_COMPUTE_BEGIN()
            V3_MMA(1, 0)
            V3_COMPUTE_END()
        }
        VTX3_WAIT_VM(0);                                      // the zero-fill units staged past the end
        V3_STAGGER(wave < 4);                                 // the first group catches up
        V3_STAMP(2);
    }
#undef V3_READ_A
#undef V3_READ_B
#undef V3_MMA
    if (V3_ABL(128)) return;
    // ---- the next virtual block: its first K tile goes out NOW when this tile's epilogue is the strip epilogue (which then
    // uses units 5, 6 and the head of 7: O.A1, O.B0, O.B1 -- the staging below fills units 0-4: E and O.A0)
    int Ln = 0, tile_n = 0, m0n = 0, n0n = 0;
    bool more = false;
    staged = false;
    if constexpr (PERS) {
        Ln = L + gridDim.x;
        more = Ln < vb_tiles;
        if (more) {
            tile_n = v3_vb_tile(abl, Ln, vb_tiles);
            m0n = (tile_n / tiles_n) * BM; n0n = (tile_n % tiles_n) * BN;
            if (v3_lean_ok<BM, BN>(ep, m0, n0) && (abl & 256) == 0 && kt0 < kt1) {
                sa.init(al, m0n, wave, lane);                  // (every wave is past its last fragment read: the loop's last barrier)
                sb.init(bl, n0n, wave, lane);
                sb.template issue<0>(bl, kt0 * 64, true, lds + 2 * V3_UNIT, wave);
                sa.template issue<0>(al, kt0 * 64, true, lds + 0 * V3_UNIT, wave);
                sb.template issue<1>(bl, kt0 * 64, true, lds + 3 * V3_UNIT, wave);
                sa.template issue<1>(al, kt0 * 64, true, lds + 1 * V3_UNIT, wave);
                sa.template issue<0>(al, (kt0 + 1) * 64, kt0 + 1 < kt1, lds + BUF + 0 * V3_UNIT, wave);
                staged = true;
            }
        }
    }
    if (!v3_lean_epilogue<BM, BN, WM, WN>(ep, acc, lds + (staged ? 5 * V3_UNIT : 0), m0, n0, lane, wave, (abl & 256) == 0))
        tile_epilogue<BM, BN, WM, WN, 2 * BUF * 2, EP, LEAN>(ep, acc, pre, lds, m0, n0, tile / tiles_n, tile % tiles_n, tid, lane, wave);
    V3_STAMP(3);
    if (!more) break;
    VTX3_WAIT_LGKM(0);                                         // every wave is done with the epilogue's LDS (strips: units 6 / 7 are staged
    __builtin_amdgcn_s_barrier();                              // next; general epilogue: all of it) -- no vmcnt drain: the stores and the staged units stay in flight
    if (!staged) { sa.init(al, m0n, wave, lane); sb.init(bl, n0n, wave, lane); }
    L = Ln; tile = tile_n; m0 = m0n; n0 = n0n;
    }
}

// ------------------------------------------------------------------ 256 x 128: waves 4 (M) x 2 (N), wave tile 64 x 64
// Units per K tile: A0 A1 (sub-blocks of 32 rows), B (the wave's 64 columns; all 128 rows of the tile); THREE LDS buffers
// of three units (144 KiB): K tile t lives in buffer t % 3 and is worked off in two phases,
//   phase 1  reads b (8) then a0 (4)    MFMA a0 x b      phase 2  reads a1 (4)    MFMA a1 x b
// while tile t+2 is staged into buffer (t+2) % 3 = (t-1) % 3 (three wave-instructions per phase: B, A0.0 | A0.1, A1; every
// unit was last read two phases earlier) and `vmcnt(6)` in phase 2 leaves exactly that tile in flight: tile t+1 is complete.
template <class AL, class BL, class EP, bool LEAN = false>
__global__ __launch_bounds__(512, 2) void contraction_v3_256x128_kernel(AL al, BL bl, EP ep, int K, int tiles_n, int kt_per_split,
                                                                        int abl, unsigned long long* dbg, int /*vb_tiles: 256x256 only*/) {
    constexpr int BM = 256, BN = 128, WM = 4, WN = 2, MT = 4, NT = 4;
    constexpr int BUF = 3 * V3_UNIT;
    typedef UnitStager<64, 32, AL> SA;
    typedef UnitStager<64, 64, BL> SB;
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
    sa.init(al, m0, wave, lane);
    sb.init(bl, n0, wave, lane);
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int PPT = EpiShape<BN, 8, EP>::PPT;
    float pre[PPT][4];
    epi_prefetch<BN, 8>(ep, pre, tid, n0);

    constexpr int A0 = 0, A1 = V3_UNIT, B = 2 * V3_UNIT;
    const int al0 = v3_lane_off(wm * 32, lane, 0), al1 = v3_lane_off(wm * 32, lane, 1);
    const int bl0 = v3_lane_off(wn * 64, lane, 0), bl1 = v3_lane_off(wn * 64, lane, 1);
    bf16x8_t fa[2][2], fb[4][2];
    if (V3_ABL(2)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i][0] = fa[i][1] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j][0] = fb[j][1] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    }

#define V3_READ_A(X, U)                                                        \
    if (!V3_ABL(2)) _Pragma("unroll") for (int i = 0; i < 2; ++i) {                             \
        fa[i][0] = v3_ld((X) + (U) + al0 + i * 1024);                           \
        fa[i][1] = v3_ld((X) + (U) + al1 + i * 1024);                           \
    }
#define V3_READ_B(X)                                                           \
    if (!V3_ABL(2)) _Pragma("unroll") for (int j = 0; j < 4; ++j) {                             \
        fb[j][0] = v3_ld((X) + B + bl0 + j * 1024);                             \
        fb[j][1] = v3_ld((X) + B + bl1 + j * 1024);                             \
    }
#define V3_MMA(UA)                                                                                                    \
    if (!V3_ABL(1)) _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                 \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
                acc[(UA) * 2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][h], fa[i][h], acc[(UA) * 2 + i][j], 0, 0, 0);
// one K tile: X = its buffer, Y = the buffer tile +2 is staged into, V = tile +2 exists
#define V3_TILE(X, Y, K2, V)                                                   \
    V3_READ_B(X)                                                               \
    VTX3_FENCE();                                                              \
    V3_READ_A(X, A0)                                                           \
    if (!V3_ABL(4)) { sb.template issue<0>(bl, K2, V, (Y) + B, wave);          \
                      sa.template issue1<0, 0>(al, K2, V, (Y) + A0, wave); }   \
    V3_COMPUTE_BEGIN()                                                         \
    V3_MMA(0)                                                                  \
    V3_COMPUTE_END()                                                           \
    V3_READ_A(X, A1)                                                           \
    if (!V3_ABL(4)) { sa.template issue1<0, 1>(al, K2, V, (Y) + A0, wave);     \
                      sa.template issue<1>(al, K2, V, (Y) + A1, wave); }       \
    VTX3_FENCE();                                                              \
    VTX3_WAIT_VM(6);                                                           \
    V3_COMPUTE_BEGIN()                                                         \
    V3_MMA(1)                                                                  \
    V3_COMPUTE_END()

    if (kt0 < kt1 && !V3_ABL(64)) {
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
            V3_TILE(X0, X2, (kt + 2) * 64, kt + 2 < kt1)
            V3_TILE(X1, X0, (kt + 3) * 64, kt + 3 < kt1)
            V3_TILE(X2, X1, (kt + 4) * 64, kt + 4 < kt1)
        }
        VTX3_WAIT_VM(0);
        V3_STAGGER(wave < 4);
        V3_STAMP(2);
    }
#undef V3_READ_A
#undef V3_READ_B
#undef V3_MMA
#undef V3_TILE
    if (V3_ABL(128)) return;
    if (!v3_lean_epilogue<BM, BN, WM, WN>(ep, acc, lds, m0, n0, lane, wave, (abl & 256) == 0))
        tile_epilogue<BM, BN, WM, WN, 3 * BUF * 2, EP, LEAN>(ep, acc, pre, lds, m0, n0, tile / tiles_n, tile % tiles_n, tid, lane, wave);
    V3_STAMP(3);
}

#include "gemm_v3mc.h"

// ------------------------------------------------------------------ host side
extern unsigned long long* g_vtx_dbg;   // measurement builds: time-stamp buffer of the generation-3 kernels (vtx_set_debug_buffer), else null
extern int g_vtx_sw_gen3_pers;   // vtx_set_switch("gen3_pers"): 0 = off, n = blocks of the persistent 256x256 form (taken from more than n tiles; 256 = one per CU)
extern int g_vtx_sw_gen3;        // vtx_set_switch("gen3"): 0 = generation 3 only when forced by the tile override (20 / 21), 1 = automatic

template <int BN, class AL, class BL, class EP>
inline int launch_v3(const AL& al, const BL& bl, const EP& ep_in, int M, int N, int K, int split_k, hipStream_t st) {
    constexpr int BM = 256, WN = BN == 256 ? 4 : 2;
    const int tiles_m = vtx_cdiv(M, BM), tiles_n = vtx_cdiv(N, BN);
    const int nkt = vtx_cdiv(K, 64);
    if (split_k < 1) split_k = 1;
    if (split_k > nkt) split_k = nkt > 0 ? nkt : 1;
    const int per = vtx_cdiv(nkt, split_k);
    split_k = per > 0 ? vtx_cdiv(nkt, per) : 1;
    constexpr size_t lds_bytes = BN == 256 ? 2 * 4 * V3_UNIT * 2 : 3 * 3 * V3_UNIT * 2;
    auto pick = [](auto lean) {
        if constexpr (AL::MC) {                 // k-major pairs (weight gradients): gemm_v3mc.h, plain fp32 epilogue
            if constexpr (BN == 256) return contraction_v3mc_256x256_kernel<AL, BL, EP>;
            else return contraction_v3mc_256x128_kernel<AL, BL, EP>;
        } else {
            if constexpr (BN == 256) return contraction_v3_256x256_kernel<AL, BL, EP, decltype(lean)::value>;
            else return contraction_v3_256x128_kernel<AL, BL, EP, decltype(lean)::value>;
        }
    };
    auto kern = pick(std::false_type{});
    if constexpr (EP::STATS && EP::STAGED) {
        if (lean_host_ok(ep_in, M, N, BM, BN)) kern = pick(std::true_type{});
    }
    // persistent form (256x256, row-major operands, plain epilogue, no split-K, more tiles than CUs): one block per CU
    bool pers = false;
    if constexpr (!AL::MC && BN == 256 && V3Lean<EP>::OK && sizeof(typename EP::Out) == 2) {   // (fp32 strips: 68 KiB, no room beside the staged units)
        pers = g_vtx_sw_gen3_pers && split_k == 1 && tiles_m * tiles_n > g_vtx_sw_gen3_pers;
        if (pers) kern = contraction_v3_256x256_kernel<AL, BL, EP, false, true>;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)pick(std::false_type{}), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if constexpr (EP::STATS && EP::STAGED) hipFuncSetAttribute((const void*)pick(std::true_type{}), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if constexpr (!AL::MC && BN == 256 && V3Lean<EP>::OK && sizeof(typename EP::Out) == 2)
            hipFuncSetAttribute((const void*)contraction_v3_256x256_kernel<AL, BL, EP, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    const int vb_tiles = tiles_m * tiles_n;
    dim3 grid(pers ? g_vtx_sw_gen3_pers : vb_tiles, split_k), block(512);
    g_vtx_last_colgroups = tiles_n * WN;
    EP ep = ep_in;
    if constexpr (EP::STAGED) ep.nt = vtx_nt_policy((double)M * N * sizeof(typename EP::Out));
    bool prof = g_vtx_prof_on != 0;
    if (prof) {
        static const int cls = vtx_prof_register(__PRETTY_FUNCTION__);
        prof = g_vtx_prof_only < 0 || g_vtx_prof_only == cls;
        if (prof) {
            hipEvent_t e0, e1;
            vtx_prof_events(cls, 2.0 * M * N * K, 2.0 * (algo_elems(al) + algo_elems(bl)) + epi_bytes(ep, (double)M * N, split_k), &e0, &e1);
            hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds_bytes, st, e0, e1, 0, al, bl, ep, K, tiles_n, per, g_vtx_ablate, g_vtx_dbg, vb_tiles);
            return tiles_m;
        }
    }
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, al, bl, ep, K, tiles_n, per, g_vtx_ablate, g_vtx_dbg, vb_tiles);
    return tiles_m;
}
