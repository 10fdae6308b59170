// Dense GEMM entry points (text-head linears, 1x1 stride-1 convolutions, tied output
// projection).  Kernel: gemm_kernel.h.
//
// Replaces aten::linear / addmm / mm reached from
//   /root/reference/virtex/modules/textual_heads.py:245 (visual_projection), :270-275
//   (nn.TransformerDecoder: in_proj, out_proj, linear1, linear2), :277 (tied output), and
//   the 1x1 convolutions of torchvision's Bottleneck (visual_backbones.py:68-74).
#include <stdlib.h>

#include "gemm_kernel.h"

using namespace vtxg;

vtxg::EpiStore<float> vtx_splitk_epilogue(float* C, long ldc, float alpha, int M, int N, int split_k, float* ws);
void vtx_splitk_reduce(const float* ws, int S, int M, int N, float* C, long ldc, hipStream_t st);
int vtx_pick_split_k(int M, int N, int K, int bk, long ws_floats, int gather = 0);
int vtx_expand1x1_try(int M, int N, int K, const void* A, long lda, const void* W, long ldw, void* Y, long ldy,
                      const float* shift, float* parts, hipStream_t st);

namespace {

template <class T, class TO>
int gemm_nt_t(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
              const float* bias, const void* residual, long ldr, void* preact, int act, float alpha,
              Dropout drop, float* stat_parts, const float* stat_shift, int* stat_strips, hipStream_t st) {
    int strips = 0;
    auto mk_a = [&](auto& a) { a.p = (const T*)A; a.ld = lda; a.rows = M; a.K = K; };
    auto mk_b = [&](auto& b) { b.p = (const T*)B; b.ld = ldb; b.rows = N; b.K = K; };
    bool done = false;
    if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) {
        if (stat_parts && !bias && !residual && !preact && act == ACT_NONE && alpha == 1.f && drop.thresh == 0u) {
            // write-heavy 1x1 "expand" convolutions: the streaming kernel (expand1x1.hip) when it takes the problem
            const int s = vtx_expand1x1_try(M, N, K, A, lda, B, ldb, C, ldc, stat_shift, stat_parts, st);
            if (s < 0) return s;
            if (s > 0) { if (stat_strips) *stat_strips = s; return VTX_OK; }
        }
        if (stat_parts && !bias && !preact && act == ACT_NONE && drop.thresh == 0u && N % 8 == 0 && ldc == N) {     // statistics epilogues are compiled without the bias / activation / dropout paths
            EpiStore<TO, STATS_FWD> ep{(TO*)C, ldc, bias, (const TO*)residual, ldr, (TO*)preact, act, alpha, drop, M, N};
            ep.stat_parts = stat_parts; ep.stat_shift = stat_shift;
            strips = launch_auto<T, PlainKC, PlainKC>(mk_a, mk_b, ep, M, N, K, 1, st);
            done = true;
        }
    }
    if (!done) {
        EpiStore<TO> ep{(TO*)C, ldc, bias, (const TO*)residual, ldr, (TO*)preact, act, alpha, drop, M, N};
        launch_auto<T, PlainKC, PlainKC>(mk_a, mk_b, ep, M, N, K, 1, st);
    }
    if (stat_strips) *stat_strips = strips;
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

template <class T>
int gemm_tn_t(int M, int N, int K, const void* A, long lda, const void* B, long ldb, float* C, long ldc,
              float alpha, int split_k, float* ws, hipStream_t st) {
    EpiStore<float> ep = vtx_splitk_epilogue(C, ldc, alpha, M, N, split_k, ws);
    launch_auto<T, PlainMC, PlainMC>(
        [&](auto& a) { a.p = (const T*)A; a.ld = lda; a.rows = M; a.K = K; },
        [&](auto& b) { b.p = (const T*)B; b.ld = ldb; b.rows = N; b.K = K; }, ep, M, N, K, split_k, st);
    if (split_k > 1) vtx_splitk_reduce(ws, split_k, M, N, C, ldc, st);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// shared with conv_dgrad.hip: fills the statistics half of a STATS_BWD epilogue from the C-ABI struct
void vtx_fill_bn_bwd(vtxg::EpiStore<bf16_t, vtxg::STATS_BWD>& ep, const VtxBnBwdFusion* f, long ld, float* parts) {
    ep.stat_parts = parts;
    ep.bn_x = (const bf16_t*)f->x; ep.ldx = ld;
    ep.bn_y = (const bf16_t*)f->ymask; ep.ldy = ld;
    ep.bn_ybits = f->ybits;
    ep.bn_mean = f->mean; ep.bn_rstd = f->rstd; ep.bn_gamma = f->gamma; ep.bn_beta = f->beta;
}
int vtx_check_bn_bwd(const char* who, const VtxBnBwdFusion* f, int M, int N) {
    VTX_CHECK(f->x && f->mean && f->rstd && f->parts, VTX_ERR_ARG, "%s: BatchNorm fusion needs x, mean, rstd and parts", who);
    VTX_CHECK(!((f->ymask || f->ybits) && f->beta), VTX_ERR_ARG, "%s: pass either ymask / ybits or gamma/beta for the ReLU mask", who);
    VTX_CHECK(!f->beta || f->gamma, VTX_ERR_ARG, "%s: a recomputed mask needs gamma and beta", who);
    VTX_CHECK(N % 8 == 0 && ((uintptr_t)f->x & 15) == 0 && (!f->ymask || ((uintptr_t)f->ymask & 15) == 0), VTX_ERR_SHAPE,
              "%s: fused BatchNorm backward needs N %% 8 == 0 and 16-byte aligned tensors", who);
    VTX_CHECK((long)(vtx_cdiv(M, 64) + 4) * 2 * N <= f->parts_cap, VTX_ERR_WORKSPACE,
              "%s: parts holds %ld floats, (ceil(M/64)+4)*2*N = %ld needed", who, f->parts_cap, (long)(vtx_cdiv(M, 64) + 4) * 2 * N);
    return VTX_OK;
}

vtxg::EpiStore<float> vtx_splitk_epilogue(float* C, long ldc, float alpha, int M, int N, int split_k, float* ws);
void vtx_splitk_reduce(const float* ws, int S, int M, int N, float* C, long ldc, hipStream_t st);
float* vtx_splitk_region(float* ws, long ws_floats, long* cap, hipStream_t st);

// Split-K policy for the weight-gradient GEMMs: enough slices to give every CU ~2 blocks, never fewer than 8 K-steps per slice, bounded by the workspace
// ([slices][M][N] fp32 partial sums).
extern int g_vtx_sw_splitk_blocks;
int vtx_pick_split_k(int M, int N, int K, int bk, long ws_floats, int gather) {
    if (bk == 32 && vtxg::g_vtx_contraction_generation >= 2) {          // generation 3 plans its own slices (gemm_kernel.h)
        int s3 = 1;
        if (vtxg::plan_gen3_mc(M, N, K, gather != 0, &s3) && (s3 == 1 || (long)s3 * M * N <= ws_floats)) return s3;
    }
    long tiles = (long)vtx_cdiv(M, 128) * vtx_cdiv(N, N <= 64 ? 64 : 128);
    // launch_auto takes one 64x256 tile -- only on the bf16 LDS-DMA kernel (bk == 32) and without a forced tile
    if (vtxg::g_vtx_sw_tile64x256 && bk == 32 && vtxg::g_vtx_contraction_generation >= 2 && vtxg::g_vtx_tile_override < 0 &&
        M <= 64 && N > 128 && N <= 256) tiles = 1;
    const int nkt = vtx_cdiv(K, bk);
    const long target = g_vtx_sw_splitk_blocks;          // VIRTEX_AMD_SPLITK_BLOCKS / vtx_set_switch("splitk_blocks")
    long s = (target + tiles - 1) / tiles;
    if (s > nkt / 8) s = nkt / 8;
    const long cap = ws_floats / ((long)M * N);
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    // the launch code rounds the slice length up; make the slice count exact
    const int per = vtx_cdiv(nkt, (int)s);
    return vtx_cdiv(nkt, per);
}

namespace {
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// C[m][n] += sum_s ws[s][m][n]      (alpha already applied by the GEMM epilogue)
// Block = (256/G float4 columns) x (G slice groups): a small weight matrix split hundreds of ways (the 1x1
// convolutions of stage 1: K = 802,816) would otherwise be a handful of threads walking a serial chain of
// dependent loads.  Each group sums its slices four loads at a time; LDS folds the groups.
template <int G>
__device__ __forceinline__ void splitk_reduce_body(const float* __restrict__ ws, int S, long MN, int N, float* __restrict__ C, long ldc,
                                                   int block, int nblocks, float4* red) {
    constexpr int COLS = 256 / G;
    const int col = threadIdx.x % COLS, grp = threadIdx.x / COLS;
    const long nv = MN / 4;
    for (long i0 = (long)block * COLS; i0 < nv; i0 += (long)nblocks * COLS) {
        const long i = i0 + col;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nv) {
            int s2 = grp;
            for (; s2 + 3 * G < S; s2 += 4 * G) {
                const float4 b0 = reinterpret_cast<const float4*>(ws + (long)s2 * MN)[i];
                const float4 b1 = reinterpret_cast<const float4*>(ws + (long)(s2 + G) * MN)[i];
                const float4 b2 = reinterpret_cast<const float4*>(ws + (long)(s2 + 2 * G) * MN)[i];
                const float4 b3 = reinterpret_cast<const float4*>(ws + (long)(s2 + 3 * G) * MN)[i];
                a = f4add(a, f4add(f4add(b0, b1), f4add(b2, b3)));
            }
            for (; s2 < S; s2 += G) a = f4add(a, reinterpret_cast<const float4*>(ws + (long)s2 * MN)[i]);
        }
        if (G > 1) {
            __syncthreads();
            red[threadIdx.x] = a;
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int g = 1; g < G; ++g) a = f4add(a, red[g * COLS + col]);
            }
        }
        if (grp == 0 && i < nv) {
            const long e = i * 4, m = e / N, n = e - m * N;      // N % 4 == 0: a float4 never straddles rows
            float4* dst = reinterpret_cast<float4*>(C + m * ldc + n);
            *dst = f4add(*dst, a);
        }
    }
}
template <int G>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, long MN, int N,
                                                            float* __restrict__ C, long ldc) {
    __shared__ float4 red[G > 1 ? 256 : 1];
    splitk_reduce_body<G>(ws, S, MN, N, C, ldc, blockIdx.x, gridDim.x, red);
}

// ---- several reductions in ONE launch (round 6).  The weight gradients of one Bottleneck are three or four split-K
// contractions, each followed by its own 7-10 us reduce launch (68 per step); between vtx_splitk_batch_begin() and
// vtx_splitk_batch_end(stream) the library carves the split-K workspace into consecutive regions (every contraction keeps its
// partial tiles alive) and defers the reductions: ONE launch folds them all -- the same sums in the same order (bit-identical
// results), block ranges per reduction from a small table passed by value.
constexpr int VTX_SPLITK_BATCH_MAX = 8;
struct SplitKBatchEntry { const float* ws; float* C; long MN, ldc; int S, N, G, first_block; };
struct SplitKBatchArgs { SplitKBatchEntry e[VTX_SPLITK_BATCH_MAX]; int n; };
__global__ __launch_bounds__(256) void splitk_reduce_batched_kernel(SplitKBatchArgs a) {
    __shared__ float4 red[256];
    int k = 0;
#pragma unroll
    for (int j = 1; j < VTX_SPLITK_BATCH_MAX; ++j)
        if (j < a.n && (int)blockIdx.x >= a.e[j].first_block) k = j;
    const SplitKBatchEntry& e = a.e[k];
    const int nb = (k + 1 < a.n ? a.e[k + 1].first_block : (int)gridDim.x) - e.first_block, b = (int)blockIdx.x - e.first_block;
    if (e.G == 1) splitk_reduce_body<1>(e.ws, e.S, e.MN, e.N, e.C, e.ldc, b, nb, red);
    else if (e.G == 4) splitk_reduce_body<4>(e.ws, e.S, e.MN, e.N, e.C, e.ldc, b, nb, red);
    else splitk_reduce_body<16>(e.ws, e.S, e.MN, e.N, e.C, e.ldc, b, nb, red);
}

struct SplitKBatchState {
    bool on = false;
    float* base = nullptr;       // the workspace the regions are carved from (one per (device, stream): the caller's)
    long used = 0;               // floats of it that hold partial tiles of pending reductions
    SplitKBatchArgs args{};
    double bytes = 0;
    hipStream_t stream = nullptr; // the stream the pending contractions were issued on: their reduction goes there, whatever happens
};
thread_local SplitKBatchState t_batch;

void splitk_plan(long MN, int S, int* G_out, long* blocks_out) {
    const long nv = MN / 4;
    // widen over the slices until the launch has ~1k blocks (or the slices run out)
    int G = 1;
    while (G < 16 && G * 4 <= S && (nv * G + 255) / 256 < 1024) G *= 4;
    long g = (nv * G + 255) / 256;
    if (g > 4096) g = 4096;
    *G_out = G; *blocks_out = g;
}

int splitk_batch_flush() {
    SplitKBatchState& b = t_batch;
    hipStream_t st = b.stream;
    if (b.args.n > 0) {
        int total = 0;
        for (int j = 0; j < b.args.n; ++j) {
            int G; long g;
            splitk_plan(b.args.e[j].MN, b.args.e[j].S, &G, &g);
            b.args.e[j].G = G; b.args.e[j].first_block = total; total += (int)g;
        }
        VTX_KLAUNCH("splitk_reduce", 0, b.bytes, splitk_reduce_batched_kernel, dim3(total), dim3(256), 0, st, b.args);
    }
    b.args.n = 0; b.used = 0; b.bytes = 0;
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
}  // namespace

vtxg::EpiStore<float> vtx_splitk_epilogue(float* C, long ldc, float alpha, int M, int N, int split_k, float* ws) {
    using namespace vtxg;
    if (split_k > 1)   // slices write plain partial tiles into the workspace
        return EpiStore<float>{ws, (long)N, nullptr, nullptr, 0, nullptr, ACT_NONE, alpha, make_dropout(0.f, 0), M, N,
                               (long)M * N};
    // one slice: read-modify-write of the fp32 gradient itself (tiles are disjoint: no atomics needed)
    return EpiStore<float>{C, ldc, nullptr, C, ldc, nullptr, ACT_NONE, alpha, make_dropout(0.f, 0), M, N, 0};
}

void vtx_splitk_reduce_now(const float* ws, int S, int M, int N, float* C, long ldc, hipStream_t st) {
    const long MN = (long)M * N;
    int G; long g;
    splitk_plan(MN, S, &G, &g);
    if (G == 1) VTX_KLAUNCH("splitk_reduce", 0, 4.0 * MN * (S + 2), splitk_reduce_kernel<1>, dim3((int)g), dim3(256), 0, st, ws, S, MN, N, C, ldc);
    else if (G == 4) VTX_KLAUNCH("splitk_reduce", 0, 4.0 * MN * (S + 2), splitk_reduce_kernel<4>, dim3((int)g), dim3(256), 0, st, ws, S, MN, N, C, ldc);
    else VTX_KLAUNCH("splitk_reduce", 0, 4.0 * MN * (S + 2), splitk_reduce_kernel<16>, dim3((int)g), dim3(256), 0, st, ws, S, MN, N, C, ldc);
}

// Inside a batch (vtx_splitk_batch_begin ... _end) a reduction whose partial tiles sit in the region vtx_splitk_region handed
// out is DEFERRED to the batch's single launch; anything else (a caller-owned partial buffer that may be freed before the
// flush: vtx_partials_reduce_acc) runs at once.
void vtx_splitk_reduce(const float* ws, int S, int M, int N, float* C, long ldc, hipStream_t st) {
    SplitKBatchState& b = t_batch;
    const long MN = (long)M * N;
    if (b.on && b.base && ws == b.base + b.used && MN % 4 == 0) {
        SplitKBatchEntry& e = b.args.e[b.args.n++];
        e.ws = ws; e.C = C; e.MN = MN; e.ldc = ldc; e.S = S; e.N = N; e.G = 1; e.first_block = 0;
        b.used += (((long)S * MN + 1023) / 1024) * 1024;
        b.bytes += 4.0 * MN * (S + 2);
        b.stream = st;
        if (b.args.n == VTX_SPLITK_BATCH_MAX) (void)splitk_batch_flush();
        return;
    }
    vtx_splitk_reduce_now(ws, S, M, N, C, ldc, st);
}

// The workspace a split-K contraction may use right now: all of it outside a batch; inside one, what the pending reductions
// leave free (a batch that has eaten three quarters of the workspace is flushed first, so that the slice policy keeps room).
float* vtx_splitk_region(float* ws, long ws_floats, long* cap, hipStream_t st) {
    SplitKBatchState& b = t_batch;
    if (!ws) { *cap = 0; return ws; }
    if (!b.on) { *cap = ws_floats; return ws; }
    // another workspace or another stream (a contraction on the branch stream between two on the weight-gradient stream): what is
    // pending goes out on ITS stream first -- a batch never mixes streams
    if (b.base != ws || (b.args.n > 0 && b.stream != st)) { (void)splitk_batch_flush(); b.base = ws; }
    if (ws_floats - b.used < ws_floats / 4) (void)splitk_batch_flush();
    *cap = ws_floats - b.used;
    return ws + b.used;
}

// (a batch that was begun and never ended -- an exception between the two in the caller -- belongs to a step that died: its
// pending entries are dropped, not launched; batches do not nest)
extern "C" int vtx_splitk_batch_begin(void) {
    SplitKBatchState& b = t_batch;
    b.on = true; b.base = nullptr; b.used = 0; b.args.n = 0; b.bytes = 0; b.stream = nullptr;
    return VTX_OK;
}
// pending reductions NOW, the batch stays open: for a caller that reads a gradient (or any split-K result) in the middle of a batch
extern "C" int vtx_splitk_batch_flush(void* stream) { (void)stream; return splitk_batch_flush(); }
extern "C" int vtx_splitk_batch_end(void* stream) {
    (void)stream;                 // the pending reductions go to the stream their contractions ran on (normally this one)
    const int rc = splitk_batch_flush();
    t_batch.on = false; t_batch.base = nullptr;
    return rc;
}

extern "C" int vtx_gemm_nt(int dtype, int M, int N, int K, const void* A, long lda, const void* B,
                           long ldb, void* C, long ldc, const float* bias, const void* residual,
                           long ldr, void* preact, int act, float alpha, float p_drop, uint64_t seed,
                           int out_f32, float* bn_parts, const float* bn_shift, int* bn_strips, void* stream) {
    VTX_CHECK(A && B && C, VTX_ERR_ARG, "gemm_nt: null pointer");
    VTX_CHECK(M >= 0 && N > 0 && K > 0, VTX_ERR_ARG, "gemm_nt: bad shape %dx%dx%d", M, N, K);
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "gemm_nt: bad dtype %d", dtype);
    VTX_CHECK(K % vec == 0 && lda % vec == 0 && ldb % vec == 0 && N % 4 == 0 && ldc % 4 == 0 &&
                  (!residual || ldr % 4 == 0),
              VTX_ERR_SHAPE, "gemm_nt: K/lda/ldb must be multiples of %d, N/ldc/ldr of 4 (M=%d N=%d K=%d)",
              vec, M, N, K);
    VTX_CHECK(aligned16(A) && aligned16(B) && aligned16(C), VTX_ERR_SHAPE, "gemm_nt: operands must be 16-byte aligned");
    if (bn_strips) *bn_strips = 0;
    if (M == 0) return VTX_OK;
    Dropout d = make_dropout(p_drop, seed);
    if (dtype == VTX_BF16 && out_f32)
        return gemm_nt_t<bf16_t, float>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, preact, act, alpha, d, bn_parts, bn_shift, bn_strips, (hipStream_t)stream);
    if (dtype == VTX_BF16)
        return gemm_nt_t<bf16_t, bf16_t>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, preact, act, alpha, d, bn_parts, bn_shift, bn_strips, (hipStream_t)stream);
    return gemm_nt_t<float, float>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, preact, act, alpha, d, bn_parts, bn_shift, bn_strips, (hipStream_t)stream);
}

extern "C" int vtx_gemm_tn_acc(int dtype, int M, int N, int K, const void* A, long lda, const void* B,
                               long ldb, float* C, long ldc, float alpha, int split_k, float* workspace,
                               long workspace_floats, void* stream) {
    VTX_CHECK(A && B && C, VTX_ERR_ARG, "gemm_tn_acc: null pointer");
    VTX_CHECK(M > 0 && N > 0 && K >= 0, VTX_ERR_ARG, "gemm_tn_acc: bad shape %dx%dx%d", M, N, K);
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "gemm_tn_acc: bad dtype %d", dtype);
    VTX_CHECK(M % vec == 0 && N % vec == 0 && lda % vec == 0 && ldb % vec == 0, VTX_ERR_SHAPE,
              "gemm_tn_acc: M/N/lda/ldb must be multiples of %d (M=%d N=%d K=%d)", vec, M, N, K);
    VTX_CHECK(aligned16(A) && aligned16(B), VTX_ERR_SHAPE, "gemm_tn_acc: operands must be 16-byte aligned");
    VTX_CHECK(N % 4 == 0 && ldc % 4 == 0 && aligned16(C), VTX_ERR_SHAPE, "gemm_tn_acc: N and ldc must be multiples of 4, C 16-byte aligned");
    if (K == 0) return VTX_OK;
    long wsf = workspace ? workspace_floats : 0;
    workspace = vtx_splitk_region(workspace, wsf, &wsf, (hipStream_t)stream);
    if (split_k <= 0) split_k = vtx_pick_split_k(M, N, K, 4 * vec, wsf);
    else {
        const int nkt = vtx_cdiv(K, 4 * vec);
        if (split_k > nkt) split_k = nkt;
        split_k = vtx_cdiv(nkt, vtx_cdiv(nkt, split_k));
    }
    VTX_CHECK(split_k == 1 || (long)split_k * M * N <= wsf, VTX_ERR_WORKSPACE,
              "gemm_tn_acc: split_k=%d needs %ld workspace floats, got %ld", split_k, (long)split_k * M * N, wsf);
    if (dtype == VTX_BF16)
        return gemm_tn_t<bf16_t>(M, N, K, A, lda, B, ldb, C, ldc, alpha, split_k, workspace, (hipStream_t)stream);
    return gemm_tn_t<float>(M, N, K, A, lda, B, ldb, C, ldc, alpha, split_k, workspace, (hipStream_t)stream);
}

extern "C" int vtx_gemm_nt_bnbwd(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                                 void* C, long ldc, const void* residual, long ldr, VtxBnBwdFusion* f, void* stream) {
    VTX_CHECK(f, VTX_ERR_ARG, "gemm_nt_bnbwd: null fusion descriptor");
    f->strips = 0;
    const bool fuse = dtype == VTX_BF16 && g_vtx_contraction_generation >= 2;
    if (!fuse)      // fp32 parity mode / generation-1 kernel: plain gradient, the caller runs the stand-alone BatchNorm backward
        return vtx_gemm_nt(dtype, M, N, K, A, lda, B, ldb, C, ldc, nullptr, residual, ldr, nullptr, ACT_NONE, 1.f, 0.f, 0, 0,
                           nullptr, nullptr, nullptr, stream);
    VTX_CHECK(A && B && C, VTX_ERR_ARG, "gemm_nt_bnbwd: null pointer");
    VTX_CHECK(M > 0 && N > 0 && K > 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc == N && (!residual || ldr == N),
              VTX_ERR_SHAPE, "gemm_nt_bnbwd: K/lda/ldb must be multiples of 8 and C / residual dense (M=%d N=%d K=%d)", M, N, K);
    VTX_CHECK(aligned16(A) && aligned16(B) && aligned16(C) && (!residual || aligned16(residual)), VTX_ERR_SHAPE,
              "gemm_nt_bnbwd: operands must be 16-byte aligned");
    int rc = vtx_check_bn_bwd("gemm_nt_bnbwd", f, M, N);
    if (rc) return rc;
    EpiStore<bf16_t, STATS_BWD> ep{(bf16_t*)C, ldc, nullptr, (const bf16_t*)residual, ldr, nullptr, ACT_NONE, 1.f,
                                   make_dropout(0.f, 0), M, N};
    vtx_fill_bn_bwd(ep, f, N, f->parts);
    f->strips = launch_auto<bf16_t, PlainKC, PlainKC>(
        [&](auto& a) { a.p = (const bf16_t*)A; a.ld = lda; a.rows = M; a.K = K; },
        [&](auto& b) { b.p = (const bf16_t*)B; b.ld = ldb; b.rows = N; b.K = K; }, ep, M, N, K, 1, (hipStream_t)stream);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
