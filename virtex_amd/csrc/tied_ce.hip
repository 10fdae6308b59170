// Tied output projection + softmax cross-entropy WITHOUT materialised logits (training path).
//
//   loss = mean_{r : target[r] != ignore} ( logsumexp_c(h[r].W[c] + b[c]) - (h[r].W[target[r]] + b[target[r]]) )
//
// Replaces aten::linear (the tied output layer, /root/reference/virtex/modules/textual_heads.py:199-200,277) followed by
// aten::log_softmax + nll_loss of nn.CrossEntropyLoss(ignore_index) (/root/reference/virtex/models/captioning.py:69,
// 111-114,127-132) and their backward.  The reference materialises (B,T,V) fp32 logits (307 MB per head at B = 256);
// here they exist only as MFMA accumulators:
//   forward : the GEMM's epilogue (EpiRowLse, gemm_kernel.h) emits per (row, column group) max / sum-exp partials and the
//             target logit; ce_lse_combine folds them into lse[r] and the per-row loss; ce_mean gives {loss, count};
//   backward: the SAME GEMM is recomputed with an epilogue that turns each accumulator into
//             d = g/count * (exp(v - lse[r]) - [c == target[r]]) in the compute dtype -- the one [R][V] tensor of the path,
//             bf16, consumed by the two gradient GEMMs (dh = d.W, dW = d^T.h).  Keeping d out of HBM as well would need
//             [BM][H] fp32 accumulators per block (H = 1024: 4x the register file), so it is written once instead.
#include "gemm_kernel.h"

using namespace vtxg;

namespace {

// Folds the per-column-group partials of 32 rows per block: thread (g, r) walks groups g, g+8, ... of row r with the
// online max / sum-exp merge (one pass, 128-byte coalesced across the rows), LDS folds the 8 walkers of a row.
__global__ __launch_bounds__(256) void ce_lse_combine_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                             const float* __restrict__ tgt_logit,
                                                             const long long* __restrict__ targets, float* __restrict__ lse,
                                                             float* __restrict__ row_loss, int R, int P, int V, int ignore_index) {
    __shared__ float sm[8][32], ss[8][32];
    const int rl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int r = blockIdx.x * 32 + rl;
    float m = -INFINITY, s = 0.f;
    if (r < R) {
        for (int p = g; p < P; p += 8) {
            const float pm = pmax[(size_t)p * R + r], ps = psum[(size_t)p * R + r];
            const float nm = fmaxf(m, pm);
            if (nm > -INFINITY) { s = s * __expf(m - nm) + ps * __expf(pm - nm); m = nm; }
        }
    }
    sm[g][rl] = m; ss[g][rl] = s;
    __syncthreads();
    if (g == 0 && r < R) {
        float mm = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) mm = fmaxf(mm, sm[k][rl]);
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (sm[k][rl] > -INFINITY) tot += ss[k][rl] * __expf(sm[k][rl] - mm);
        const float l = mm + __logf(tot);
        lse[r] = l;
        const long long t = targets[r];
        row_loss[r] = (t == ignore_index || t < 0 || t >= V) ? 0.f : l - tgt_logit[r];
    }
}

// single block: out[0] = sum(row_loss over valid rows) / count, out[1] = count   (0/0 = NaN like torch)
__global__ __launch_bounds__(1024) void ce_mean_kernel(const float* __restrict__ row_loss, const long long* __restrict__ targets,
                                                       float* __restrict__ out, int R, int V, int ignore_index) {
    __shared__ float red[16];
    float s = 0.f, n = 0.f;
    for (int r = threadIdx.x; r < R; r += 1024) {
        const long long t = targets[r];
        if (t != ignore_index && t >= 0 && t < V) { s += row_loss[r]; n += 1.f; }
    }
    s = block_sum<16>(s, red);
    n = block_sum<16>(n, red);
    if (threadIdx.x == 0) { out[0] = s / n; out[1] = n; }
}

template <class T>
int tied_ce_fwd_t(int R, int V, int H, const void* hidden, long ldh, const void* weight, long ldw, const float* bias,
                  const long long* targets, int ignore_index, float* pmax, float* psum, long part_cap, float* tgt_logit, float* lse,
                  float* row_loss, float* loss_and_count, hipStream_t st) {
    EpiRowLse ep{bias, 1.f, R, V, targets, tgt_logit, pmax, psum};
    launch_auto<T, PlainKC, PlainKC>(
        [&](auto& a) { a.p = (const T*)hidden; a.ld = ldh; a.rows = R; a.K = H; },
        [&](auto& b) { b.p = (const T*)weight; b.ld = ldw; b.rows = V; b.K = H; }, ep, R, V, H, 1, st);
    const int P = g_vtx_last_colgroups;
    VTX_CHECK((long)P * R <= part_cap, VTX_ERR_WORKSPACE, "tied_ce_fwd: %d column groups x %d rows exceed the partial buffers (%ld)", P, R, part_cap);
    VTX_KLAUNCH("tied_ce_fwd", 0, 8.0 * P * R, ce_lse_combine_kernel, dim3(vtx_cdiv(R, 32)), dim3(256), 0, st, pmax, psum, tgt_logit, targets,
                lse, row_loss, R, P, V, ignore_index);
    VTX_KLAUNCH("cross_entropy_reduce", 0, 12.0 * R, ce_mean_kernel, dim3(1), dim3(1024), 0, st, row_loss, targets, loss_and_count, R, V, ignore_index);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

template <class T>
int tied_ce_bwd_t(int R, int V, int H, const void* hidden, long ldh, const void* weight, long ldw, const float* bias,
                  const long long* targets, int ignore_index, const float* lse, const float* loss_and_count, const float* grad_out,
                  void* d, hipStream_t st) {
    EpiStore<T> ep{(T*)d, V, bias, nullptr, 0, nullptr, ACT_SOFTMAX_GRAD, 1.f, make_dropout(0.f, 0), R, V};
    ep.ce_lse = lse; ep.ce_targets = targets; ep.ce_gout = grad_out; ep.ce_lc = loss_and_count; ep.ce_ignore = ignore_index;
    launch_auto<T, PlainKC, PlainKC>(
        [&](auto& a) { a.p = (const T*)hidden; a.ld = ldh; a.rows = R; a.K = H; },
        [&](auto& b) { b.p = (const T*)weight; b.ld = ldw; b.rows = V; b.K = H; }, ep, R, V, H, 1, st);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// a wave tile is at least 32 columns wide: that many column groups at most
extern "C" long vtx_tied_ce_partial_floats(int R, int V) { return (long)(vtx_cdiv(V, 32) + 2) * R; }

extern "C" int vtx_tied_ce_fwd(int dtype, int R, int V, int H, const void* hidden, long ldh, const void* weight, long ldw,
                               const float* bias, const long long* targets, int ignore_index, float* pmax, float* psum,
                               long partial_floats, float* tgt_logit, float* lse, float* row_loss, float* loss_and_count, void* stream) {
    VTX_CHECK(hidden && weight && targets && pmax && psum && tgt_logit && lse && row_loss && loss_and_count, VTX_ERR_ARG, "tied_ce_fwd: null pointer");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "tied_ce_fwd: bad dtype %d", dtype);
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(R > 0 && V > 0 && H > 0 && H % vec == 0 && ldh % vec == 0 && ldw % vec == 0 && aligned16(hidden) && aligned16(weight), VTX_ERR_SHAPE,
              "tied_ce_fwd: H / row strides must be multiples of %d, operands 16-byte aligned", vec);
    if (dtype == VTX_BF16)
        return tied_ce_fwd_t<bf16_t>(R, V, H, hidden, ldh, weight, ldw, bias, targets, ignore_index, pmax, psum, partial_floats, tgt_logit, lse, row_loss, loss_and_count, (hipStream_t)stream);
    return tied_ce_fwd_t<float>(R, V, H, hidden, ldh, weight, ldw, bias, targets, ignore_index, pmax, psum, partial_floats, tgt_logit, lse, row_loss, loss_and_count, (hipStream_t)stream);
}

extern "C" int vtx_tied_ce_bwd(int dtype, int R, int V, int H, const void* hidden, long ldh, const void* weight, long ldw,
                               const float* bias, const long long* targets, int ignore_index, const float* lse,
                               const float* loss_and_count, const float* grad_out, void* dlogits, void* stream) {
    VTX_CHECK(hidden && weight && targets && lse && loss_and_count && grad_out && dlogits, VTX_ERR_ARG, "tied_ce_bwd: null pointer");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "tied_ce_bwd: bad dtype %d", dtype);
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(R > 0 && V > 0 && V % 4 == 0 && H > 0 && H % vec == 0 && ldh % vec == 0 && ldw % vec == 0 && aligned16(hidden) && aligned16(weight) && aligned16(dlogits),
              VTX_ERR_SHAPE, "tied_ce_bwd: V must be a multiple of 4, H / row strides of %d, operands 16-byte aligned", vec);
    if (dtype == VTX_BF16)
        return tied_ce_bwd_t<bf16_t>(R, V, H, hidden, ldh, weight, ldw, bias, targets, ignore_index, lse, loss_and_count, grad_out, dlogits, (hipStream_t)stream);
    return tied_ce_bwd_t<float>(R, V, H, hidden, ldh, weight, ldw, bias, targets, ignore_index, lse, loss_and_count, grad_out, dlogits, (hipStream_t)stream);
}
