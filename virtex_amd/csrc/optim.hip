// Fused optimizer tail of the training step over FLAT fp32 buffers (HBM-bound; 2 launches):
//   total_norm = ||grad_scale * g||_2 ;  clip = min(1, max_norm / (total_norm + 1e-6))
//   g' = clip * grad_scale * g + wd[seg] * p ;  m = momentum * m + g' ;  p -= lr[seg]*lr_mult * m
//   every k-th step (do_lookahead): slow += alpha * (p - slow) ; p = slow
//
// Replaces, with identical arithmetic, the reference's
//   torch.nn.utils.clip_grad_norm_ (scripts/pretrain_virtex.py:158),
//   optim.SGD(momentum) over 202 one-tensor param groups (virtex/factories.py:529-545),
//   Lookahead.step (virtex/optim/lookahead.py:82-102)
// i.e. ~1000 tiny foreach launches -> 1 reduction + 1 update kernel.  Parameters, gradients,
// momentum and slow weights share ONE layout (the data-parallel bucket order), described by a
// table of chunks that never straddle a parameter: (offset, length <= 4096, segment id).
#include "vtx_common.h"

namespace {

constexpr int CHUNK = 4096;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long n,
                                                            float* __restrict__ partials) {
    __shared__ float red[4];
    float s = 0.f;
    const long nv = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long i = nv * 4 + threadIdx.x; i < n; i += 256) s += x[i] * x[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(1024) void sumsq_final_kernel(const float* __restrict__ partials, int np,
                                                           float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += 1024) s += partials[i];
    s = block_sum<16>(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ __launch_bounds__(256) void sgd_lookahead_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ slow,
    const long long* __restrict__ chunk_off, const int* __restrict__ chunk_len, const int* __restrict__ chunk_seg,
    const float* __restrict__ seg_lr, const float* __restrict__ seg_wd, float lr_mult, float momentum,
    float grad_scale, const float* __restrict__ sumsq, float max_norm, int do_lookahead, float alpha,
    const float* __restrict__ sched) {
    // sched (optional, device memory): {LR multiplier, Lookahead-sync flag} of THIS step, computed on the device -- a
    // captured hipGraph of the step freezes the by-value arguments (vtx_sgd_lookahead_step_dev)
    if (sched) { lr_mult = sched[0]; do_lookahead = sched[1] != 0.f; }
    const int c = blockIdx.x;
    const long off = chunk_off[c];
    const int len = chunk_len[c], seg = chunk_seg[c];
    const float lr = seg_lr[seg] * lr_mult, wd = seg_wd[seg];
    float gs = grad_scale;
    if (max_norm > 0.f) {
        const float total = sqrtf(sumsq[0]) * grad_scale;
        const float coef = max_norm / (total + 1e-6f);
        gs *= coef < 1.f ? coef : 1.f;
    }
    auto update = [&](float& pv, float gv, float& mv, float& sv) {          // one element; sv is touched on Lookahead steps only
        gv = gv * gs + wd * pv;
        mv = momentum * mv + gv;
        pv -= lr * mv;
        if (do_lookahead) { sv = sv + alpha * (pv - sv); pv = sv; }
    };
    if (((off | len) & 3) == 0) {
        // 16-byte accesses (every chunk of the model qualifies: parameter sizes are multiples of 4): a quarter of the memory
        // instructions of the scalar loop below, same arithmetic per element
        float4* p4 = reinterpret_cast<float4*>(p + off); const float4* g4 = reinterpret_cast<const float4*>(g + off);
        float4* m4 = reinterpret_cast<float4*>(m + off); float4* s4 = reinterpret_cast<float4*>(slow + off);
        for (int i = threadIdx.x; i < len / 4; i += 256) {
            float4 pv = p4[i], mv = m4[i];
            const float4 gv = g4[i];
            float4 sv = do_lookahead ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            update(pv.x, gv.x, mv.x, sv.x); update(pv.y, gv.y, mv.y, sv.y); update(pv.z, gv.z, mv.z, sv.z); update(pv.w, gv.w, mv.w, sv.w);
            m4[i] = mv;
            if (do_lookahead) s4[i] = sv;
            p4[i] = pv;
        }
        return;
    }
    for (int i = threadIdx.x; i < len; i += 256) {
        const long k = off + i;
        float pv = p[k], mv = m[k], sv = do_lookahead ? slow[k] : 0.f;
        update(pv, g[k], mv, sv);
        m[k] = mv;
        if (do_lookahead) slow[k] = sv;
        p[k] = pv;
    }
}

}  // namespace

extern "C" int vtx_optim_chunk_elems(void) { return CHUNK; }

// partials: >= 1024 floats of scratch; out[0] = sum of squares
extern "C" int vtx_sumsq(const float* x, long n, float* partials, float* out, void* stream) {
    VTX_CHECK(x && partials && out && n >= 0, VTX_ERR_ARG, "sumsq: bad args");
    VTX_CHECK(((uintptr_t)x & 15) == 0, VTX_ERR_SHAPE, "sumsq: x must be 16-byte aligned");
    long nb = (n / 4 + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    VTX_KLAUNCH("optimizer_sumsq", 0, 4.0 * n, sumsq_partial_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, x, n, partials);
    VTX_KLAUNCH("optimizer_sumsq_final", 0, 4.0 * nb, sumsq_final_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partials, (int)nb, out);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

static int sgd_lookahead_launch(float* p, const float* g, float* m, float* slow,
                               const long long* chunk_off, const int* chunk_len, const int* chunk_seg,
                               int nchunks, long n_elems, const float* seg_lr, const float* seg_wd, float lr_mult,
                               float momentum, float grad_scale, const float* sumsq, float max_norm,
                               int do_lookahead, float alpha, const float* sched, void* stream) {
    VTX_CHECK(p && g && m && slow && chunk_off && chunk_len && chunk_seg && seg_lr && seg_wd, VTX_ERR_ARG,
              "sgd_lookahead_step: null pointer");
    VTX_CHECK(max_norm <= 0.f || sumsq, VTX_ERR_ARG, "sgd_lookahead_step: clipping needs the sum of squares");
    if (nchunks <= 0) return VTX_OK;
    // algorithmic bytes: the REAL element count (chunks are padded to CHUNK only in the table, never in memory) x (g read, p and m
    // read + written = 5 words; the Lookahead synchronisation adds slow read + written).  With the schedule on the device the
    // host does not know the phase: the 4-of-5 ordinary steps' 5 words are counted (a lower bound, so a fraction above 1 cannot
    // come from the accounting -- round 5 counted CHUNK x nchunks x 7 and reported 1.14 of the HBM peak)
    const double prof_elems = n_elems > 0 ? (double)n_elems : (double)CHUNK * nchunks;
    VTX_KLAUNCH("optimizer_step", 0, 4.0 * prof_elems * ((do_lookahead && !sched) ? 7 : 5), sgd_lookahead_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, p, g, m, slow, chunk_off,
                       chunk_len, chunk_seg, seg_lr, seg_wd, lr_mult, momentum, grad_scale, sumsq, max_norm,
                       do_lookahead, alpha, sched);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_sgd_lookahead_step(float* p, const float* g, float* m, float* slow,
                                      const long long* chunk_off, const int* chunk_len, const int* chunk_seg,
                                      int nchunks, long n_elems, const float* seg_lr, const float* seg_wd, float lr_mult,
                                      float momentum, float grad_scale, const float* sumsq, float max_norm,
                                      int do_lookahead, float alpha, void* stream) {
    return sgd_lookahead_launch(p, g, m, slow, chunk_off, chunk_len, chunk_seg, nchunks, n_elems, seg_lr, seg_wd, lr_mult, momentum, grad_scale,
                                sumsq, max_norm, do_lookahead, alpha, nullptr, stream);
}

// The same step with the two per-step scalars read from DEVICE memory: sched[0] = LR multiplier, sched[1] = 1.0f when this
// step ends with the Lookahead synchronisation (every k-th), else 0.0f -- what a captured hipGraph of the training step
// needs (virtex_amd/graph.py keeps the step counter, the schedule and this pair on the device).
extern "C" int vtx_sgd_lookahead_step_dev(float* p, const float* g, float* m, float* slow,
                                          const long long* chunk_off, const int* chunk_len, const int* chunk_seg,
                                          int nchunks, long n_elems, const float* seg_lr, const float* seg_wd, const float* sched,
                                          float momentum, float grad_scale, const float* sumsq, float max_norm,
                                          float alpha, void* stream) {
    VTX_CHECK(sched, VTX_ERR_ARG, "sgd_lookahead_step_dev: null schedule pointer");
    return sgd_lookahead_launch(p, g, m, slow, chunk_off, chunk_len, chunk_seg, nchunks, n_elems, seg_lr, seg_wd, 0.f, momentum, grad_scale,
                                sumsq, max_norm, 1, alpha, sched, stream);
}
