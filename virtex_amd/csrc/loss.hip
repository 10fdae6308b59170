// Softmax cross-entropy over the vocabulary (fp32), forward and backward (HBM-bound).
//
//   loss = mean_{r : target[r] != ignore} ( logsumexp(logits[r]) - logits[r][target[r]] )
//
// Replaces aten::log_softmax + nll_loss (+ backward) of nn.CrossEntropyLoss(ignore_index=
// padding_idx) at /root/reference/virtex/models/captioning.py:69,111-114,127-132.  The
// caller passes the row -> (b,t) mapping implicitly: logits row r = b*T + t predicts token
// t+1, rows with t == T-1 are skipped via target == ignore (see virtex_amd/modules).
// Forward keeps one logsumexp per row; backward re-reads the logits once and writes
// d(logits) in the compute dtype (bf16 operand of the two gradient GEMMs).
#include "vtx_common.h"

namespace {

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, long ld,
                                                     const long long* __restrict__ targets,
                                                     float* __restrict__ lse, float* __restrict__ row_loss,
                                                     int R, int V, int ignore_index) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const long long tgt = targets[r];
    if (tgt == ignore_index || tgt < 0 || tgt >= V) {  // block-uniform
        if (threadIdx.x == 0) { lse[r] = 0.f; row_loss[r] = 0.f; }
        return;
    }
    const float* row = logits + (long)r * ld;
    float m = -INFINITY;
    for (int c = threadIdx.x * 4; c < V; c += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(row + c);
        m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    m = block_max<4>(m, red);
    float s = 0.f;
    for (int c = threadIdx.x * 4; c < V; c += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(row + c);
        s += __expf(v.x - m) + __expf(v.y - m) + __expf(v.z - m) + __expf(v.w - m);
    }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) {
        const float l = m + __logf(s);
        lse[r] = l;
        row_loss[r] = l - row[tgt];
    }
}

// single block: loss = sum(row_loss) / count ; out[0] = loss, out[1] = count
__global__ __launch_bounds__(1024) void ce_reduce_kernel(const float* __restrict__ row_loss,
                                                         const long long* __restrict__ targets,
                                                         float* __restrict__ out, int R, int V, int ignore_index) {
    __shared__ float red[16];
    float s = 0.f, n = 0.f;
    for (int r = threadIdx.x; r < R; r += 1024) {
        const long long t = targets[r];
        if (t != ignore_index && t >= 0 && t < V) { s += row_loss[r]; n += 1.f; }
    }
    s = block_sum<16>(s, red);
    n = block_sum<16>(n, red);
    if (threadIdx.x == 0) { out[0] = s / n; out[1] = n; }   // 0/0 = NaN like torch
}

// dlogits[r][c] = g/count * (exp(logit - lse) - [c == target])   (0 for ignored rows)
template <class T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long ld,
                                                     const long long* __restrict__ targets,
                                                     const float* __restrict__ lse,
                                                     const float* __restrict__ loss_and_count,
                                                     const float* __restrict__ grad_out, T* __restrict__ dlogits,
                                                     long ldd, int R, int V, int ignore_index) {
    const int r = blockIdx.x;
    const long long tgt = targets[r];
    const bool ignored = (tgt == ignore_index || tgt < 0 || tgt >= V);
    const float g = ignored ? 0.f : grad_out[0] / loss_and_count[1];
    const float l = lse[r];
    const float* row = logits + (long)r * ld;
    T* drow = dlogits + (long)r * ldd;
    for (int c = threadIdx.x * 4; c < V; c += 1024) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (!ignored) {
            const float4 v = *reinterpret_cast<const float4*>(row + c);
            o[0] = g * __expf(v.x - l); o[1] = g * __expf(v.y - l);
            o[2] = g * __expf(v.z - l); o[3] = g * __expf(v.w - l);
            const int k = (int)tgt - c;
            if (k >= 0 && k < 4) o[k] -= g;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) Elem<T>::st(drow + c + j, o[j]);
    }
}

}  // namespace

extern "C" int vtx_cross_entropy_fwd(const float* logits, long ld, const long long* targets, float* lse,
                                     float* row_loss, float* loss_and_count, int R, int V, int ignore_index,
                                     void* stream) {
    VTX_CHECK(logits && targets && lse && row_loss && loss_and_count, VTX_ERR_ARG, "cross_entropy_fwd: null pointer");
    VTX_CHECK(R > 0 && V > 0 && V % 4 == 0 && ld % 4 == 0, VTX_ERR_SHAPE, "cross_entropy_fwd: V and ld must be multiples of 4");
    hipStream_t st = (hipStream_t)stream;
    VTX_KLAUNCH("cross_entropy_fwd", 0, 8.0 * R * V, ce_fwd_kernel, dim3(R), dim3(256), 0, st, logits, ld, targets, lse, row_loss, R, V, ignore_index);
    VTX_KLAUNCH("cross_entropy_reduce", 0, 12.0 * R, ce_reduce_kernel, dim3(1), dim3(1024), 0, st, row_loss, targets, loss_and_count, R, V, ignore_index);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_cross_entropy_bwd(int dtype, const float* logits, long ld, const long long* targets,
                                     const float* lse, const float* loss_and_count, const float* grad_out,
                                     void* dlogits, long ldd, int R, int V, int ignore_index, void* stream) {
    VTX_CHECK(logits && targets && lse && loss_and_count && grad_out && dlogits, VTX_ERR_ARG, "cross_entropy_bwd: null pointer");
    VTX_CHECK(R > 0 && V > 0 && V % 4 == 0 && ld % 4 == 0, VTX_ERR_SHAPE, "cross_entropy_bwd: V and ld must be multiples of 4");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("cross_entropy_bwd", 0, 6.0 * R * V, (ce_bwd_kernel<bf16_t>), dim3(R), dim3(256), 0, st, logits, ld, targets, lse, loss_and_count,
                           grad_out, (bf16_t*)dlogits, ldd, R, V, ignore_index);
    else if (dtype == VTX_F32)
        VTX_KLAUNCH("cross_entropy_bwd", 0, 8.0 * R * V, (ce_bwd_kernel<float>), dim3(R), dim3(256), 0, st, logits, ld, targets, lse, loss_and_count,
                           grad_out, (float*)dlogits, ldd, R, V, ignore_index);
    else VTX_CHECK(false, VTX_ERR_DTYPE, "cross_entropy_bwd: bad dtype");
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
