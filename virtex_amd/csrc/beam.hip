// One beam-search step on the device: log-softmax of the next-token logits, the no-immediate-repetition penalty,
// "a finished beam may only emit EOS again", the best `per_node` continuations of every beam, then the best `beam` of each
// image's beams_in * per_node candidates.  Replaces the aten::log_softmax / scatter_ / where / topk / gather chain (and the
// per-row Python loop) of /root/reference/virtex/utils/beam_search.py:115-228.  Ties go to the lowest index.
#include "vtx_common.h"

namespace {

constexpr float REPEAT_PENALTY = -10000.0f;      // beam_search.py:157
constexpr int MAX_PER_NODE = 16;

struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }

// block-wide arg max (ties -> lowest index); `red` = 4 entries of LDS
__device__ __forceinline__ ArgMax block_argmax(ArgMax x, ArgMax* red) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        ArgMax o{__shfl_xor(x.v, m, 64), __shfl_xor(x.i, m, 64)};
        x = better(x, o);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = x;
    __syncthreads();
    ArgMax r = red[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) r = better(r, red[k]);
    return r;
}

// one block (256 threads) per row: cand_lp/cand_tok[row][0..P) = the P best adjusted log-probabilities, best first
__global__ __launch_bounds__(256) void beam_row_kernel(const float* __restrict__ logits, long ld, const long long* __restrict__ last,
                                                       float* __restrict__ cand_lp, long long* __restrict__ cand_tok, int V, int eos, int P) {
    __shared__ float fred[4];
    __shared__ ArgMax ared[4];
    __shared__ int chosen[MAX_PER_NODE];
    const int row = blockIdx.x;
    const float* x = logits + (long)row * ld;
    const long long prev = last ? last[row] : -1;
    if (last && prev == eos) {                     // finished beam: EOS with log-probability 0, everything else impossible
        if (threadIdx.x < P) {
            int tok = threadIdx.x == 0 ? eos : (threadIdx.x - 1 < eos ? threadIdx.x - 1 : threadIdx.x);     // lowest indices != eos
            cand_lp[(long)row * P + threadIdx.x] = threadIdx.x == 0 ? 0.f : -INFINITY;
            cand_tok[(long)row * P + threadIdx.x] = tok;
        }
        return;
    }
    float m = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) m = fmaxf(m, x[c]);
    m = block_max<4>(m, fred);
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += __expf(x[c] - m);
    s = block_sum<4>(s, fred);
    const float lse = m + __logf(s);
    for (int p = 0; p < P; ++p) {
        ArgMax best{-INFINITY, 0x7fffffff};
        for (int c = threadIdx.x; c < V; c += 256) {
            bool taken = false;
            for (int q = 0; q < p; ++q) taken |= chosen[q] == c;
            if (taken) continue;
            const float v = (long long)c == prev ? REPEAT_PENALTY : x[c] - lse;
            best = better(best, ArgMax{v, c});
        }
        best = block_argmax(best, ared);
        if (best.i == 0x7fffffff) {                // fewer than P tokens (V < P): pad with impossible candidates
            best.v = -INFINITY; best.i = 0;
        }
        if (threadIdx.x == 0) {
            chosen[p] = best.i;
            cand_lp[(long)row * P + p] = best.v;
            cand_tok[(long)row * P + p] = best.i;
        }
        __syncthreads();
    }
}

// one block (64 threads) per image: the `beam` best of its beams_in * P candidates by cumulative score
__global__ __launch_bounds__(64) void beam_merge_kernel(const float* __restrict__ cand_lp, const long long* __restrict__ cand_tok,
                                                        const float* __restrict__ score_in, float* __restrict__ score_out,
                                                        long long* __restrict__ parent_out, long long* __restrict__ token_out,
                                                        int beams_in, int P, int beam) {
    const int img = blockIdx.x, n = beams_in * P, c = threadIdx.x;
    float total = -INFINITY;
    bool live = c < n;
    if (live) {
        const int b = c / P;
        total = cand_lp[(long)(img * beams_in + b) * P + (c % P)] + (score_in ? score_in[img * beams_in + b] : 0.f);
    }
    for (int k = 0; k < beam; ++k) {
        ArgMax x{live ? total : -INFINITY, live ? c : 0x7fffffff};
        if (!live) x.v = -INFINITY;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            ArgMax o{__shfl_xor(x.v, m, 64), __shfl_xor(x.i, m, 64)};
            x = better(x, o);
        }
        // candidates that are all -inf (or exhausted): x.i is the lowest live index, or none
        int pick = x.i;
        if (pick == 0x7fffffff) pick = 0;
        if (c == 0) {
            const int b = pick / P;
            score_out[img * beam + k] = x.i == 0x7fffffff ? -INFINITY : x.v;
            parent_out[img * beam + k] = b;
            token_out[img * beam + k] = cand_tok[(long)(img * beams_in + b) * P + (pick % P)];
        }
        if (c == pick) live = false;
    }
}

}  // namespace

extern "C" int vtx_beam_step(const float* logits, long ld, const long long* last, const float* score_in, int images,
                             int beams_in, int V, int eos, int per_node, int beam, float* cand_lp, long long* cand_tok,
                             float* score_out, long long* parent_out, long long* token_out, void* stream) {
    VTX_CHECK(logits && cand_lp && cand_tok && score_out && parent_out && token_out, VTX_ERR_ARG, "beam_step: null pointer");
    VTX_CHECK(images > 0 && beams_in > 0 && V > 0 && per_node > 0 && beam > 0 && ld >= V, VTX_ERR_ARG, "beam_step: bad shape");
    VTX_CHECK(per_node <= MAX_PER_NODE && beams_in * per_node <= 64 && beam <= beams_in * per_node, VTX_ERR_SHAPE,
              "beam_step: per_node <= %d, beams*per_node <= 64 and beam <= beams*per_node (got %d beams x %d, beam %d)",
              MAX_PER_NODE, beams_in, per_node, beam);
    hipStream_t st = (hipStream_t)stream;
    const int rows = images * beams_in;
    VTX_KLAUNCH("beam_step", 0, 4.0 * rows * V * (2 + per_node), beam_row_kernel, dim3(rows), dim3(256), 0, st, logits, ld, last, cand_lp, cand_tok, V, eos, per_node);
    VTX_KLAUNCH("beam_step", 0, 12.0 * rows * per_node, beam_merge_kernel, dim3(images), dim3(64), 0, st, cand_lp, cand_tok, score_in, score_out, parent_out,
                token_out, beams_in, per_node, beam);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
