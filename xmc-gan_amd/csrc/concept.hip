// Per-sample "concept algebra" of the sentence-conditioned attention-modulation block (InConceptBlock, df_concept_gan.py:213-253
// with CondConceptSampler 273-302, ConceptReasoner 313-326 and the gamma/beta grouped MLPs 178-200), forward and backward.
//
// Everything here lives on [16 concepts x <= 260] numbers per sample.  The reference (and round 1 of this build) runs it as ~80
// tiny framework launches per sampler (einsum, linear, tanh, matmul, relu, cat, leaky_relu, group_norm and their backward
// nodes): ~1 900 launches of 4-5 us per iteration at 128 px, a quarter of the iteration.  Here a stage is one or two launches:
//
//   xmc_concept_query_fwd/bwd : q[g,:] = GroupNorm_4( Wq[g] (4 x nef) . sent )                        (273-286; gn1)
//   xmc_concept_gquery_fwd/bwd: q[g,:] = GroupNorm_4( Wq[g] (4 x 8) . avgpool(x)[g] )                   (555-569: the self-attention block)
//   xmc_concept_head_fwd/bwd  : v = Wv[g] ctx[g];  adj = tanh(v We^T);  r = relu(v + adj v);                (291-302, 313-326)
//                               [self-attention block, 471-478: s = Ws sent; att = softmax_g <s, r[g]>; r[g] <- att[g] r[g]]
//                               for t in {gamma, beta}:  a = W1_t[g] [sent ; r[g]] + b1_t;  out_t = W2_t[g] lrelu(a) + b2_t   (238-253)
//
// The only part with any arithmetic in it is the sentence vector against the 64 (query) / 256 (MLP layer 1) weight rows of nef
// columns.  Forward: one workgroup per sample, lanes ALONG the nef columns (coalesced 16-byte weight loads, the L2-resident
// matrix is read once per sample), 64 rows' partial sums per wave reduced by a 63-shuffle transpose-reduce that leaves row l's
// sum in lane l.  Backward: the per-sample kernel writes d(pre-activation) [B, rows]; a second launch forms the two batch
// products  dW[r, i] = sum_b d[b, r] sent[b, i]  and  dsent[b, i] = sum_r d[b, r] W[r, i]  with lanes along i and one writer per
// element (no atomics).  The remaining parameter gradients (a few hundred numbers) are accumulated with f32 atomics into
// buffers the caller zeroed.
#include "common.h"

namespace {

constexpr int CARD = 16, SD = 4, PWD = 8;     // concepts, state dim p', bottleneck width p (df_concept_gan.py:110,118)
constexpr int HID = CARD * 2 * SD;            // 128 layer-1 units per MLP

__device__ __forceinline__ float lrelu02(float v) { return v > 0.f ? v : 0.2f * v; }

template <int S>
__device__ __forceinline__ void tr_step(float (&p)[64], int lane) {
    const bool hi = (lane & S) != 0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const float send = hi ? p[j] : p[j + S];
        const float keep = hi ? p[j + S] : p[j];
        p[j] = keep + __shfl_xor(send, S, 64);
    }
}

// one wave: returns in lane l  sum_i W[l*ld + i] * x[i]  (i < E; x in shared memory)
__device__ __forceinline__ float rows_dot64(const float* __restrict__ W, int ld, const float* x, int E, int lane) {
    float p[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) p[j] = 0.f;
    if (((E | ld) & 3) == 0) {
        for (int i0 = lane * 4; i0 < E; i0 += 256) {
            const float4 s = *reinterpret_cast<const float4*>(x + i0);
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                const float4 w = *reinterpret_cast<const float4*>(W + (size_t)j * ld + i0);
                p[j] += w.x * s.x + w.y * s.y + w.z * s.z + w.w * s.w;
            }
        }
    } else {
        for (int i0 = lane; i0 < E; i0 += 64) {
            const float s = x[i0];
#pragma unroll
            for (int j = 0; j < 64; ++j) p[j] += W[(size_t)j * ld + i0] * s;
        }
    }
    tr_step<32>(p, lane); tr_step<16>(p, lane); tr_step<8>(p, lane);
    tr_step<4>(p, lane);  tr_step<2>(p, lane);  tr_step<1>(p, lane);
    return p[0];
}

// ------------------------------------------------------------------------------------------------ batch products of a backward
// D [B, R] (d pre-activation), X [B, C] (the sentence vectors); rows [k*Rp, (k+1)*Rp) of D belong to weight W[k] [Rp, ldw].
// blocks [0, R/2): dW rows 2*blk, 2*blk+1 (written: columns [0, C));   blocks [R/2, R/2+B): dX[b, :] (written)
constexpr int QMAX = 64;                // weights per batch product; sampler stages per hoisted query launch: QMAX / 2
struct OuterArgs {
    const float* D; const float* X; const float* W[QMAX]; float* dW[QMAX]; float* dX;
    int B, R, Rp, C, ldw, acc_dx;       // acc_dx: dX += instead of =
    int kchunk;                         // 0: one workgroup per sample walks all R rows for dX;  > 0: one per (sample, kchunk weights), added
                                        // atomically into a dX the caller zeroed (the all-stages launches: 6 144 rows per sample)
};
__global__ __launch_bounds__(256) void concept_outer_kernel(OuterArgs a) {
    const int blk = blockIdx.x;
    if (blk < a.R / 2) {
        const int r0 = blk * 2, k = r0 / a.Rp, rr = r0 - k * a.Rp;
        for (int c = threadIdx.x; c < a.C; c += 256) {
            float s0 = 0.f, s1 = 0.f;
            for (int b = 0; b < a.B; ++b) {
                const float x = a.X[(size_t)b * a.C + c];
                s0 += a.D[(size_t)b * a.R + r0] * x;
                s1 += a.D[(size_t)b * a.R + r0 + 1] * x;
            }
            a.dW[k][(size_t)rr * a.ldw + c] = s0;
            a.dW[k][(size_t)(rr + 1) * a.ldw + c] = s1;
        }
    } else if (a.kchunk > 0) {
        const int nch = (a.R / a.Rp + a.kchunk - 1) / a.kchunk;
        const int b = (blk - a.R / 2) / nch, k0 = ((blk - a.R / 2) % nch) * a.kchunk;
        const int k1 = k0 + a.kchunk < a.R / a.Rp ? k0 + a.kchunk : a.R / a.Rp;
        for (int c = threadIdx.x; c < a.C; c += 256) {
            float s = 0.f;
            for (int k = k0; k < k1; ++k) {
                const float* w = a.W[k];
                const float* dd = a.D + (size_t)b * a.R + k * a.Rp;
#pragma unroll 8
                for (int r = 0; r < a.Rp; ++r) s += dd[r] * w[(size_t)r * a.ldw + c];
            }
            atomicAdd(&a.dX[(size_t)b * a.C + c], s);
        }
    } else {
        const int b = blk - a.R / 2;
        for (int c = threadIdx.x; c < a.C; c += 256) {
            float s = 0.f;
            for (int k = 0; k * a.Rp < a.R; ++k) {
                const float* w = a.W[k];
                const float* dd = a.D + (size_t)b * a.R + k * a.Rp;
#pragma unroll 8
                for (int r = 0; r < a.Rp; ++r) s += dd[r] * w[(size_t)r * a.ldw + c];
            }
            if (a.acc_dx) s += a.dX[(size_t)b * a.C + c];
            a.dX[(size_t)b * a.C + c] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------ query
// sent [B, E]; Wq [CARD*SD, E] (grouped 1x1: row g*SD+o is group g's output o over the WHOLE sentence vector: the reference
// feeds every group the same sentence, 276-280); gnw/gnb [CARD*SD] or NULL (GEN.NORMALIZE False); q, qraw [B, CARD*SD]
__global__ __launch_bounds__(64) void concept_query_fwd_kernel(const float* __restrict__ sent, const float* __restrict__ Wq,
                                                              const float* __restrict__ gnw, const float* __restrict__ gnb,
                                                              float* __restrict__ q, float* __restrict__ qraw, int E, float eps) {
    __shared__ __attribute__((aligned(16))) float s_sent[1024];
    const int b = blockIdx.x, c = threadIdx.x;        // c = g*SD + o
    for (int i = c; i < E; i += 64) s_sent[i] = sent[(size_t)b * E + i];
    __syncthreads();
    float x = rows_dot64(Wq, E, s_sent, E, c);
    qraw[(size_t)b * 64 + c] = x;
    if (gnw) {                                        // GroupNorm over the SD values of a concept
        float m = x + xmc_xor1(x); m += xmc_xor2(m); m *= 0.25f;
        const float dx = x - m;
        float v = dx * dx; v += xmc_xor1(v); v += xmc_xor2(v); v *= 0.25f;
        x = dx * rsqrtf(v + eps) * gnw[c] + gnb[c];
    }
    q[(size_t)b * 64 + c] = x;
}

// dq [B,64] -> dx [B,64] (d of the grouped 1x1's output, for concept_outer_kernel); dgnw/dgnb [64] atomically accumulated
__global__ __launch_bounds__(64) void concept_query_bwd_kernel(const float* __restrict__ gnw, const float* __restrict__ qraw,
                                                              const float* __restrict__ dq, float* __restrict__ dx,
                                                              float* __restrict__ dgnw, float* __restrict__ dgnb, float eps) {
    const int b = blockIdx.x, c = threadIdx.x;
    const float x = qraw[(size_t)b * 64 + c];
    float g = dq[(size_t)b * 64 + c];
    if (gnw) {
        float m = x + xmc_xor1(x); m += xmc_xor2(m); m *= 0.25f;
        const float dxm = x - m;
        float v = dxm * dxm; v += xmc_xor1(v); v += xmc_xor2(v); v *= 0.25f;
        const float rstd = rsqrtf(v + eps), xh = dxm * rstd;
        atomicAdd(&dgnw[c], g * xh);
        atomicAdd(&dgnb[c], g);
        const float gh = g * gnw[c];
        float s1 = gh + xmc_xor1(gh); s1 += xmc_xor2(s1); s1 *= 0.25f;
        float s2 = gh * xh; s2 += xmc_xor1(s2); s2 += xmc_xor2(s2); s2 *= 0.25f;
        g = rstd * (gh - s1 - xh * s2);
    }
    dx[(size_t)b * 64 + c] = g;
}

// The sentence queries of EVERY sampler stage of a generator in one launch (they depend on nothing but the sentence vector:
// xmc_concept_query_fwd_multi): blockIdx.y = stage s, weights through a pointer table in the kernel arguments, q / qraw [S][B][64].
struct QueryTab { const float* Wq[32]; const float* gnw[32]; const float* gnb[32]; };
__global__ __launch_bounds__(64) void concept_query_fwd_multi_kernel(const float* __restrict__ sent, const QueryTab T, float* __restrict__ q,
                                                                    float* __restrict__ qraw, int B, int E, float eps) {
    __shared__ __attribute__((aligned(16))) float s_sent[1024];
    const int b = blockIdx.x, s = blockIdx.y, c = threadIdx.x;
    for (int i = c; i < E; i += 64) s_sent[i] = sent[(size_t)b * E + i];
    __syncthreads();
    float x = rows_dot64(T.Wq[s], E, s_sent, E, c);
    const size_t o = ((size_t)s * B + b) * 64 + c;
    qraw[o] = x;
    if (T.gnw[s]) {
        float m = x + xmc_xor1(x); m += xmc_xor2(m); m *= 0.25f;
        const float dx = x - m;
        float v = dx * dx; v += xmc_xor1(v); v += xmc_xor2(v); v *= 0.25f;
        x = dx * rsqrtf(v + eps) * T.gnw[s][c] + T.gnb[s][c];
    }
    q[o] = x;
}
// dq [S][B][64] -> dx [B][S*64] (row layout of the ONE batch product that follows); dgn [S][2][64] atomically accumulated
__global__ __launch_bounds__(64) void concept_query_bwd_multi_kernel(const QueryTab T, const float* __restrict__ qraw, const float* __restrict__ dq,
                                                                    float* __restrict__ dx, float* __restrict__ dgn, int B, int S, float eps) {
    const int b = blockIdx.x, s = blockIdx.y, c = threadIdx.x;
    const size_t o = ((size_t)s * B + b) * 64 + c;
    const float x = qraw[o];
    float g = dq[o];
    if (T.gnw[s]) {
        float m = x + xmc_xor1(x); m += xmc_xor2(m); m *= 0.25f;
        const float dxm = x - m;
        float v = dxm * dxm; v += xmc_xor1(v); v += xmc_xor2(v); v *= 0.25f;
        const float rstd = rsqrtf(v + eps), xh = dxm * rstd;
        atomicAdd(&dgn[(s * 2 + 0) * 64 + c], g * xh);
        atomicAdd(&dgn[(s * 2 + 1) * 64 + c], g);
        const float gh = g * T.gnw[s][c];
        float s1 = gh + xmc_xor1(gh); s1 += xmc_xor2(s1); s1 *= 0.25f;
        float s2 = gh * xh; s2 += xmc_xor1(s2); s2 += xmc_xor2(s2); s2 *= 0.25f;
        g = rstd * (gh - s1 - xh * s2);
    }
    dx[(size_t)b * S * 64 + s * 64 + c] = g;
}

// query of the self-attention sampler: q0 [B, CARD*PWD] (global average of x), Wq [CARD*SD, PWD] grouped 1x1 (555-569)
__global__ __launch_bounds__(64) void concept_gquery_fwd_kernel(const float* __restrict__ q0, const float* __restrict__ Wq,
                                                               const float* __restrict__ gnw, const float* __restrict__ gnb,
                                                               float* __restrict__ q, float* __restrict__ qraw, float eps) {
    const int b = blockIdx.x, c = threadIdx.x, g = c >> 2;
    float x = 0.f;
#pragma unroll
    for (int i = 0; i < PWD; ++i) x += Wq[c * PWD + i] * q0[(size_t)b * CARD * PWD + g * PWD + i];
    qraw[(size_t)b * 64 + c] = x;
    if (gnw) {
        float m = x + xmc_xor1(x); m += xmc_xor2(m); m *= 0.25f;
        const float dx = x - m;
        float v = dx * dx; v += xmc_xor1(v); v += xmc_xor2(v); v *= 0.25f;
        x = dx * rsqrtf(v + eps) * gnw[c] + gnb[c];
    }
    q[(size_t)b * 64 + c] = x;
}
// dq [B,64] -> dq0 [B, CARD*PWD] (written); dWq [64, PWD], dgnw, dgnb atomically accumulated
__global__ __launch_bounds__(64) void concept_gquery_bwd_kernel(const float* __restrict__ q0, const float* __restrict__ Wq,
                                                               const float* __restrict__ gnw, const float* __restrict__ qraw,
                                                               const float* __restrict__ dq, float* __restrict__ dq0,
                                                               float* __restrict__ dWq, float* __restrict__ dgnw,
                                                               float* __restrict__ dgnb, float eps) {
    __shared__ float s_dx[64];
    const int b = blockIdx.x, c = threadIdx.x, g = c >> 2, d = c & 3;
    const float x = qraw[(size_t)b * 64 + c];
    float gr = dq[(size_t)b * 64 + c];
    if (gnw) {
        float m = x + xmc_xor1(x); m += xmc_xor2(m); m *= 0.25f;
        const float dxm = x - m;
        float v = dxm * dxm; v += xmc_xor1(v); v += xmc_xor2(v); v *= 0.25f;
        const float rstd = rsqrtf(v + eps), xh = dxm * rstd;
        atomicAdd(&dgnw[c], gr * xh);
        atomicAdd(&dgnb[c], gr);
        const float gh = gr * gnw[c];
        float s1 = gh + xmc_xor1(gh); s1 += xmc_xor2(s1); s1 *= 0.25f;
        float s2 = gh * xh; s2 += xmc_xor1(s2); s2 += xmc_xor2(s2); s2 *= 0.25f;
        gr = rstd * (gh - s1 - xh * s2);
    }
    s_dx[c] = gr;
#pragma unroll
    for (int i = 0; i < PWD; ++i) atomicAdd(&dWq[c * PWD + i], gr * q0[(size_t)b * CARD * PWD + g * PWD + i]);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {         // lane (g, d) writes d q0[g][i], i = d*2 + h
        const int i = d * 2 + h;
        float a = 0.f;
#pragma unroll
        for (int o = 0; o < SD; ++o) a += s_dx[g * SD + o] * Wq[(g * SD + o) * PWD + i];
        dq0[(size_t)b * CARD * PWD + g * PWD + i] = a;
    }
}

// ------------------------------------------------------------------------------------------------ head
struct HeadParams {
    const float* Wv;       // [CARD*SD, PWD]     value_gconv (grouped 1x1, no bias)
    const float* We;       // [CARD, SD]         ConceptReasoner.proj_edge
    const float* W1[2];    // [HID, E+SD]        gamma / beta MLP layer 1 (grouped): columns [0,E) sentence, [E,E+SD) concept state
    const float* b1[2];    // [HID]
    const float* W2[2];    // [CARD*PWD, 2*SD]   layer 2 (grouped)
    const float* b2[2];    // [CARD*PWD]
    const float* Ws;       // [SD, E] sent_linear of the self-attention block (471-478), or NULL
};
struct HeadGrads {
    float* Wv; float* We; float* W1[2]; float* b1[2]; float* W2[2]; float* b2[2];
};

// reasoner state of one sample in shared memory; every array is indexed [g][.]
struct HeadState {
    float ctx[CARD][PWD];      // input
    float v[CARD][SD];
    float adj[CARD][CARD];
    float pre[CARD][SD];       // v + adj v  (before the ReLU)
    float r[CARD][SD];
    float a[2][CARD][2 * SD];  // layer-1 pre-activations
    float s[SD], att[CARD];    // sentence->concept attention (P.Ws != NULL): s = Ws sent, att = softmax_g <s, r[g]>
    float c[CARD][SD];         // what the MLPs see: r, or att[g] * r[g]
};

// value projection + ConceptReasoner by the first wave of the workgroup (tid < 64: lane = (g, d)); ends with a barrier for all
__device__ __forceinline__ void reasoner_forward(HeadState& S, const HeadParams& P, int tid) {
    const int g = (tid >> 2) & 15, d = tid & 3;
    if (tid < 64) {   // v[g][d] = sum_i Wv[g*SD+d][i] ctx[g][i]
        float x = 0.f;
#pragma unroll
        for (int i = 0; i < PWD; ++i) x += P.Wv[(g * SD + d) * PWD + i] * S.ctx[g][i];
        S.v[g][d] = x;
    }
    __syncthreads();
    if (tid < 64) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {      // adj[g][k] = tanh(sum_d v[g][d] We[k][d]); lane (g, d) does k = d*4 + kk
            const int k = d * 4 + kk;
            float e = 0.f;
#pragma unroll
            for (int dd = 0; dd < SD; ++dd) e += S.v[g][dd] * P.We[k * SD + dd];
            S.adj[g][k] = tanhf(e);
        }
    }
    __syncthreads();
    if (tid < 64) {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < CARD; ++k) m += S.adj[g][k] * S.v[k][d];
        const float p = S.v[g][d] + m;
        S.pre[g][d] = p;
        S.r[g][d] = fmaxf(p, 0.f);
    }
    __syncthreads();
}

// S.c = the concept state the MLPs are conditioned on.  sent in shared memory; NW waves; ends with a barrier
template <int NW>
__device__ __forceinline__ void context_forward(HeadState& S, const HeadParams& P, const float* s_sent, int E, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    if (P.Ws) {
        for (int d = wave; d < SD; d += NW) {                // one wave per row of Ws
            float x = 0.f;
            for (int i = lane; i < E; i += 64) x += P.Ws[(size_t)d * E + i] * s_sent[i];
            x = wave_sum(x);
            if (lane == 0) S.s[d] = x;
        }
        __syncthreads();
        if (tid < CARD) {                                     // 16 lanes of wave 0: softmax over the concepts
            float l = 0.f;
#pragma unroll
            for (int d = 0; d < SD; ++d) l += S.s[d] * S.r[tid][d];
            float m = l;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const float e = __expf(l - m);
            float z = e;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) z += __shfl_xor(z, o, 64);
            S.att[tid] = e / z;
        }
        __syncthreads();
    }
    if (tid < 64) {
        const int g = tid >> 2, d = tid & 3;
        S.c[g][d] = P.Ws ? S.att[g] * S.r[g][d] : S.r[g][d];
    }
    __syncthreads();
}

// ctx [B,CARD,PWD], sent [B,E] -> gamma, beta [B, CARD*PWD], hid [B, 2*HID] (layer-1 pre-activations, kept for the backward)
// a_pre (optional) f32 [2][B][HID]: the sentence part of layer 1, W1[t][:, :E] . sent, computed ahead for every stage of the generator in one
// grouped GEMM (it depends on nothing but the sentence vector; xmc_concept_head_fwd_pre) -- the kernel then skips its 256 rows x E dot
__global__ __launch_bounds__(256) void concept_head_fwd_kernel(const float* __restrict__ ctx, const float* __restrict__ sent,
                                                              HeadParams P, float* __restrict__ gamma, float* __restrict__ beta,
                                                              float* __restrict__ hid, int E, const float* __restrict__ a_pre, int B) {
    __shared__ HeadState S;
    __shared__ __attribute__((aligned(16))) float s_sent[1024];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < CARD * PWD) (&S.ctx[0][0])[tid] = ctx[(size_t)b * CARD * PWD + tid];
    for (int i = tid; i < E; i += 256) s_sent[i] = sent[(size_t)b * E + i];
    __syncthreads();
    // thread tid owns layer-1 unit (t, row) = (tid >> 7, tid & 127), row = g*8 + o
    const int t = wave >> 1, row = (wave & 1) * 64 + lane, ld = E + SD;
    float a = (a_pre ? a_pre[((size_t)t * B + b) * HID + row] : rows_dot64(P.W1[t] + (size_t)(wave & 1) * 64 * ld, ld, s_sent, E, lane)) + P.b1[t][row];
    reasoner_forward(S, P, tid);
    context_forward<4>(S, P, s_sent, E, tid);
    {
        const int g = row >> 3;
        const float* w = P.W1[t] + (size_t)row * ld + E;
#pragma unroll
        for (int dd = 0; dd < SD; ++dd) a += w[dd] * S.c[g][dd];
        (&S.a[t][0][0])[row] = a;
        hid[(size_t)b * 2 * HID + tid] = a;
    }
    __syncthreads();
    {   // out[t][g][o]: the same (t, row) indexing, row = g*PWD + o
        const int g = row >> 3;
        float x = P.b2[t][row];
#pragma unroll
        for (int i = 0; i < 2 * SD; ++i) x += P.W2[t][row * 2 * SD + i] * lrelu02(S.a[t][g][i]);
        (t == 0 ? gamma : beta)[(size_t)b * CARD * PWD + row] = x;
    }
}

// dgamma, dbeta [B, CARD*PWD], hid -> dctx [B,CARD,PWD] (written), da [B, 2*HID] (written: d of the layer-1 pre-activations,
// for concept_outer_kernel), small parameter gradients (atomics).
// One WAVE per sample, HB_SPW samples per workgroup.  The parameter gradients are sums over the batch of per-sample products.  Formed by
// the sample's lanes as they went, each lane issued 56 float atomics (same addresses for every sample; as LDS atomics 16 lanes per bank):
// 19 of the kernel's 32 us at B = 64, 24 launches per iteration.  Here the waves leave their factors in LDS (they are there anyway), and
// after the last per-sample phase ALL threads of the workgroup form the HB_SPW-sample partial sum of 16 gradient elements each -- plain LDS
// reads and FMAs -- and add it to memory once: no LDS atomics, 1 / HB_SPW of the global ones.
constexpr int HB_SPW = 4;
struct HeadBwdState {
    float d_o[2][CARD][PWD], da[2][CARD][2 * SD], dr[CARD][SD], dm[CARD][SD], de[CARD][CARD], dv[CARD][SD];
};
__global__ __launch_bounds__(64 * HB_SPW) void concept_head_bwd_kernel(const float* __restrict__ ctx, const float* __restrict__ sent,
                                                                      const float* __restrict__ hid, HeadParams P,
                                                                      const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                                      float* __restrict__ dctx, float* __restrict__ da_out,
                                                                      float* __restrict__ ds_out, HeadGrads G, int E, int B) {
    __shared__ HeadState Sw[HB_SPW];
    __shared__ HeadBwdState Tw[HB_SPW];
    extern __shared__ __attribute__((aligned(16))) float s_sent_all[];        // [HB_SPW][E] (read only when P.Ws)
    const int wave = threadIdx.x >> 6, tid = threadIdx.x & 63;               // tid: lane = (g, d) of this wave's sample
    const int bs = blockIdx.x * HB_SPW + wave;
    const bool on = bs < B;
    const int b = on ? bs : B - 1;                                            // a spare wave walks the last sample and adds nothing
    HeadState& S = Sw[wave];
    HeadBwdState& T = Tw[wave];
    float* s_sent = s_sent_all + (size_t)wave * E;
    const int g = tid >> 2, d = tid & 3;
    for (int i = tid; i < CARD * PWD; i += 64) {
        (&S.ctx[0][0])[i] = ctx[(size_t)b * CARD * PWD + i];
        (&T.d_o[0][0][0])[i] = on ? dgamma[(size_t)b * CARD * PWD + i] : 0.f;
        (&T.d_o[1][0][0])[i] = on ? dbeta[(size_t)b * CARD * PWD + i] : 0.f;
    }
    for (int i = tid; i < 2 * HID; i += 64) (&S.a[0][0][0])[i] = hid[(size_t)b * 2 * HID + i];
    if (P.Ws) for (int i = tid; i < E; i += 64) s_sent[i] = sent[(size_t)b * E + i];
    __syncthreads();
    reasoner_forward(S, P, tid);
    context_forward<1>(S, P, s_sent, E, tid);
    const int ld = E + SD;
    // ---- layer 2: out = W2 lrelu(a) + b2
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {     // lane owns hidden unit i = d*2+h of group g: dh_i = sum_o W2[o][i] do[o]
            const int i = d * 2 + h;
            float dh = 0.f;
#pragma unroll
            for (int o = 0; o < PWD; ++o) dh += P.W2[t][(g * PWD + o) * 2 * SD + i] * T.d_o[t][g][o];
            const float ai = S.a[t][g][i];
            const float da = dh * (ai > 0.f ? 1.f : 0.2f);
            T.da[t][g][i] = da;
            if (on) da_out[(size_t)b * 2 * HID + t * HID + g * 2 * SD + i] = da;
        }
    }
    __syncthreads();
    {   // d r[g][d] through layer 1's concept-state columns
        float dr = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int o = 0; o < 2 * SD; ++o) dr += P.W1[t][(size_t)(g * 2 * SD + o) * ld + E + d] * T.da[t][g][o];
        if (P.Ws) {                       // c[g][d] = att[g] r[g][d], att = softmax_g(l), l[g] = <s, r[g]>, s = Ws sent
            const float dc = dr;
            float da_g = dc * S.r[g][d];                          // d att[g] = <dc[g], r[g]>
            da_g += xmc_xor1(da_g); da_g += xmc_xor2(da_g);
            float dot = (d == 0) ? S.att[g] * da_g : 0.f;         // sum_g att[g] d att[g]
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) dot += __shfl_xor(dot, o, 64);
            const float dl = S.att[g] * (da_g - dot);             // d l[g]
            dr = dc * S.att[g] + dl * S.s[d];
            float dsd = dl * S.r[g][d];                           // d s[d] = sum_g dl[g] r[g][d]: lanes with the same d
#pragma unroll
            for (int o = 32; o >= 4; o >>= 1) dsd += __shfl_xor(dsd, o, 64);
            if (g == 0 && on) ds_out[(size_t)b * SD + d] = dsd;
        }
        T.dr[g][d] = dr;
        // ---- reasoner: r = relu(pre), pre = v + adj v, adj = tanh(v We^T)
        T.dm[g][d] = S.pre[g][d] > 0.f ? dr : 0.f;          // d pre (= d m, and the direct part of d v)
    }
    __syncthreads();
    {
        float dv = T.dm[g][d];    // direct
#pragma unroll
        for (int gg = 0; gg < CARD; ++gg) dv += S.adj[gg][g] * T.dm[gg][d];      // through m[gg] = sum_k adj[gg][k] v[k]
        T.dv[g][d] = dv;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {      // d adj[g][k] = sum_d dm[g][d] v[k][d];  de = dadj * (1 - adj^2)
        const int k = d * 4 + kk;
        float da = 0.f;
#pragma unroll
        for (int dd = 0; dd < SD; ++dd) da += T.dm[g][dd] * S.v[k][dd];
        const float aj = S.adj[g][k];
        T.de[g][k] = da * (1.f - aj * aj);
    }
    __syncthreads();
    {
        float dv = T.dv[g][d];
#pragma unroll
        for (int k = 0; k < CARD; ++k) dv += T.de[g][k] * P.We[k * SD + d];       // e[g][k] = sum_d v[g][d] We[k][d]
        T.dv[g][d] = dv;
    }
    __syncthreads();
    // ---- value projection: v[g][d] = sum_i Wv[g*SD+d][i] ctx[g][i]
    if (on) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = d * 2 + h;
            float dc = 0.f;
#pragma unroll
            for (int dd = 0; dd < SD; ++dd) dc += P.Wv[(g * SD + dd) * PWD + i] * T.dv[g][dd];
            dctx[(size_t)b * CARD * PWD + g * PWD + i] = dc;
        }
    }
    __syncthreads();
    // ---- parameter gradients: this workgroup's samples summed per element, one atomic per element
    constexpr int NTH = 64 * HB_SPW;
    for (int e = threadIdx.x; e < 2 * CARD * PWD * 2 * SD; e += NTH) {        // dW2[t][g*PWD+o][i] = sum do[t][g][o] lrelu(a[t][g][i])
        const int t = e >> 10, r = e & 1023, go = r >> 3, i = r & 7, gq = go >> 3, o = go & 7;
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < HB_SPW; ++w) x += Tw[w].d_o[t][gq][o] * lrelu02(Sw[w].a[t][gq][i]);
        atomicAdd(&G.W2[t][r], x);
    }
    for (int e = threadIdx.x; e < 2 * HID * SD; e += NTH) {                    // dW1[t][g*8+i][E+dd] = sum da[t][g][i] c[g][dd]
        const int t = e >> 9, r = e & 511, row = r >> 2, dd = r & 3, gq = row >> 3, i = row & 7;
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < HB_SPW; ++w) x += Tw[w].da[t][gq][i] * Sw[w].c[gq][dd];
        atomicAdd(&G.W1[t][(size_t)row * ld + E + dd], x);
    }
    for (int e = threadIdx.x; e < 2 * 2 * HID; e += NTH) {                     // db2[t][g*PWD+o] = sum do,  db1[t][g*8+i] = sum da
        const int which = e >> 8, t = (e >> 7) & 1, r = e & 127, gq = r >> 3, i = r & 7;
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < HB_SPW; ++w) x += which ? Tw[w].da[t][gq][i] : Tw[w].d_o[t][gq][i];
        atomicAdd(which ? &G.b1[t][r] : &G.b2[t][r], x);
    }
    for (int e = threadIdx.x; e < CARD * SD * PWD; e += NTH) {                 // dWv[g*SD+d][i] = sum dv[g][d] ctx[g][i]
        const int gd = e >> 3, i = e & 7, gq = gd >> 2, dq = gd & 3;
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < HB_SPW; ++w) x += Tw[w].dv[gq][dq] * Sw[w].ctx[gq][i];
        atomicAdd(&G.Wv[e], x);
    }
    if (threadIdx.x < CARD * SD) {                                            // dWe[k][d] = sum_g de[g][k] v[g][d]
        const int k = threadIdx.x >> 2, dq = threadIdx.x & 3;
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < HB_SPW; ++w)
#pragma unroll
            for (int gg = 0; gg < CARD; ++gg) x += Tw[w].de[gg][k] * Sw[w].v[gg][dq];
        atomicAdd(&G.We[threadIdx.x], x);
    }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int xmc_concept_query_fwd(const float* sent, const float* Wq, const float* gnw, const float* gnb, float* q, float* qraw,
                                     int B, int E, float eps, void* stream) {
    if (!sent || !Wq || !q || !qraw || B < 1 || E < 1 || E > 1024 || (gnw == nullptr) != (gnb == nullptr)) return XMC_EINVAL;
    hipLaunchKernelGGL(concept_query_fwd_kernel, dim3(B), dim3(64), 0, ST(stream), sent, Wq, gnw, gnb, q, qraw, E, eps);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_concept_query_bwd(const float* sent, const float* Wq, const float* gnw, const float* qraw, const float* dq,
                                     float* dsent, float* dWq, float* dgnw, float* dgnb, float* scratch, int B, int E, float eps,
                                     void* stream) {
    if (!sent || !Wq || !qraw || !dq || !dsent || !dWq || !scratch || B < 1 || E < 1) return XMC_EINVAL;
    if (gnw && (!dgnw || !dgnb)) return XMC_EINVAL;
    hipLaunchKernelGGL(concept_query_bwd_kernel, dim3(B), dim3(64), 0, ST(stream), gnw, qraw, dq, scratch, dgnw, dgnb, eps);
    XMC_LAUNCH_CHECK();
    OuterArgs a;
    a.D = scratch; a.X = sent; a.W[0] = Wq; a.W[1] = nullptr; a.dW[0] = dWq; a.dW[1] = nullptr; a.dX = dsent;
    a.B = B; a.R = 64; a.Rp = 64; a.C = E; a.ldw = E; a.acc_dx = 0; a.kchunk = 0;
    hipLaunchKernelGGL(concept_outer_kernel, dim3(a.R / 2 + B), dim3(256), 0, ST(stream), a);
    XMC_LAUNCH_CHECK();
    return 0;
}

// all S <= 32 sampler stages at once.  Wq / gnw / gnb: host arrays of S device pointers ([64][E], [64], [64]; gnw[s] and gnb[s] both NULL =
// no GroupNorm); q, qraw f32 [S][B][64].  Backward: dq [S][B][64] -> dsent [B][E] (written: the sum over the stages), dWq f32 [S][64][E]
// (written), dgn f32 [S][2][64] = (d gnw, d gnb) (accumulated: zeroed by the caller), scratch f32 [B][S*64].
extern "C" int xmc_concept_query_fwd_multi(const float* sent, const float* const* Wq, const float* const* gnw, const float* const* gnb, int S,
                                           float* q, float* qraw, int B, int E, float eps, void* stream) {
    if (!sent || !Wq || !gnw || !gnb || !q || !qraw || S < 1 || S > 32 || B < 1 || E < 1 || E > 1024) return XMC_EINVAL;
    QueryTab T;
    for (int s = 0; s < 32; ++s) {
        T.Wq[s] = s < S ? Wq[s] : nullptr; T.gnw[s] = s < S ? gnw[s] : nullptr; T.gnb[s] = s < S ? gnb[s] : nullptr;
        if (s < S && (!Wq[s] || (gnw[s] == nullptr) != (gnb[s] == nullptr))) return XMC_EINVAL;
    }
    hipLaunchKernelGGL(concept_query_fwd_multi_kernel, dim3(B, S), dim3(64), 0, ST(stream), sent, T, q, qraw, B, E, eps);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_concept_query_bwd_multi(const float* sent, const float* const* Wq, const float* const* gnw, int S, const float* qraw,
                                           const float* dq, float* dsent, float* dWq, float* dgn, float* scratch, int B, int E, float eps,
                                           void* stream) {
    if (!sent || !Wq || !gnw || !qraw || !dq || !dsent || !dWq || !dgn || !scratch || S < 1 || S > 32 || B < 1 || E < 1) return XMC_EINVAL;
    QueryTab T;
    for (int s = 0; s < 32; ++s) { T.Wq[s] = s < S ? Wq[s] : nullptr; T.gnw[s] = s < S ? gnw[s] : nullptr; T.gnb[s] = nullptr; }
    hipLaunchKernelGGL(concept_query_bwd_multi_kernel, dim3(B, S), dim3(64), 0, ST(stream), T, qraw, dq, scratch, dgn, B, S, eps);
    XMC_LAUNCH_CHECK();
    OuterArgs a;
    for (int s = 0; s < QMAX; ++s) { a.W[s] = s < S ? Wq[s] : nullptr; a.dW[s] = s < S ? dWq + (size_t)s * 64 * E : nullptr; }
    a.D = scratch; a.X = sent; a.dX = dsent;
    a.B = B; a.R = S * 64; a.Rp = 64; a.C = E; a.ldw = E; a.acc_dx = 0; a.kchunk = 8;        // dsent: zeroed by the caller, added atomically
    if (xmc_zero_acc(dsent, sizeof(float) * (size_t)B * E, ST(stream)) != hipSuccess) return XMC_EINVAL;
    hipLaunchKernelGGL(concept_outer_kernel, dim3(a.R / 2 + B * ((S + 7) / 8)), dim3(256), 0, ST(stream), a);
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_concept_gquery_fwd(const float* q0, const float* Wq, const float* gnw, const float* gnb, float* q, float* qraw,
                                      int B, float eps, void* stream) {
    if (!q0 || !Wq || !q || !qraw || B < 1 || (gnw == nullptr) != (gnb == nullptr)) return XMC_EINVAL;
    hipLaunchKernelGGL(concept_gquery_fwd_kernel, dim3(B), dim3(64), 0, ST(stream), q0, Wq, gnw, gnb, q, qraw, eps);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_concept_gquery_bwd(const float* q0, const float* Wq, const float* gnw, const float* qraw, const float* dq,
                                      float* dq0, float* dWq, float* dgnw, float* dgnb, int B, float eps, void* stream) {
    if (!q0 || !Wq || !qraw || !dq || !dq0 || !dWq || B < 1 || (gnw && (!dgnw || !dgnb))) return XMC_EINVAL;
    hipLaunchKernelGGL(concept_gquery_bwd_kernel, dim3(B), dim3(64), 0, ST(stream), q0, Wq, gnw, qraw, dq, dq0, dWq, dgnw, dgnb, eps);
    XMC_LAUNCH_CHECK();
    return 0;
}

// params / grads: 11 pointers each in the order Wv, We, W1g, b1g, W2g, b2g, W1b, b1b, W2b, b2b, Ws (Ws: sent_linear of the
// self-attention block or NULL, then grads[10] is not used)
static int head_params(const float* const* p, HeadParams& P) {
    P.Ws = p[10];
    for (int i = 0; i < 10; ++i) if (!p[i]) return 0;
    P.Wv = p[0]; P.We = p[1];
    P.W1[0] = p[2]; P.b1[0] = p[3]; P.W2[0] = p[4]; P.b2[0] = p[5];
    P.W1[1] = p[6]; P.b1[1] = p[7]; P.W2[1] = p[8]; P.b2[1] = p[9];
    return 1;
}
extern "C" int xmc_concept_head_fwd(const float* ctx, const float* sent, const float* const* params, float* gamma, float* beta,
                                    float* hid, int B, int E, void* stream) {
    HeadParams P;
    if (!ctx || !sent || !params || !gamma || !beta || !hid || B < 1 || E < 1 || E > 1024 || !head_params(params, P)) return XMC_EINVAL;
    hipLaunchKernelGGL(concept_head_fwd_kernel, dim3(B), dim3(256), 0, ST(stream), ctx, sent, P, gamma, beta, hid, E, (const float*)nullptr, B);
    XMC_LAUNCH_CHECK();
    return 0;
}
// the same with the sentence part of layer 1 handed in: a_pre f32 [2][B][128], a_pre[t][b][row] = sum_i W1_t[row][i] sent[b][i] (i < E)
extern "C" int xmc_concept_head_fwd_pre(const float* ctx, const float* sent, const float* const* params, const float* a_pre, float* gamma,
                                        float* beta, float* hid, int B, int E, void* stream) {
    HeadParams P;
    if (!ctx || !sent || !params || !a_pre || !gamma || !beta || !hid || B < 1 || E < 1 || E > 1024 || !head_params(params, P)) return XMC_EINVAL;
    hipLaunchKernelGGL(concept_head_fwd_kernel, dim3(B), dim3(256), 0, ST(stream), ctx, sent, P, gamma, beta, hid, E, a_pre, B);
    XMC_LAUNCH_CHECK();
    return 0;
}
static int head_bwd_go(const float* ctx, const float* sent, const float* hid, const float* const* params, const float* dgamma,
                       const float* dbeta, float* dctx, float* dsent, float* const* grads, float* scratch, int B, int E, void* stream, bool defer);
extern "C" int xmc_concept_head_bwd(const float* ctx, const float* sent, const float* hid, const float* const* params,
                                    const float* dgamma, const float* dbeta, float* dctx, float* dsent, float* const* grads,
                                    float* scratch, int B, int E, void* stream) {
    return head_bwd_go(ctx, sent, hid, params, dgamma, dbeta, dctx, dsent, grads, scratch, B, E, stream, false);
}
// the same WITHOUT the batch products of layer 1's sentence columns (dW1[:, :E] and its share of dsent): scratch[:, :256] = d of the layer-1
// pre-activations is the result the caller collects from every stage and hands to ONE xmc_concept_outer_multi.  dsent is written only for
// the self-attention kind (its sent_linear term), and not read.
extern "C" int xmc_concept_head_bwd_pre(const float* ctx, const float* sent, const float* hid, const float* const* params,
                                        const float* dgamma, const float* dbeta, float* dctx, float* dsent, float* const* grads,
                                        float* scratch, int B, int E, void* stream) {
    return head_bwd_go(ctx, sent, hid, params, dgamma, dbeta, dctx, dsent, grads, scratch, B, E, stream, true);
}
// D [B][nW * Rp] x X [B][C]: dW[k][r][c < C] = sum_b D[b][k Rp + r] X[b][c] (written, row pitch ldw), dX[b][c] = sum_k,r D[b][k Rp + r] W[k][r][c]
// (written); nW <= 64.  The batch products of every stage's heads in one launch.
extern "C" int xmc_concept_outer_multi(const float* D, const float* X, const float* const* W, float* const* dW, int nW, int Rp, float* dX,
                                       int B, int C, int ldw, void* stream) {
    if (!D || !X || !W || !dW || !dX || nW < 1 || nW > QMAX || Rp < 2 || (Rp & 1) || B < 1 || C < 1 || ldw < C) return XMC_EINVAL;
    OuterArgs a;
    for (int k = 0; k < QMAX; ++k) { a.W[k] = k < nW ? W[k] : nullptr; a.dW[k] = k < nW ? dW[k] : nullptr; if (k < nW && (!W[k] || !dW[k])) return XMC_EINVAL; }
    a.D = D; a.X = X; a.dX = dX; a.B = B; a.R = nW * Rp; a.Rp = Rp; a.C = C; a.ldw = ldw; a.acc_dx = 0; a.kchunk = 4;
    if (xmc_zero_acc(dX, sizeof(float) * (size_t)B * C, ST(stream)) != hipSuccess) return XMC_EINVAL;
    hipLaunchKernelGGL(concept_outer_kernel, dim3(a.R / 2 + B * ((nW + 3) / 4)), dim3(256), 0, ST(stream), a);
    XMC_LAUNCH_CHECK();
    return 0;
}
static int head_bwd_go(const float* ctx, const float* sent, const float* hid, const float* const* params, const float* dgamma,
                       const float* dbeta, float* dctx, float* dsent, float* const* grads, float* scratch, int B, int E, void* stream, bool defer) {
    HeadParams P;
    if (!ctx || !sent || !hid || !params || !dgamma || !dbeta || !dctx || !dsent || !grads || !scratch || B < 1 || E < 1 || E > 1024 ||
        !head_params(params, P))
        return XMC_EINVAL;
    for (int i = 0; i < 10; ++i) if (!grads[i]) return XMC_EINVAL;
    if (P.Ws && !grads[10]) return XMC_EINVAL;
    HeadGrads G;
    G.Wv = grads[0]; G.We = grads[1];
    G.W1[0] = grads[2]; G.b1[0] = grads[3]; G.W2[0] = grads[4]; G.b2[0] = grads[5];
    G.W1[1] = grads[6]; G.b1[1] = grads[7]; G.W2[1] = grads[8]; G.b2[1] = grads[9];
    float* ds = scratch + (size_t)B * 2 * HID;       // [B, SD]: d (Ws sent)
    hipLaunchKernelGGL(concept_head_bwd_kernel, dim3((B + HB_SPW - 1) / HB_SPW), dim3(64 * HB_SPW), sizeof(float) * HB_SPW * (size_t)E, ST(stream),
                       ctx, sent, hid, P, dgamma, dbeta, dctx, scratch, ds, G, E, B);
    XMC_LAUNCH_CHECK();
    OuterArgs a;
    a.D = scratch; a.X = sent; a.W[0] = P.W1[0]; a.W[1] = P.W1[1]; a.dW[0] = G.W1[0]; a.dW[1] = G.W1[1]; a.dX = dsent;
    a.B = B; a.R = 2 * HID; a.Rp = HID; a.C = E; a.ldw = E + SD; a.acc_dx = 0; a.kchunk = 0;
    if (!defer) {
        hipLaunchKernelGGL(concept_outer_kernel, dim3(a.R / 2 + B), dim3(256), 0, ST(stream), a);
        XMC_LAUNCH_CHECK();
    }
    if (P.Ws) {                                       // dWs = ds^T sent;  dsent += ds Ws  (deferred: dsent = ds Ws)
        a.D = ds; a.W[0] = P.Ws; a.W[1] = nullptr; a.dW[0] = grads[10]; a.dW[1] = nullptr;
        a.R = SD; a.Rp = SD; a.ldw = E; a.acc_dx = defer ? 0 : 1;
        hipLaunchKernelGGL(concept_outer_kernel, dim3(a.R / 2 + B), dim3(256), 0, ST(stream), a);
        XMC_LAUNCH_CHECK();
    }
    return 0;
}
