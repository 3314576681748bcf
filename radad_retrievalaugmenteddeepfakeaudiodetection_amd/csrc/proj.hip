// proj.hip -- ProjectionLayer / RADADModel inference forward (projection.py:68-106, radad_model.py:32-41) on gfx950.
//
//   s = W2 tanh(W1 x + b1) + b2          (:69-71)      a = softmax_K(s)            (:87)
//   c = W4 relu(W3 x + b3) + b4          (:74-76)      u = sum_K a c               (:88-89)
//   out = W6 LN(W5 u + b5; eps 1e-6) + b6 (:94-101)    (dropout is the identity in eval)
//
// Two algebraic folds keep every D-wide tensor except x itself out of HBM (inference: weights are frozen):
//   * sum_K a_k (W4 h_k + b4) = W4 (sum_K a_k h_k) + b4           (the softmax weights sum to one)
//   * W5 (W4 hbar + b4) + b5  = (W5 W4) hbar + (W5 b4 + b5)       (W54 [H,H] folded once, float64 accumulate)
// so the forward is ONE pass over x -- a split-K NT GEMM against the stacked [W1;W3] rows on
// v_mfma_f32_32x32x2_f32 (128x128x32 tiles, double-buffered LDS; split-K because B*K/128 x 2H/128 tiles alone
// would occupy 4-40 of the 256 CUs) -- followed by one block per batch row that reduces the K-splits, applies
// tanh / relu, the softmax over the K neighbours, the H x H fold, LayerNorm and the output Linear from LDS.
// The same GEMM + a per-row tail serve RADADModel's fuse Linear and the detection MLP (radad_model.py:39-40).
#include "common.h"

namespace {

constexpr int GT = 128;       // tile edge (rows of A and rows of W)
constexpr int GK = 32;
constexpr int GLD = GK + 4;
constexpr int G_THREADS = 256;
constexpr int TAIL_THREADS = 256;
constexpr int SPLIT_TARGET_BLOCKS = 512;   // two resident blocks per CU

enum { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2 };

struct GemmParams {
    const float* a; int64_t lda;      // [M, K]
    const float* w;                   // rows [0, n_lo) of the [N, K] weight (nn.Linear layout)
    const float* w_hi; int n_lo;      // rows [n_lo, N): a second Linear stacked under the first (or unused)
    int64_t ldw;
    float* part;                      // [S, M, N] raw K-split partial products
    int M, N, K, chunks_per_split;
};

// part[z] = A[:, Kz] W[:, Kz]^T for the K range of split z
__global__ __launch_bounds__(G_THREADS, 2) void k_gemm_nt_splitk(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sA = reinterpret_cast<float*>(smem);   // [2][GT][GLD]
    float* sB = sA + 2 * GT * GLD;                // [2][GT][GLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * GT, n0 = blockIdx.y * GT;
    const int nk = (p.K + GK - 1) / GK;
    const int kb = blockIdx.z * p.chunks_per_split;
    const int ke = min(nk, kb + p.chunks_per_split);
    const int ld_r = tid >> 3, ld_c = (tid & 7) * 4;
    f32x4 ra[4], rb[4];
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const float* arow[4];
    const float* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ld_r + 32 * i, n = n0 + ld_r + 32 * i;
        arow[i] = m < p.M ? p.a + (int64_t)m * p.lda : nullptr;
        wrow[i] = n >= p.N ? nullptr : (n < p.n_lo ? p.w + (int64_t)n * p.ldw : p.w_hi + (int64_t)(n - p.n_lo) * p.ldw);
    }
    auto gload = [&](int kc) {
        const int kcol = kc * GK + ld_c;
        const bool kin = kcol < p.K;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = (kin && arow[i]) ? *reinterpret_cast<const f32x4*>(arow[i] + kcol) : z;
            rb[i] = (kin && wrow[i]) ? *reinterpret_cast<const f32x4*>(wrow[i] + kcol) : z;
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ld_r + 32 * i;
            *reinterpret_cast<f32x4*>(sA + (buf * GT + r) * GLD + ld_c) = ra[i];
            *reinterpret_cast<f32x4*>(sB + (buf * GT + r) * GLD + ld_c) = rb[i];
        }
    };
    gload(kb);
    swrite(0);
    __syncthreads();
    for (int kc = kb; kc < ke; ++kc) {
        const int buf = (kc - kb) & 1;
        if (kc + 1 < ke) gload(kc + 1);
        const float* a_base = sA + (buf * GT + wm * 64 + l31) * GLD + 4 * lh;
        const float* b_base = sB + (buf * GT + wn * 64 + l31) * GLD + 4 * lh;
#pragma unroll
        for (int kk = 0; kk < GK / 8; ++kk) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(a_base + kk * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(a_base + 32 * GLD + kk * 8);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b_base + kk * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(b_base + 32 * GLD + kk * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
        }
        if (kc + 1 < ke) swrite(buf ^ 1);
        __syncthreads();
    }
    // acc[mt][nt][r]: row m0 + wm*64 + mt*32 + (r&3) + 8(r>>2) + 4lh, column n0 + wn*64 + nt*32 + l31
    float* out = p.part + (int64_t)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + wn * 64 + nt * 32 + l31;
        if (n >= p.N) continue;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M) out[(int64_t)m * p.N + n] = acc[mt][nt][r];
            }
    }
}

// out[m, n] = act(sum_z part[z, m, n] + bias[n])
__global__ __launch_bounds__(256) void k_splitk_finish(const float* __restrict__ part, int S, int64_t M, int N,
                                                       const float* __restrict__ bias, int act, float* __restrict__ out,
                                                       int64_t ldo) {
    const int64_t total = M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / N;
        const int n = (int)(i - m * N);
        float v = 0.f;
        for (int z = 0; z < S; ++z) v += part[(int64_t)z * total + i];
        if (bias) v += bias[n];
        if (act == ACT_TANH) v = tanhf(v);
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        out[m * ldo + n] = v;
    }
}

// W54t[h, j] = sum_d W5[j, d] W4[d, h] (stored input-major so the tail reads it coalesced); b54 = W5 b4 + b5
__global__ __launch_bounds__(256) void k_fold_w54(const float* __restrict__ w4, const float* __restrict__ b4,
                                                  const float* __restrict__ w5, const float* __restrict__ b5, int D, int H,
                                                  float* __restrict__ w54t, float* __restrict__ b54) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * H) return;
    const int j = idx / H, h = idx - j * H;
    const float* w5r = w5 + (int64_t)j * D;
    double acc = 0.0;
    for (int d = 0; d < D; ++d) acc += (double)w5r[d] * (double)w4[(int64_t)d * H + h];
    w54t[(int64_t)h * H + j] = (float)acc;
    if (h == 0) {
        double bb = (double)b5[j];
        for (int d = 0; d < D; ++d) bb += (double)w5r[d] * (double)b4[d];
        b54[j] = (float)bb;
    }
}

__device__ __forceinline__ float block_sum(float v, float* s_red) {   // s_red: >= 4 floats, all threads call
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
    return t;
}

struct TailParams {
    const float* part; int S;        // [S, B*K, 2H]: columns [0,H) = W1 x, [H,2H) = W3 x
    const float *b1, *w2, *b2, *b3;
    const float *w54t, *b54;         // [H(in), H(out)], [H]
    const float *ln_g, *ln_b;
    const float *w6, *b6;            // [O, H], [O]
    float* out;                      // [B, O]
    int64_t B; int K, H, O;
};

// one block per batch row: K-split reduce -> tanh / relu -> scores -> softmax_K -> hbar -> W54 -> LN -> W6
__global__ __launch_bounds__(TAIL_THREADS) void k_proj_tail(TailParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K, H = p.H;
    float* s_hc = reinterpret_cast<float*>(smem);   // [K][H] relu branch
    float* s_sw = s_hc + K * H;                     // [K][H] tanh(.) * w2
    float* s_hbar = s_sw + K * H;                   // [H]
    float* s_y = s_hbar + H;                        // [H]
    float* s_score = s_y + H;                       // [K]
    float* s_red = s_score + K;                     // [4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;
    const int64_t MN = p.B * K * 2 * H;
    for (int idx = tid; idx < K * H; idx += TAIL_THREADS) {
        const int k = idx / H, h = idx - k * H;
        const float* src = p.part + ((b * K + k) * 2) * H + h;
        float sa = 0.f, sc = 0.f;
        for (int z = 0; z < p.S; ++z) {
            sa += src[(int64_t)z * MN];
            sc += src[(int64_t)z * MN + H];
        }
        s_sw[idx] = tanhf(sa + p.b1[h]) * p.w2[h];
        s_hc[idx] = fmaxf(sc + p.b3[h], 0.f);
    }
    __syncthreads();
    for (int k = wave; k < K; k += TAIL_THREADS / 64) {
        float s = 0.f;
        for (int h = lane; h < H; h += 64) s += s_sw[k * H + h];
        s = wave_sum(s);
        if (lane == 0) s_score[k] = s + p.b2[0];
    }
    __syncthreads();
    float mx = -INFINITY, den = 0.f;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, s_score[k]);
    for (int k = 0; k < K; ++k) den += expf(s_score[k] - mx);
    const float rden = 1.0f / den;
    for (int h = tid; h < H; h += TAIL_THREADS) {
        float v = 0.f;
        for (int k = 0; k < K; ++k) v += expf(s_score[k] - mx) * rden * s_hc[k * H + h];
        s_hbar[h] = v;
    }
    __syncthreads();
    float lsum = 0.f;
    for (int j = tid; j < H; j += TAIL_THREADS) {
        float v = p.b54[j];
        for (int h = 0; h < H; ++h) v += p.w54t[(int64_t)h * H + j] * s_hbar[h];
        s_y[j] = v;
        lsum += v;
    }
    const float mu = block_sum(lsum, s_red) / (float)H;
    float lvar = 0.f;
    for (int j = tid; j < H; j += TAIL_THREADS) { const float d = s_y[j] - mu; lvar += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum(lvar, s_red) / (float)H + 1e-6f);   // biased variance, eps 1e-6 (:51)
    for (int j = tid; j < H; j += TAIL_THREADS) s_hbar[j] = (s_y[j] - mu) * rstd * p.ln_g[j] + p.ln_b[j];
    __syncthreads();
    for (int o = wave; o < p.O; o += TAIL_THREADS / 64) {
        const float* wr = p.w6 + (int64_t)o * H;
        float v = 0.f;
        for (int h = lane; h < H; h += 64) v += wr[h] * s_hbar[h];
        v = wave_sum(v);
        if (lane == 0) p.out[b * p.O + o] = v + p.b6[o];
    }
}

constexpr int HEAD_MAX_LAYERS = RADAD_HEAD_MAX_LAYERS;

struct HeadParams {
    const float* part; int S;          // [S, B, P]: tpp . Wf[:, :D]^T
    const float* wf; int64_t ldwf; int D;   // Wf [P, D+P]
    const float* bf;
    const float* proj;                 // [B, P]
    const float *lw[HEAD_MAX_LAYERS], *lb[HEAD_MAX_LAYERS], *ls[HEAD_MAX_LAYERS], *lt[HEAD_MAX_LAYERS];
    int dims[HEAD_MAX_LAYERS + 1];
    int n_layers;
    float* fused;                      // [B, P] or nullptr
    float* logits;                     // [B, dims[n_layers]]
    int64_t B; int P, max_dim;
};

// one block per batch row: fused = Wf [tpp ; proj] + bf (radad_model.py:39), then the detection MLP
// (detection_model.py:41-72 in eval: Linear -> BatchNorm as scale/shift -> ReLU, dropout = identity)
__global__ __launch_bounds__(TAIL_THREADS) void k_fuse_head(HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_a = reinterpret_cast<float*>(smem);   // [max_dim]
    float* s_b = s_a + p.max_dim;                  // [max_dim]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;
    const int P = p.P;
    for (int j = tid; j < P; j += TAIL_THREADS) s_b[j] = p.proj[b * P + j];
    __syncthreads();
    for (int o = wave; o < P; o += TAIL_THREADS / 64) {
        const float* wr = p.wf + (int64_t)o * p.ldwf + p.D;
        float v = 0.f;
        for (int j = lane; j < P; j += 64) v += wr[j] * s_b[j];
        v = wave_sum(v);
        if (lane == 0) {
            float t = 0.f;
            for (int z = 0; z < p.S; ++z) t += p.part[((int64_t)z * p.B + b) * P + o];
            v += t + p.bf[o];
            s_a[o] = v;
            if (p.fused) p.fused[b * P + o] = v;
        }
    }
    __syncthreads();
    float* cur = s_a;
    float* nxt = s_b;
    for (int l = 0; l < p.n_layers; ++l) {
        const int din = p.dims[l], dout = p.dims[l + 1];
        const bool last = l + 1 == p.n_layers;
        for (int o = wave; o < dout; o += TAIL_THREADS / 64) {
            const float* wr = p.lw[l] + (int64_t)o * din;
            float v = 0.f;
            for (int j = lane; j < din; j += 64) v += wr[j] * cur[j];
            v = wave_sum(v);
            if (lane == 0) {
                v += p.lb[l][o];
                if (p.ls[l]) v = v * p.ls[l][o] + p.lt[l][o];
                if (!last) v = fmaxf(v, 0.f);
                if (last) p.logits[b * dout + o] = v; else nxt[o] = v;
            }
        }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
    }
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// K-split plan: enough blocks to fill the chip, never an empty split
inline void split_plan(int64_t M, int N, int K, int* splits, int* chunks_per_split) {
    const int nk = (K + GK - 1) / GK;
    const int64_t tiles = std::max<int64_t>(1, ceil_div64(M, GT) * ceil_div64(N, GT));
    int64_t want = ceil_div64(SPLIT_TARGET_BLOCKS, tiles);
    if (want < 1) want = 1;
    if (want > nk) want = nk;
    const int cps = (int)std::min<int64_t>(nk, std::max<int64_t>(4, ceil_div64(nk, want)));   // >= 4 chunks: the tail re-reads S partials
    *chunks_per_split = cps;
    *splits = (int)ceil_div64(nk, cps);
}

int gemm_nt_splitk(const float* a, int64_t lda, const float* w, const float* w_hi, int n_lo, int64_t ldw, float* part, int64_t M,
                   int N, int K, int* splits_out, hipStream_t st) {
    int S, cps;
    split_plan(M, N, K, &S, &cps);
    GemmParams p{a, lda, w, w_hi, n_lo, ldw, part, (int)M, N, K, cps};
    constexpr size_t lds = sizeof(float) * 4 * GT * GLD;   // 73 728 B > the 64 KB default: raise the limit
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_nt_splitk),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    RADAD_HIP_CHECK(attr);
    hipLaunchKernelGGL(k_gemm_nt_splitk, dim3((unsigned)ceil_div64(M, GT), (unsigned)ceil_div64(N, GT), (unsigned)S),
                       dim3(G_THREADS), lds, st, p);
    RADAD_HIP_CHECK(hipGetLastError());
    *splits_out = S;
    return RADAD_OK;
}

inline size_t part_bytes(int64_t M, int N, int K) {
    int S, cps;
    split_plan(M, N, K, &S, &cps);
    return al256((size_t)S * (size_t)M * (size_t)N * sizeof(float));
}

template <typename Kern>
int raise_lds(Kern kern, size_t lds) {
    RADAD_REQUIRE(lds <= 160 * 1024, "projection tail: K*hidden does not fit the 160 KB LDS");
    if (lds > 48 * 1024) {
        const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        RADAD_HIP_CHECK(attr);
    }
    return RADAD_OK;
}

}  // namespace

extern "C" {

int64_t radad_projection_workspace_bytes(int64_t batch, int k, int dim, int hidden, int out_dim) {
    (void)out_dim;
    if (batch < 0 || k <= 0 || dim <= 0 || hidden <= 0) return -1;
    const size_t fold = al256((size_t)hidden * hidden * sizeof(float)) + al256((size_t)hidden * sizeof(float));
    return (int64_t)(part_bytes(batch * k, 2 * hidden, dim) + fold);
}

int radad_projection_fold(const radad_proj_weights* w, int dim, int hidden, float* w54t_out_dev, float* b54_out_dev, int device,
                          void* stream) {
    RADAD_REQUIRE(w && dim > 0 && hidden > 0 && hidden <= 4096, "radad_projection_fold: bad shape");
    RADAD_REQUIRE(w->w4 && w->b4 && w->w5 && w->b5 && w54t_out_dev && b54_out_dev, "radad_projection_fold: NULL buffer");
    DeviceGuard g(device);
    hipLaunchKernelGGL(k_fold_w54, dim3((unsigned)ceil_div64((int64_t)hidden * hidden, 256)), dim3(256), 0, (hipStream_t)stream,
                       w->w4, w->b4, w->w5, w->b5, dim, hidden, w54t_out_dev, b54_out_dev);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

int radad_projection_forward(const radad_proj_weights* w, const float* x_dev, int64_t batch, int k, int dim, int hidden,
                             int out_dim, float* out_dev, float* workspace_dev, int64_t workspace_bytes, int device,
                             void* stream) {
    RADAD_REQUIRE(w && batch >= 0 && k >= 1 && dim > 0 && hidden > 0 && out_dim > 0, "radad_projection_forward: bad shape");
    RADAD_REQUIRE(dim % 4 == 0, "radad_projection_forward: dim must be a multiple of 4");
    RADAD_REQUIRE(batch * (int64_t)k < (1ll << 31) - GT && hidden <= 4096, "radad_projection_forward: batch*k or hidden too large");
    if (batch == 0) return RADAD_OK;
    RADAD_REQUIRE(x_dev && out_dev && workspace_dev, "radad_projection_forward: NULL buffer");
    RADAD_REQUIRE(workspace_bytes >= radad_projection_workspace_bytes(batch, k, dim, hidden, out_dim),
                  "radad_projection_forward: workspace too small");
    DeviceGuard g(device);
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = batch * k;
    char* ws = reinterpret_cast<char*>(workspace_dev);
    float* part = (float*)ws;
    const float* w54t = w->w54t;
    const float* b54 = w->b54;
    int rc;
    if (!w54t || !b54) {   // caller did not keep the fold: rebuild it for this call
        float* f_w = (float*)(ws + part_bytes(M, 2 * hidden, dim));
        float* f_b = (float*)((char*)f_w + al256((size_t)hidden * hidden * sizeof(float)));
        if ((rc = radad_projection_fold(w, dim, hidden, f_w, f_b, device, stream))) return rc;
        w54t = f_w;
        b54 = f_b;
    }
    int S = 1;
    if ((rc = gemm_nt_splitk(x_dev, dim, w->w1, w->w3, hidden, dim, part, M, 2 * hidden, dim, &S, st))) return rc;
    TailParams tp{part, S, w->b1, w->w2, w->b2, w->b3, w54t, b54, w->ln_g, w->ln_b, w->w6, w->b6, out_dev, batch, k, hidden, out_dim};
    const size_t lds = sizeof(float) * ((size_t)2 * k * hidden + 2 * hidden + k + 8);
    if ((rc = raise_lds(k_proj_tail, lds))) return rc;
    hipLaunchKernelGGL(k_proj_tail, dim3((unsigned)batch), dim3(TAIL_THREADS), lds, st, tp);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

int64_t radad_linear_workspace_bytes(int64_t rows, int out_features, int in_features) {
    if (rows < 0 || out_features <= 0 || in_features <= 0) return -1;
    return (int64_t)part_bytes(rows, out_features, in_features);
}

int radad_linear_forward(const float* x_dev, int64_t ldx, const float* w_dev, int64_t ldw, const float* bias_dev, int act,
                         int64_t rows, int out_features, int in_features, float* out_dev, int64_t ldo, float* workspace_dev,
                         int64_t workspace_bytes, int device, void* stream) {
    RADAD_REQUIRE(rows >= 0 && out_features > 0 && in_features > 0 && rows < (1ll << 31) - GT, "radad_linear_forward: bad shape");
    RADAD_REQUIRE(act >= ACT_NONE && act <= ACT_RELU, "radad_linear_forward: act must be 0 (none), 1 (tanh) or 2 (relu)");
    RADAD_REQUIRE(in_features % 4 == 0 && ldx % 4 == 0 && ldw % 4 == 0 && ldx >= in_features && ldw >= in_features &&
                      ldo >= out_features,
                  "radad_linear_forward: in_features / row strides must be multiples of 4 and cover the row");
    if (rows == 0) return RADAD_OK;
    RADAD_REQUIRE(x_dev && w_dev && out_dev && workspace_dev, "radad_linear_forward: NULL buffer");
    RADAD_REQUIRE(workspace_bytes >= radad_linear_workspace_bytes(rows, out_features, in_features),
                  "radad_linear_forward: workspace too small");
    DeviceGuard g(device);
    hipStream_t st = (hipStream_t)stream;
    int S = 1, rc;
    if ((rc = gemm_nt_splitk(x_dev, ldx, w_dev, w_dev, out_features, ldw, workspace_dev, rows, out_features, in_features, &S, st)))
        return rc;
    const int64_t total = rows * out_features;
    hipLaunchKernelGGL(k_splitk_finish, dim3((unsigned)std::min<int64_t>(ceil_div64(total, 256), 4096)), dim3(256), 0, st,
                       workspace_dev, S, rows, out_features, bias_dev, act, out_dev, ldo);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

int64_t radad_fuse_head_workspace_bytes(int64_t batch, int dim, int proj_dim) {
    if (batch < 0 || dim <= 0 || proj_dim <= 0) return -1;
    return (int64_t)part_bytes(batch, proj_dim, dim);
}

int radad_fuse_head_forward(const radad_head_weights* w, const float* tpp_dev, const float* proj_dev, int64_t batch, int dim,
                            int proj_dim, float* fused_out_dev, float* logits_out_dev, float* workspace_dev,
                            int64_t workspace_bytes, int device, void* stream) {
    RADAD_REQUIRE(w && batch >= 0 && dim > 0 && proj_dim > 0 && batch < (1ll << 31) - GT, "radad_fuse_head_forward: bad shape");
    RADAD_REQUIRE(dim % 4 == 0 && proj_dim % 4 == 0, "radad_fuse_head_forward: dim and proj_dim must be multiples of 4");
    RADAD_REQUIRE(w->n_layers >= 0 && w->n_layers <= HEAD_MAX_LAYERS, "radad_fuse_head_forward: too many head layers");
    RADAD_REQUIRE(w->n_layers == 0 || w->dims[0] == proj_dim, "radad_fuse_head_forward: head input width != proj_dim");
    if (batch == 0) return RADAD_OK;
    RADAD_REQUIRE(w->wf && w->bf && tpp_dev && proj_dev && workspace_dev, "radad_fuse_head_forward: NULL buffer");
    RADAD_REQUIRE(w->n_layers == 0 ? fused_out_dev != nullptr : logits_out_dev != nullptr, "radad_fuse_head_forward: no output buffer");
    RADAD_REQUIRE(workspace_bytes >= radad_fuse_head_workspace_bytes(batch, dim, proj_dim),
                  "radad_fuse_head_forward: workspace too small");
    HeadParams hp{};
    int max_dim = proj_dim;
    for (int l = 0; l < w->n_layers; ++l) {
        RADAD_REQUIRE(w->lw[l] && w->lb[l] && w->dims[l] > 0 && w->dims[l + 1] > 0, "radad_fuse_head_forward: bad head layer");
        RADAD_REQUIRE((w->bn_scale[l] == nullptr) == (w->bn_shift[l] == nullptr), "radad_fuse_head_forward: bn scale/shift mismatch");
        hp.lw[l] = w->lw[l]; hp.lb[l] = w->lb[l]; hp.ls[l] = w->bn_scale[l]; hp.lt[l] = w->bn_shift[l];
        hp.dims[l] = w->dims[l];
        max_dim = std::max(max_dim, std::max(w->dims[l], w->dims[l + 1]));
    }
    hp.dims[w->n_layers] = w->n_layers ? w->dims[w->n_layers] : proj_dim;
    RADAD_REQUIRE(max_dim <= 8192, "radad_fuse_head_forward: head layer wider than 8192");
    DeviceGuard g(device);
    hipStream_t st = (hipStream_t)stream;
    int S = 1, rc;
    const int64_t ldwf = (int64_t)dim + proj_dim;
    if ((rc = gemm_nt_splitk(tpp_dev, dim, w->wf, w->wf, proj_dim, ldwf, workspace_dev, batch, proj_dim, dim, &S, st))) return rc;
    hp.part = workspace_dev; hp.S = S; hp.wf = w->wf; hp.ldwf = ldwf; hp.D = dim; hp.bf = w->bf; hp.proj = proj_dev;
    hp.n_layers = w->n_layers; hp.fused = fused_out_dev; hp.logits = logits_out_dev; hp.B = batch; hp.P = proj_dim; hp.max_dim = max_dim;
    const size_t lds = sizeof(float) * 2 * (size_t)max_dim;
    if ((rc = raise_lds(k_fuse_head, lds))) return rc;
    hipLaunchKernelGGL(k_fuse_head, dim3((unsigned)batch), dim3(TAIL_THREADS), lds, st, hp);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

}  // extern "C"
