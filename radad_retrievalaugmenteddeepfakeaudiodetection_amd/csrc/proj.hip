// proj.hip -- ProjectionLayer inference forward (projection.py:68-106) on gfx950.
//
//   s = W2 tanh(W1 x + b1) + b2          (:69-71)      a = softmax_K(s)            (:87)
//   c = W4 relu(W3 x + b3) + b4          (:74-76)      u = sum_K a c               (:88-89)
//   out = W6 LN(W5 u + b5; eps 1e-6) + b6 (:94-101)    (dropout is the identity in eval)
//
// The [B,K,D] tensor c is never built: sum_K a_k (W4 h_k + b4) = W4 (sum_K a_k h_k) + b4 because the softmax
// weights sum to one, so the only D-wide products are x W1^T / x W3^T (shared input), hbar W4^T and u W5^T.
// All four are NT GEMMs (activation rows x nn.Linear [out,in] rows, both K-contiguous) on
// v_mfma_f32_32x32x2_f32: 128x128x32 tiles through double-buffered LDS, bias + activation in the epilogue.
#include "common.h"

namespace {

constexpr int GT = 128;       // tile edge (rows of A and rows of W)
constexpr int GK = 32;
constexpr int GLD = GK + 4;
constexpr int G_THREADS = 256;

enum { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2 };

struct GemmParams {
    const float* a; int64_t lda;   // [M, K]
    const float* w; int64_t ldw;   // [N, K]  (nn.Linear weight)
    const float* bias;             // [N] or nullptr
    float* c; int64_t ldc;         // [M, N]
    int M, N, K;
};

template <int ACT>
__global__ __launch_bounds__(G_THREADS, 2) void k_gemm_nt(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sA = reinterpret_cast<float*>(smem);   // [2][GT][GLD]
    float* sB = sA + 2 * GT * GLD;                // [2][GT][GLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * GT, n0 = blockIdx.y * GT;
    const int nk = (p.K + GK - 1) / GK;
    const int ld_r = tid >> 3, ld_c = (tid & 7) * 4;
    f32x4 ra[4], rb[4];
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto gload = [&](int kc) {
        const int kcol = kc * GK + ld_c;
        const bool kin = kcol < p.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ld_r + 32 * i;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[i] = (kin && m0 + r < p.M) ? *reinterpret_cast<const f32x4*>(p.a + (int64_t)(m0 + r) * p.lda + kcol) : z;
            rb[i] = (kin && n0 + r < p.N) ? *reinterpret_cast<const f32x4*>(p.w + (int64_t)(n0 + r) * p.ldw + kcol) : z;
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
    gload(0);
    swrite(0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) gload(kc + 1);
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
        if (kc + 1 < nk) swrite(buf ^ 1);
        __syncthreads();
    }
    // acc[mt][nt][r]: row m0 + wm*64 + mt*32 + (r&3) + 8(r>>2) + 4lh, column n0 + wn*64 + nt*32 + l31
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + wn * 64 + nt * 32 + l31;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M) {
                    float v = acc[mt][nt][r] + bv;
                    if (ACT == ACT_TANH) v = tanhf(v);
                    if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
                    p.c[(int64_t)m * p.ldc + n] = v;
                }
            }
    }
}

// one wave per batch row: scores over the K neighbours, softmax, hbar = sum_k a_k * hc[b,k,:]
__global__ __launch_bounds__(256) void k_attn_mix(const float* __restrict__ ha, const float* __restrict__ hc,
                                                  const float* __restrict__ w2, const float* __restrict__ b2, int64_t B, int K,
                                                  int H, float* __restrict__ hbar) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float mx = -INFINITY;
    // scores are recomputed in the second pass instead of being kept in a runtime-indexed array
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int h = lane; h < H; h += 64) s += ha[(b * K + k) * H + h] * w2[h];
        s = wave_sum(s) + b2[0];
        mx = fmaxf(mx, s);
    }
    float den = 0.f;
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int h = lane; h < H; h += 64) s += ha[(b * K + k) * H + h] * w2[h];
        s = wave_sum(s) + b2[0];
        den += expf(s - mx);
    }
    for (int h = lane; h < H; h += 64) hbar[b * H + h] = 0.f;
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int h = lane; h < H; h += 64) s += ha[(b * K + k) * H + h] * w2[h];
        s = wave_sum(s) + b2[0];
        const float a = expf(s - mx) / den;
        for (int h = lane; h < H; h += 64) hbar[b * H + h] += a * hc[(b * K + k) * H + h];
    }
}

// one wave per row: LayerNorm(eps 1e-6), biased variance (torch.nn.LayerNorm)
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ g,
                                                   const float* __restrict__ bta, int64_t B, int H, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float s = 0.f;
    for (int h = lane; h < H; h += 64) s += x[b * H + h];
    const float mu = wave_sum(s) / (float)H;
    float v = 0.f;
    for (int h = lane; h < H; h += 64) { const float d = x[b * H + h] - mu; v += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)H + 1e-6f);
    for (int h = lane; h < H; h += 64) y[b * H + h] = (x[b * H + h] - mu) * rstd * g[h] + bta[h];
}

template <int ACT>
int gemm_nt(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias, float* c, int64_t ldc, int64_t M,
            int N, int K, hipStream_t st) {
    GemmParams p{a, lda, w, ldw, bias, c, ldc, (int)M, N, K};
    constexpr size_t lds = sizeof(float) * 4 * GT * GLD;   // 73 728 B > the 64 KB default: raise the limit
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_nt<ACT>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    RADAD_HIP_CHECK(attr);
    hipLaunchKernelGGL(k_gemm_nt<ACT>, dim3((unsigned)ceil_div64(M, GT), (unsigned)ceil_div64(N, GT)), dim3(G_THREADS), lds, st, p);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int64_t radad_projection_workspace_bytes(int64_t batch, int k, int dim, int hidden, int out_dim) {
    (void)out_dim;
    if (batch < 0 || k <= 0 || dim <= 0 || hidden <= 0) return -1;
    const size_t mh = al256((size_t)batch * k * hidden * sizeof(float));
    const size_t bh = al256((size_t)batch * hidden * sizeof(float));
    const size_t bd = al256((size_t)batch * dim * sizeof(float));
    return (int64_t)(2 * mh + 3 * bh + bd);
}

int radad_projection_forward(const radad_proj_weights* w, const float* x_dev, int64_t batch, int k, int dim, int hidden,
                             int out_dim, float* out_dev, float* workspace_dev, int64_t workspace_bytes, int device,
                             void* stream) {
    RADAD_REQUIRE(w && batch >= 0 && k >= 1 && dim > 0 && hidden > 0 && out_dim > 0, "radad_projection_forward: bad shape");
    RADAD_REQUIRE(dim % 4 == 0 && hidden % 4 == 0, "radad_projection_forward: dim and hidden must be multiples of 4");
    RADAD_REQUIRE(batch * (int64_t)k < (1ll << 31), "radad_projection_forward: batch*k too large");
    if (batch == 0) return RADAD_OK;
    RADAD_REQUIRE(x_dev && out_dev && workspace_dev, "radad_projection_forward: NULL buffer");
    RADAD_REQUIRE(workspace_bytes >= radad_projection_workspace_bytes(batch, k, dim, hidden, out_dim),
                  "radad_projection_forward: workspace too small");
    DeviceGuard g(device);
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = batch * k;
    char* ws = reinterpret_cast<char*>(workspace_dev);
    const size_t mh = al256((size_t)M * hidden * sizeof(float));
    const size_t bh = al256((size_t)batch * hidden * sizeof(float));
    float* ha = (float*)ws;                 // tanh(W1 x + b1)   [M,H]
    float* hc = (float*)(ws + mh);          // relu(W3 x + b3)   [M,H]
    float* hbar = (float*)(ws + 2 * mh);    // [B,H]
    float* y = (float*)(ws + 2 * mh + bh);  // [B,H]
    float* yn = (float*)(ws + 2 * mh + 2 * bh);
    float* u = (float*)(ws + 2 * mh + 3 * bh);   // [B,D]
    int rc;
    if ((rc = gemm_nt<ACT_TANH>(x_dev, dim, w->w1, dim, w->b1, ha, hidden, M, hidden, dim, st))) return rc;
    if ((rc = gemm_nt<ACT_RELU>(x_dev, dim, w->w3, dim, w->b3, hc, hidden, M, hidden, dim, st))) return rc;
    hipLaunchKernelGGL(k_attn_mix, dim3((unsigned)ceil_div64(batch, 4)), dim3(256), 0, st, ha, hc, w->w2, w->b2, batch, k, hidden, hbar);
    RADAD_HIP_CHECK(hipGetLastError());
    if ((rc = gemm_nt<ACT_NONE>(hbar, hidden, w->w4, hidden, w->b4, u, dim, batch, dim, hidden, st))) return rc;
    if ((rc = gemm_nt<ACT_NONE>(u, dim, w->w5, dim, w->b5, y, hidden, batch, hidden, dim, st))) return rc;
    hipLaunchKernelGGL(k_layernorm, dim3((unsigned)ceil_div64(batch, 4)), dim3(256), 0, st, y, w->ln_g, w->ln_b, batch, hidden, yn);
    RADAD_HIP_CHECK(hipGetLastError());
    return gemm_nt<ACT_NONE>(yn, hidden, w->w6, hidden, w->b6, out_dev, out_dim, batch, out_dim, hidden, st);
}

}  // extern "C"
