// embed.hip -- segment -> log-mel -> frame projection -> temporal pyramid pooling -> segment mean (gfx950).
//
// Replaces, for a whole batch and without leaving the GPU, the four nested Python loops of
// process_audio_batch (pipeline.py:392-414):
//   segmenter.py:25-39            -> the (clip, segment) -> sample-range plan built in build_plan()
//   feature_extractor.py:25-30    -> per-segment zero-mean / unit-variance       (k_logmel prologue)
//   feature_extractor.py:94-97    -> HF Whisper log-mel (hann 400 / hop 160 / 80 mels) (k_logmel)
//   feature_extractor.py:33 etc.  -> the encoder stage: a dense frame projection [T,80]x[80,F] (k_proj_pool)
//   pooling.py:66-103             -> adaptive max/avg pyramid bins               (k_proj_pool epilogue)
//   pipeline.py:411               -> mean over the clip's segments               (k_proj_pool epilogue)
//
// k_logmel  : one workgroup (7 waves) per segment.  The segment (normalised, reflect-padded) lives in LDS.
//             The 400-point real DFT is folded on the symmetric hann window into two 200-deep
//             contractions  Re = C . (x[n]+x[400-n]),  Im = S . (x[n]-x[400-n])  and run on
//             v_mfma_f32_32x32x2_f32 with the (windowed) basis as the A operand, so a lane owns one
//             frame and its 16 registers are 16 frequency bins; |X|^2 is then fed straight back as the
//             B operand of a second MFMA against the mel filter bank (no LDS round trip).
// k_proj_pool: one workgroup (8 waves) per (clip, 256 features).  W fragments stay in registers;
//             log-mel frames are staged through LDS (clamp max-8 and (x+4)/4 applied on the way in);
//             the 32-frame x 32-feature accumulators are pooled into the pyramid bins, then the
//             segment vectors are averaged.
#include "common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

namespace {

constexpr int N_FFT = 400;
constexpr int FFT_HOP = 160;
constexpr int N_BINS = 201;
constexpr int N_MELS = 80;
constexpr int LM_WAVES = 7;             // 7 frame tiles of 32 cover up to 224 frames
constexpr int LM_THREADS = LM_WAVES * 64;
constexpr int NBT = 7;                  // bin tiles of 32 (224 >= 201)
constexpr int NKK = 25;                 // K = 200 folded taps = 25 groups of 8
constexpr int KCH = 5;                  // kk groups per staged basis chunk
constexpr int NCH = NKK / KCH;
constexpr int NMT = 3;                  // mel tiles of 32 (96 >= 80)
constexpr int CHUNK_FLOATS = KCH * 2 * 64 * 4;   // 2560 floats = 10 KB
constexpr int MAX_SEG_LEN = 32000;
constexpr int SIG_FLOATS = 32800;       // (160*201+400) * 161/160 rounded up
constexpr int MAX_BINS_TOTAL = 16;      // sum(levels)

__device__ __forceinline__ int sig_pos(int c) { return c + c / FFT_HOP; }   // +1 float per 160: bank-conflict-free frame stride

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

// block-wide sum over LM_THREADS threads (scratch: >= 8 floats); result broadcast to all threads
__device__ __forceinline__ float block_sum(float v, float* scratch, int nwaves) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < nwaves; ++w) t += scratch[w];
    return t;
}

// one chunk of k_logmel_h_clip's work list (logmel_h.inc): clip, position in the clip
struct ChunkRec {
    int64_t beg;                 // absolute sample offset of the clip
    int seg0, n_seg;             // the clip's first segment in the launch's segment list, its segments
    int cidx;                    // index of the chunk inside the clip
    int avail;                   // samples of the clip its segments cover (the rest reads as zero)
};

struct LogmelParams {
    const float* wave;           // all clips, back to back
    const int64_t* seg_start;    // [S] absolute sample offset of each segment
    const int* seg_valid;        // [S] real samples in the segment (rest zero padded to seg_len)
    int seg_len;                 // L
    int normalize;
    int padded;                  // P: virtual length after zero padding (== L in self mode)
    int nf;                      // frames actually computed per segment (<= 224)
    const float* basis;          // [NBT][NKK][2][64][4]                      (k_logmel: folded fp32 basis)
    const _Float16* basis_h;     // [NBT][30 groups][4][64][8]: 25 x (cos_hi, cos_lo, sin_hi, sin_lo) x 2^11, then NMT x filter bank x 2^16 (k_logmel_h)
    const float* fbfrag;         // [NBT][NMT][16][64]
    unsigned nzmask;             // bit (bt*NMT+mt): that (bin tile, mel tile) block of the filter bank is non-zero
    float* logmel;               // [S][nf][80] log10(max(mel,1e-10))  (before the max-8 clamp)
    float* seg_max;              // [S] max over the segment (pre-initialised)
    float* norm_out;             // optional [S][L]: the normalised segment (stage parity); nullptr otherwise
    const int* n_seg_dev;        // optional: segments of the device-built plan (the grid is an upper bound; blocks beyond leave)
    int split_f0;                // k_logmel_h: frames of a segment's first workgroup (the second takes the rest; >= nf: one workgroup)
    int plane_halfs;             // k_logmel_h: halfs per LDS plane of a workgroup
    int n_seg;                   // k_logmel_h: segments of this launch (its grid is rounded up)
    // k_logmel_h_clip (frames of overlapping segments transformed once: logmel_h.inc)
    const float* seg_stats;      // [S][4] mean, inv = 1 / sqrt(var + 1e-7), inv^2, 2 log10(inv) of every segment (k_seg_stats)
    const int64_t* clip_seg;     // [clips + 1] first segment of every clip
    const ChunkRec* chunk_rec;   // [chunk_cap] what a chunk workgroup needs to know, in one record (k_build_plan)
    int prefetch_dist;           // a chunk workgroup touches the samples of chunk (own index + this) while its matrix phase runs
    int seg_hop;                 // segment hop in samples, and ...
    int seg_hop_frames;          // ... in frames (H)
    int chunk_cap;               // grid size: chunks (workgroups past the true count leave)
    int n_chunks;                // true count when the plan was sized on the host (else n_seg_dev[2])
    float fb1[2];                // filter-bank weights of bin 1 in mel bands 0 and 1
    // k_logmel_fft_clip (the same work list on the vector ALU: logmel_fft.inc)
    const float* fft_tab;        // [FC_TAB_FLOATS] lane tables (fc_build_tables)
    int fft_adv2;                // bit k1: some lane's band index rises by two after its bin k1
    int debug;                   // timing experiments only (RADAD_DEBUG_LOGMEL): 1 = skip the MFMA loop, 2 = skip the prologue
};

__global__ __launch_bounds__(LM_THREADS, 2) void k_logmel(LogmelParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sig = reinterpret_cast<float*>(smem);                 // [SIG_FLOATS]
    float* sbas = sig + SIG_FLOATS;                              // [2][CHUNK_FLOATS]
    float* scratch = sbas + 2 * CHUNK_FLOATS;                    // [16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int s = blockIdx.x;
    if (p.n_seg_dev && s >= *p.n_seg_dev) return;       // device-built plan: the grid is an upper bound on the segments
    const int L = p.seg_len;
    const float* src = p.wave + p.seg_start[s];
    const int valid = p.seg_valid[s];

    // ---- stage the segment: HBM -> registers -> (stats, normalise in registers) -> LDS, then reflect borders ----------
    // A thread holds its 18 x 4 samples in registers for the whole prologue: 16-byte loads all in flight at once (the HBM
    // latency is paid once), mean and variance (two-pass, as numpy does) reduced straight from the registers, and the
    // normalised values written to LDS exactly once -- no LDS read-modify-write passes.
    constexpr int UNR = (MAX_SEG_LEN / 4 + LM_THREADS - 1) / LM_THREADS;      // 18
    const int nv = L >> 2;                                                  // L % 160 == 0
    f32x4 raw[UNR];
    {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i4 = u * LM_THREADS + tid;
            const int i = i4 * 4;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (i4 < nv && !RADAD_DBG(p.debug, 2)) {
                if (i + 4 <= valid) t = *reinterpret_cast<const f32x4_a4*>(src + i);     // (dword alignment is all a 128-bit load needs)
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = i + e < valid ? src[i + e] : 0.f;
                }
            }
            raw[u] = t;
        }
    }
    float mean = 0.f, inv = 1.f;
    if (p.normalize) {
        float lsum = 0.f;
#pragma unroll
        for (int u = 0; u < UNR; ++u) lsum += (raw[u][0] + raw[u][1]) + (raw[u][2] + raw[u][3]);   // lanes past nv hold zeros
        mean = block_sum(lsum, scratch, LM_WAVES) / (float)L;
        float lsq = 0.f;
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (u * LM_THREADS + tid < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = raw[u][e] - mean; lsq += d * d; }
            }
        const float var = block_sum(lsq, scratch, LM_WAVES) / (float)L;
        inv = 1.0f / sqrtf(var + 1e-7f);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int i4 = u * LM_THREADS + tid;
        if (i4 < nv) {
            f32x4 v = raw[u];
            if (p.normalize) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean) * inv;
            }
            float* d = sig + sig_pos(i4 * 4 + 200);     // 4 | 200 and 4 | 160: the four samples share one 160-block
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = v[e];
            if (p.norm_out) *reinterpret_cast<f32x4*>(p.norm_out + (int64_t)s * L + i4 * 4) = v;
        }
    }
    __syncthreads();
    const int CL = FFT_HOP * (p.nf - 1) + N_FFT + 1;   // chunk samples touched: c in [0, CL)
    // left border: u = c-200 < 0 reflects to i = 200-c  (c in [0,200))
    for (int c = tid; c < 200; c += LM_THREADS) sig[sig_pos(c)] = sig[sig_pos(400 - c)];
    // right border: u >= L.  Self mode (P == L): reflect i = 2(L-1)-u.  Padded mode: zeros (the reflection
    // at u >= P is never reached by a computed frame).
    for (int c = L + 200 + tid; c < CL; c += LM_THREADS) {
        const int u = c - 200;
        float v = 0.f;
        if (p.padded == L) {
            const int i = 2 * (L - 1) - u;
            v = sig[sig_pos(i + 200)];
        }
        sig[sig_pos(c)] = v;
    }
    // first basis chunk
    for (int i = tid; i < CHUNK_FLOATS / 4; i += LM_THREADS)
        reinterpret_cast<f32x4*>(sbas)[i] = reinterpret_cast<const f32x4*>(p.basis)[i];
    __syncthreads();

    const int t = min(wave * 32 + l31, p.nf - 1);      // this lane's frame (clamped: extra lanes recompute the last one)
    // LDS position of sample n of frame t: P(160t + n) = 161t + n + n/160.  With n = 40ch + 8kq + 4lh + j + 1 the
    // quotient is constant inside a chunk (ch) except for ONE tap (n = 160: ch 3, kq 4, j 3, upper half-wave), so every
    // read below is `per-chunk base + immediate`: no per-tap address arithmetic.
    const int pb1 = 161 * t + 4 * lh;                 // + 40ch + [ch == 4]                      -> x[n]
    const int pb2 = 161 * t - 4 * lh + 400 - 40;      // - 40ch + (ch <= 1 ? 2 : 1)              -> x[400 - n]
    f32x16 zacc[NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) zacc[m][r] = 0.f;

    int step = 0;   // global chunk counter (bt*NCH + ch)
    for (int bt = 0; bt < (RADAD_DBG(p.debug, 1) ? 0 : NBT); ++bt) {
        f32x16 re, im;
#pragma unroll
        for (int r = 0; r < 16; ++r) { re[r] = 0.f; im[r] = 0.f; }
        for (int ch = 0; ch < NCH; ++ch, ++step) {
            const int buf = step & 1;
            // prefetch the next chunk (global -> registers) while this one is consumed
            f32x4 pre[2];
            const bool more = step + 1 < NBT * NCH;
            const f32x4* nsrc = reinterpret_cast<const f32x4*>(p.basis) + (int64_t)(step + 1) * (CHUNK_FLOATS / 4);
            if (more) {
                pre[0] = nsrc[tid];
                if (tid + LM_THREADS < CHUNK_FLOATS / 4) pre[1] = nsrc[tid + LM_THREADS];
            }
            const float* cb = sbas + buf * CHUNK_FLOATS;
            const float* p1 = sig + (pb1 + 40 * ch + (ch == 4 ? 1 : 0));
            const float* p2 = sig + (pb2 - 40 * ch + (ch <= 1 ? 2 : 1));
            const float* p1s = p1 + ((ch == 3) ? lh : 0);      // the one tap that crosses n = 160
#pragma unroll
            for (int kq = 0; kq < KCH; ++kq) {
                const f32x4 bc = *reinterpret_cast<const f32x4*>(cb + (kq * 2 + 0) * 256 + lane * 4);
                const f32x4 bs = *reinterpret_cast<const f32x4*>(cb + (kq * 2 + 1) * 256 + lane * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = 8 * kq + j + 1;               // compile-time
                    const float x1 = (kq == 4 && j == 3) ? p1s[o] : p1[o];
                    const float x2 = p2[40 - o];
                    re = __builtin_amdgcn_mfma_f32_32x32x2f32(bc[j], x1 + x2, re, 0, 0, 0);
                    im = __builtin_amdgcn_mfma_f32_32x32x2f32(bs[j], x1 - x2, im, 0, 0, 0);
                }
            }
            if (more) {
                f32x4* dst = reinterpret_cast<f32x4*>(sbas + (buf ^ 1) * CHUNK_FLOATS);
                dst[tid] = pre[0];
                if (tid + LM_THREADS < CHUNK_FLOATS / 4) dst[tid + LM_THREADS] = pre[1];
            }
            __syncthreads();
        }
        // |X|^2 for 16 bins of this lane's frame, then mel += fb^T . pow on the matrix core
        f32x16 pw;
#pragma unroll
        for (int r = 0; r < 16; ++r) pw[r] = re[r] * re[r] + im[r] * im[r];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            if ((p.nzmask >> (bt * NMT + mt)) & 1u) {
                const float* fb = p.fbfrag + ((bt * NMT + mt) * 16) * 64 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    zacc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[r * 64], pw[r], zacc[mt], 0, 0, 0);
            }
        }
    }

    // ---- log10, segment max, store [frame][mel] --------------------------------------------------------
    const bool fvalid = wave * 32 + l31 < p.nf;
    float lmax = -INFINITY;
    float* orow = p.logmel + ((int64_t)s * p.nf + t) * N_MELS;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m0 = 32 * mt + 8 * g + 4 * lh;       // registers 4g..4g+3 = mels m0..m0+3
            if (m0 < N_MELS) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = log10f(fmaxf(zacc[mt][4 * g + i], 1e-10f));
                    lmax = fmaxf(lmax, v[i]);
                }
                if (fvalid) *reinterpret_cast<f32x4*>(orow + m0) = v;
            }
        }
    if (!fvalid) lmax = -INFINITY;
    lmax = wave_max(lmax);
    if (lane == 0 && lmax > -INFINITY) atomic_max_float(p.seg_max + s, lmax);
}

#include "logmel_h.inc"
#include "logmel_fft.inc"

// ---- projection + pooling -----------------------------------------------------------------------------
// WAVES x 32 features per workgroup: 16 waves (all 512 features of the benchmark: each segment's log-mel is staged once
// per clip) when F > 256, 8 waves otherwise
constexpr int PP_FB = 224;           // frames staged per block pass (7 tiles)
constexpr int PP_LDH = N_MELS + 8;   // padded log-mel row in LDS, in halfs (176 B: conflict-free b128 frame reads)
constexpr int PP_ACT_SHIFT = 12;     // staged log-mel values ((x+4)/4, |v| <= ~2.5) are split after scaling by 2^12

struct ProjPoolParams {
    const float* logmel;        // [S][nf][80]
    const float* seg_max;       // [S]
    const int64_t* clip_seg;    // [B+1] segment range of each clip
    int nf;                     // frames stored per segment
    int T;                      // frames per segment seen by the pooling (>= nf; frames >= nf are silence)
    int F;
    const _Float16* wfrag_h;    // [F/32][5 k-steps][hi, lo][64][8]: W x 2^e(feature), split f16, B-fragment order
    const float* wscale;        // [F] 2^-(e(feature) + PP_ACT_SHIFT): un-scales an accumulator
    const float* bias;          // [F]
    int n_levels;
    int levels[RADAD_MAX_LEVELS];
    int pool_mode;
    float* out;                 // [B][nbins*F]      (pooled + segment mean)
    unsigned short* out_bf16;   // same, emitted as bfloat16 (round to nearest even) instead of `out` when non-null
    float* frames_out;          // optional [S][T][F]: per-frame features (extract_features protocol); no pooling
    const int* n_groups_dev;    // optional: number of valid groups (device-built plan: the grid is an upper bound)
    int debug;                  // timing experiments only (-DRADAD_DEBUG_HOOKS, RADAD_DEBUG_LOGMEL): 8 = no log-mel loads, 16 = no MFMA tiles
};

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float v) {      // plain cast: v_cvt_pk_bf16_f32, NaN stays NaN
    const __bf16 b = (__bf16)v;
    return __builtin_bit_cast(unsigned short, b);
}

template <bool FRAMES_OUT, int PP_WAVES>
__global__ __launch_bounds__(PP_WAVES * 64, PP_WAVES / 4) void k_proj_pool(ProjPoolParams p) {
    constexpr int PP_THREADS = PP_WAVES * 64;
    constexpr int PP_FEATS = PP_WAVES * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the frame projection runs on the f16 matrix pipe in split form: the staged log-mel frames are two
    // f16 planes (hi, lo of v 2^12), W comes pre-split per feature with a power-of-two column scale
    _Float16* slm_h = reinterpret_cast<_Float16*>(smem);     // [PP_FB][PP_LDH]
    _Float16* slm_l = slm_h + PP_FB * PP_LDH;                // [PP_FB][PP_LDH]
    float* spool = reinterpret_cast<float*>(slm_l + PP_FB * PP_LDH);   // [nbins][PP_FEATS]  current segment
    float* sclip = spool + MAX_BINS_TOTAL * PP_FEATS;        // [nbins][PP_FEATS]  running sum over segments

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int clip = blockIdx.x;
    if (p.n_groups_dev && clip >= *p.n_groups_dev) return;   // (uniform per workgroup: nobody is left at a barrier)
    const int fblk = blockIdx.y;                             // 256-feature block
    const bool active = (fblk * PP_WAVES + wave) * 32 < p.F; // waves past F only keep the barriers company
    const int ft = active ? fblk * PP_WAVES + wave : 0;      // this wave's 32-feature tile
    const int feat = ft * 32 + l31;
    const int fl = wave * 32 + l31;                          // feature column inside the block

    int nbins = 0;
    for (int l = 0; l < p.n_levels; ++l) nbins += p.levels[l];
    // frame range [lo, hi) of every pooling bin (pooling.py:78-83), once per workgroup: computed per tile they were two 64-bit
    // divisions per bin in front of 15 MFMAs
    __shared__ int s_lo[MAX_BINS_TOTAL], s_hi[MAX_BINS_TOTAL];
    if (tid < nbins) {
        int b0 = 0, l = 0;
        while (tid >= b0 + p.levels[l]) { b0 += p.levels[l]; ++l; }
        const int lv = p.levels[l], i = tid - b0;
        s_lo[tid] = (int)(((int64_t)i * p.T) / lv);
        s_hi[tid] = (int)(((int64_t)(i + 1) * p.T + lv - 1) / lv);
    }
    __syncthreads();

    // W fragments: lane holds W[16 st + 8 lh + j][feat] (hi and lo), st = 0..4 (K = 80)
    f16x8 wh[5], wl[5];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        wh[st] = *reinterpret_cast<const f16x8*>(p.wfrag_h + ((((int64_t)ft * 5 + st) * 2 + 0) * 64 + lane) * 8);
        wl[st] = *reinterpret_cast<const f16x8*>(p.wfrag_h + ((((int64_t)ft * 5 + st) * 2 + 1) * 64 + lane) * 8);
    }
    const float wsc = p.wscale[feat];
    const float bias = p.bias[feat];

    // spool/sclip columns are wave-private (only lanes lh == 0 of the owning wave touch column fl), so no
    // barrier is needed between a segment's pooling, its fold into the clip sum and the next segment's reset
    if (!FRAMES_OUT && lh == 0)
        for (int b = 0; b < nbins; ++b) sclip[b * PP_FEATS + fl] = 0.f;

    // clip_seg == nullptr: one group per segment (ragged batches are pooled per segment and averaged afterwards)
    const int64_t s_begin = p.clip_seg ? p.clip_seg[clip] : clip, s_end = p.clip_seg ? p.clip_seg[clip + 1] : clip + 1;
    for (int64_t s = s_begin; s < s_end; ++s) {
        const float smax = p.seg_max[s];
        const float floor_v = smax - 8.0f;
        if (!FRAMES_OUT && lh == 0) {
            const float init = p.pool_mode == RADAD_POOL_MAX ? -INFINITY : 0.f;
            for (int b = 0; b < nbins; ++b) spool[b * PP_FEATS + fl] = init;
        }
        for (int f0 = 0; f0 < p.T; f0 += PP_FB) {
            const int fcount = min(PP_FB, p.T - f0);
            __syncthreads();   // previous pass finished reading slm
            // stage frames f0..f0+fcount: clamp to max-8, (x+4)/4   (feature_extraction_whisper.py:161-162)
            for (int i = tid; i < PP_FB * (N_MELS / 4); i += PP_THREADS) {
                const int fr = i / (N_MELS / 4), c4 = i % (N_MELS / 4);
                f32x4 v = {-10.f, -10.f, -10.f, -10.f};       // log10(1e-10): a frame of pure zero padding
                const int tt = f0 + fr;
                if (fr < fcount && tt < p.nf && !RADAD_DBG(p.debug, 8))
                    v = *reinterpret_cast<const f32x4*>(p.logmel + ((int64_t)s * p.nf + tt) * N_MELS + c4 * 4);
                f16x4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sc = ldexpf((fmaxf(v[e], floor_v) + 4.0f) / 4.0f, PP_ACT_SHIFT);
                    vh[e] = (_Float16)sc;
                    vl[e] = (_Float16)(sc - (float)vh[e]);
                }
                *reinterpret_cast<f16x4*>(slm_h + fr * PP_LDH + c4 * 4) = vh;
                *reinterpret_cast<f16x4*>(slm_l + fr * PP_LDH + c4 * 4) = vl;
            }
            __syncthreads();
            const int ntile = RADAD_DBG(p.debug, 16) ? 0 : (fcount + 31) / 32;
            for (int tile = 0; tile < ntile; ++tile) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const int aoff = (tile * 32 + l31) * PP_LDH + 8 * lh;
#pragma unroll
                for (int st = 0; st < 5; ++st) {
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(slm_h + aoff + st * 16);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(slm_l + aoff + st * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[st], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[st], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[st], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaf(acc[r], wsc, bias);     // exact power-of-two un-scale, then the bias
                // acc[r] = feature `feat` of frame tfirst + (r&3) + 8(r>>2) + 4lh
                const int tfirst = f0 + tile * 32;
                if (!active) continue;
                if (FRAMES_OUT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int tt = tfirst + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (tt < p.T) p.frames_out[((int64_t)s * p.T + tt) * p.F + feat] = acc[r];
                    }
                } else {
                    const int tlast = min(tfirst + 32, p.T);   // exclusive
                    for (int b = 0; b < nbins; ++b) {
                        {
                            const int lo = s_lo[b], hi = s_hi[b];
                            if (hi <= tfirst || lo >= tlast) continue;        // wave-uniform
                            float v = p.pool_mode == RADAD_POOL_MAX ? -INFINITY : 0.f;
                            if (lo <= tfirst && tfirst + 32 <= hi) {       // whole tile inside the bin (wave-uniform)
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const float x = acc[r];
                                    if (p.pool_mode == RADAD_POOL_MAX) v = fmaxf(v, x);
                                    else v += x;
                                }
                            } else {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int tt = tfirst + (r & 3) + 8 * (r >> 2) + 4 * lh;
                                    const bool in = tt >= lo && tt < hi;       // hi <= T
                                    const float x = acc[r];
                                    if (p.pool_mode == RADAD_POOL_MAX) v = in ? fmaxf(v, x) : v;
                                    else v += in ? x : 0.f;
                                }
                            }
                            const float o = __shfl_xor(v, 32, 64);
                            if (lh == 0) {
                                float* dst = spool + b * PP_FEATS + fl;
                                if (p.pool_mode == RADAD_POOL_MAX) *dst = fmaxf(*dst, fmaxf(v, o));
                                else *dst += v + o;
                            }
                        }
                    }
                }
            }
        }
        if (!FRAMES_OUT) {
            // segment vector -> running clip sum (each wave only touches its own 32 columns)
            if (lh == 0 && active) {
                for (int b = 0; b < nbins; ++b) {
                    float v = spool[b * PP_FEATS + fl];
                    if (p.pool_mode == RADAD_POOL_AVG) v = v / (float)(s_hi[b] - s_lo[b]);
                    sclip[b * PP_FEATS + fl] += v;
                }
            }
        }
    }
    if (!FRAMES_OUT && lh == 0 && active) {
        const float nseg = (float)(s_end - s_begin);
        for (int b = 0; b < nbins; ++b) {
            const float v = sclip[b * PP_FEATS + fl] / nseg;
            const int64_t o = (int64_t)clip * nbins * p.F + (int64_t)b * p.F + feat;
            if (p.out_bf16) p.out_bf16[o] = f32_to_bf16_rne(v);
            else p.out[o] = v;
        }
    }
}

// ---- projection + pooling, software-pipelined (the path of radad_embed_forward) ---------------------------------------------------
// k_proj_pool above runs `stage a segment's log-mel rows (global loads -> clamp, scale, split -> LDS)` and `multiply, pool` back to
// back behind barriers: 0.21 ms for 0.077 ms of matrix work.  Not the loads' latency (a first pipelined form that only prefetched
// the rows into registers measured the same 0.21): the VECTOR work.  A gfx950 SIMD issues a wave64 vector instruction in 4 cycles,
// and the kernel spent 11 of them per MFMA -- converting 36 values per thread and pass (9 instructions each) with the matrix pipe
// idle behind a barrier, then scaling, biasing and pooling every accumulator element in both pooling modes at once (4 per element).
// Here: 8 waves x 64 features walk the (clip, segment, 128-frame pass) list of a persistent workgroup through TWO LDS plane
// buffers.  While the MFMAs of pass k run from one buffer, the rows of pass k + 1 (loaded a pass earlier) are converted into the
// other, a slice per tile -- vector instructions that issue between the MFMAs of the same wave -- and the rows of pass k + 2 are
// requested as soon as that conversion has freed their registers.  The conversion is 4.5 instructions per value (one fma for (x + 4) / 4 . 2^12, packed f16 conversions); max pooling is
// one v_max3_f32 per two accumulator elements, scale and bias applied to the pooled value (a monotone map commutes with max, bit
// for bit); the pooling mode is a template parameter.
constexpr int PP2_WAVES = 8;
constexpr int PP2_THREADS = PP2_WAVES * 64;
constexpr int PP2_PB = 128;                                  // frames per pass: 4 tiles
constexpr int PP2_RAW = PP2_PB * (N_MELS / 4) / PP2_THREADS; // 128-bit row pieces per thread and pass: 5
static_assert(PP2_RAW * PP2_THREADS == PP2_PB * (N_MELS / 4), "a pass is a whole number of 128-bit pieces per thread");
constexpr size_t projpool2_lds_bytes(int nt) {
    return sizeof(_Float16) * 2 * 2 * PP2_PB * PP_LDH + sizeof(float) * 2 * MAX_BINS_TOTAL * PP2_WAVES * 32 * nt;
}
__device__ __forceinline__ int64_t pp2_scalar_i64(const int64_t* ptr) {          // through the scalar cache (written by k_build_plan only)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) int64_t* ci_p;
    return *((ci_p)(uintptr_t)ptr);
#else
    return *ptr;
#endif
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct PP2Item { int clip, s, s_first, s_end, f0, valid; };      // (plain ints, copied field by field: hipcc otherwise keeps the items in scratch)

template <int NT, int MODE>
__global__ __launch_bounds__(PP2_THREADS, 2) void k_proj_pool2(ProjPoolParams p, int n_groups_host) {
    constexpr int FEATS = PP2_WAVES * 32 * NT;
    constexpr int PLANE = PP2_PB * PP_LDH;                   // halfs per plane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* planes = reinterpret_cast<_Float16*>(smem);    // [2 buffers][hi, lo][PP2_PB][PP_LDH]
    float* spool = reinterpret_cast<float*>(planes + 4 * PLANE);       // [nbins][FEATS]  current segment
    float* sclip = spool + MAX_BINS_TOTAL * FEATS;           // [nbins][FEATS]  running sum over the clip's segments
    __shared__ int s_lo[MAX_BINS_TOTAL], s_hi[MAX_BINS_TOTAL];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int fblk = blockIdx.y;
    const int n_groups = p.n_groups_dev ? min(*p.n_groups_dev, n_groups_host) : n_groups_host;
    if ((int)blockIdx.x >= n_groups) return;                 // (uniform per workgroup)

    int nbins = 0;
    for (int l = 0; l < p.n_levels; ++l) nbins += p.levels[l];
    if (tid < nbins) {                                       // frame range [lo, hi) of every pooling bin (pooling.py:78-83)
        int b0 = 0, l = 0;
        while (tid >= b0 + p.levels[l]) { b0 += p.levels[l]; ++l; }
        const int lv = p.levels[l], i = tid - b0;
        s_lo[tid] = (int)(((int64_t)i * p.T) / lv);
        s_hi[tid] = (int)(((int64_t)(i + 1) * p.T + lv - 1) / lv);
    }
    // W fragments of the wave's NT feature tiles: lane holds W[16 st + 8 lh + j][feat] (hi and lo), st = 0..4 (K = 80)
    bool active[NT];
    int feat[NT], fl[NT];
    f16x8 wh[NT][5], wl[NT][5];
    float wsc[NT], bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ftile = (fblk * PP2_WAVES + wave) * NT + nt;
        active[nt] = ftile * 32 < p.F;                       // tiles past F only keep the barriers company
        const int ft = active[nt] ? ftile : 0;
        feat[nt] = ft * 32 + l31;
        fl[nt] = (wave * NT + nt) * 32 + l31;                // feature column inside the block
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            wh[nt][st] = *reinterpret_cast<const f16x8*>(p.wfrag_h + ((((int64_t)ft * 5 + st) * 2 + 0) * 64 + lane) * 8);
            wl[nt][st] = *reinterpret_cast<const f16x8*>(p.wfrag_h + ((((int64_t)ft * 5 + st) * 2 + 1) * 64 + lane) * 8);
        }
        wsc[nt] = p.wscale[feat[nt]];
        bias[nt] = p.bias[feat[nt]];
    }

    // the walk: (group, segment, pass); a group is a clip, or one segment when clip_seg is null (ragged batches)
    auto first_of = [&](int c, PP2Item& it) {
        it.clip = c; it.f0 = 0; it.valid = c < n_groups ? 1 : 0;
        it.s = it.s_first = it.s_end = 0;
        if (it.valid) {
            if (p.clip_seg) { it.s = (int)pp2_scalar_i64(p.clip_seg + c); it.s_end = (int)pp2_scalar_i64(p.clip_seg + c + 1); }
            else { it.s = c; it.s_end = c + 1; }
            it.s_first = it.s;
        }
    };
    auto next_of = [&](const PP2Item& cur, PP2Item& nx) {
        const int clip = cur.clip, sg = cur.s, sf = cur.s_first, se = cur.s_end, f0 = cur.f0, v = cur.valid;
        if (v && f0 + PP2_PB < p.T) { nx.clip = clip; nx.s = sg; nx.s_first = sf; nx.s_end = se; nx.f0 = f0 + PP2_PB; nx.valid = 1; }
        else if (v && sg + 1 < se) { nx.clip = clip; nx.s = sg + 1; nx.s_first = sf; nx.s_end = se; nx.f0 = 0; nx.valid = 1; }
        else if (v) first_of(clip + (int)gridDim.x, nx);
        else { nx.clip = clip; nx.s = sg; nx.s_first = sf; nx.s_end = se; nx.f0 = f0; nx.valid = 0; }
    };
    // rows f0 .. f0 + 127 of the item's segment into registers, clamped into the stored frames (what lies beyond is decided when
    // converting); the segment's maximum with them
    auto issue = [&](const PP2Item& it, f32x4 (&raw)[PP2_RAW], float& smax) {
        if (!it.valid) return;
        const float* base = p.logmel + (int64_t)it.s * p.nf * N_MELS;
#pragma unroll
        for (int u = 0; u < PP2_RAW; ++u) {
            const int i = u * PP2_THREADS + tid;
            const int fr = i / (N_MELS / 4), c4 = i % (N_MELS / 4);
            const int tt = min(it.f0 + fr, p.nf - 1);
            raw[u] = *reinterpret_cast<const f32x4*>(base + (int64_t)tt * N_MELS + c4 * 4);
        }
        smax = p.seg_max[it.s];
    };
    // one 128-bit piece of a pass: clamp to max - 8, (x + 4) / 4 (feature_extraction_whisper.py:161-162) scaled by 2^12 -- one fma,
    // rounded once like (x + 4) / 4 . 2^12 --, split into the two f16 planes of buffer `buf`
    auto convert = [&](const PP2Item& it, const f32x4 (&raw)[PP2_RAW], float smax, int u, int buf) {
        if (RADAD_DBG(p.debug, 64)) { asm volatile("" : : "v"(raw[u])); return; }
        const int i = u * PP2_THREADS + tid;
        const int fr = i / (N_MELS / 4), c4 = i % (N_MELS / 4);
        const int tt = it.f0 + fr;
        f32x4 v = raw[u];
        if (!(tt < p.T && tt < p.nf)) v = f32x4{-10.f, -10.f, -10.f, -10.f};      // log10(1e-10): a frame of pure zero padding
        const float floor_v = smax - 8.0f;
        f16x4 vh, vl;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            const float s0 = fmaf(fmaxf(v[e], floor_v), 1024.f, 4096.f), s1 = fmaf(fmaxf(v[e + 1], floor_v), 1024.f, 4096.f);
            const f16x2 h2 = f16x2{(_Float16)s0, (_Float16)s1};
            const f16x2 l2 = f16x2{(_Float16)(s0 - (float)h2[0]), (_Float16)(s1 - (float)h2[1])};
            vh[e] = h2[0]; vh[e + 1] = h2[1];
            vl[e] = l2[0]; vl[e + 1] = l2[1];
        }
        _Float16* ph = planes + (size_t)buf * 2 * PLANE;
        *reinterpret_cast<f16x4*>(ph + fr * PP_LDH + c4 * 4) = vh;
        *reinterpret_cast<f16x4*>(ph + PLANE + fr * PP_LDH + c4 * 4) = vl;
    };

    // the pieces converted beside tile `tile` (wave-uniform; static register indices: a run-time index would put the rows in scratch)
    auto convert_slice = [&](const PP2Item& it, const f32x4 (&raw)[PP2_RAW], float smax, int tile, int buf) {
        switch (tile) {
            case 0: convert(it, raw, smax, 0, buf); break;
            case 1: convert(it, raw, smax, 1, buf); break;
            case 2: convert(it, raw, smax, 2, buf); break;
            default: convert(it, raw, smax, 3, buf); convert(it, raw, smax, 4, buf); break;
        }
    };
    // one pass: MFMAs of `cur` from buffer `buf`, the conversion of `nxt` (rows in rawc) into the other buffer a slice per tile
    auto pass = [&](const PP2Item& cur, int buf, const PP2Item& nxt, const f32x4 (&rawc)[PP2_RAW], float smaxc) {
        const _Float16* slm_h = planes + (size_t)buf * 2 * PLANE;
        const _Float16* slm_l = slm_h + PLANE;
        if (cur.f0 == 0 && lh == 0) {                        // (columns are wave-private: no barrier for these)
            const float init = MODE == RADAD_POOL_MAX ? -INFINITY : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                for (int b = 0; b < nbins; ++b) spool[b * FEATS + fl[nt]] = init;
                if (cur.s == cur.s_first)
                    for (int b = 0; b < nbins; ++b) sclip[b * FEATS + fl[nt]] = 0.f;
            }
        }
        const int fcount = min(PP2_PB, p.T - cur.f0);
        const int ntile = (fcount + 31) / 32;
        // pooling of one accumulator tile: acc[r] = feature feat[nt] of frame tfirst + (r&3) + 8(r>>2) + 4lh, before the scale and the bias
        auto pool_tile = [&](int nt, const f32x16& acc, int tfirst, int tlast) {
            if (!active[nt]) return;
            if (RADAD_DBG(p.debug, 32)) { asm volatile("" : : "v"(acc)); return; }
            for (int b = 0; b < nbins; ++b) {
                const int lo = s_lo[b], hi = s_hi[b];
                if (hi <= tfirst || lo >= tlast) continue;        // wave-uniform
                float v = MODE == RADAD_POOL_MAX ? -INFINITY : 0.f;
                if (lo <= tfirst && tfirst + 32 <= hi) {          // whole tile inside the bin (wave-uniform)
                    if (MODE == RADAD_POOL_MAX) {
                        // v_max3_f32 by hand: fmaxf() quiets NaNs first (one v_max x, x per operand: 28 instructions instead of 8)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(v) : "v"(acc[r]), "v"(acc[r + 1]));
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) v += fmaf(acc[r], wsc[nt], bias[nt]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int tt = tfirst + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const bool in = tt >= lo && tt < hi;      // hi <= T
                        if (MODE == RADAD_POOL_MAX) v = in ? fmaxf(v, acc[r]) : v;
                        else v += in ? fmaf(acc[r], wsc[nt], bias[nt]) : 0.f;
                    }
                }
                const float o = __shfl_xor(v, 32, 64);
                if (lh == 0) {
                    float* dst = spool + b * FEATS + fl[nt];
                    // max: the exact power-of-two un-scale and the bias are monotone, so they commute with the maximum bit for bit
                    if (MODE == RADAD_POOL_MAX) *dst = fmaxf(*dst, fmaf(fmaxf(v, o), wsc[nt], bias[nt]));
                    else *dst += v + o;
                }
            }
        };
#pragma unroll 1
        for (int tile = 0; tile < 4; ++tile) {
            if (tile < ntile) {
                const int aoff = (tile * 32 + l31) * PP_LDH + 8 * lh;
                const int tfirst = cur.f0 + tile * 32;
                const int tlast = min(tfirst + 32, p.T);     // exclusive
                // A wave issues in order: vector work placed BEHIND a block of MFMAs only starts when the last of them has been issued,
                // i.e. when the pipe is nearly drained.  So the tile's vector work sits INSIDE the MFMA chains: the conversion of the next
                // pass's rows behind the first k-step of the first feature tile, the pooling of the first feature tile behind the
                // second k-step of the second one (its accumulators are complete by then); only the last pooling is exposed.
                f32x16 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
                    for (int st = 0; st < 5; ++st) {
                        const f16x8 ah = *reinterpret_cast<const f16x8*>(slm_h + aoff + st * 16);
                        const f16x8 al = *reinterpret_cast<const f16x8*>(slm_l + aoff + st * 16);
                        if (!RADAD_DBG(p.debug, 16)) {
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[nt][st], acc[nt], 0, 0, 0);
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[nt][st], acc[nt], 0, 0, 0);
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[nt][st], acc[nt], 0, 0, 0);
                        } else asm volatile("" : "+v"(acc[nt]) : "v"(ah), "v"(al));
                        // (fences only around the inserted blocks: elsewhere hipcc keeps the fragment reads two k-steps ahead of their MFMAs)
                        if (nt == 0 && st == 0 && nxt.valid) {
                            __builtin_amdgcn_sched_barrier(0);
                            convert_slice(nxt, rawc, smaxc, tile, buf ^ 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (nt == 1 && st == 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            pool_tile(0, acc[0], tfirst, tlast);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                pool_tile(NT - 1, acc[NT - 1], tfirst, tlast);
            } else if (nxt.valid) convert_slice(nxt, rawc, smaxc, tile, buf ^ 1);      // a short pass: the slices its missing tiles would have converted
        }
        // segment done: its vector into the clip's running sum; clip done: the mean leaves (pipeline.py:411)
        if (cur.f0 + PP2_PB >= p.T && lh == 0) {
            const bool last_seg = cur.s + 1 >= cur.s_end;
            const float nseg = (float)(cur.s_end - cur.s_first);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (!active[nt]) continue;
                for (int b = 0; b < nbins; ++b) {
                    float v = spool[b * FEATS + fl[nt]];
                    if (MODE == RADAD_POOL_AVG) v = v / (float)(s_hi[b] - s_lo[b]);
                    v += sclip[b * FEATS + fl[nt]];
                    if (!last_seg) sclip[b * FEATS + fl[nt]] = v;
                    else {
                        v = v / nseg;
                        const int64_t o = (int64_t)cur.clip * nbins * p.F + (int64_t)b * p.F + feat[nt];
                        if (p.out_bf16) p.out_bf16[o] = f32_to_bf16_rne(v);
                        else p.out[o] = v;
                    }
                }
            }
        }
    };

    // Three passes in the pipe: it0 multiplies from buffer 0 / 1, it1 (its rows loaded a whole pass ago) is converted into the other
    // buffer, it2 loads.  Two register sets take turns (static indices: the loop body is written out for both roles).  A single set
    // -- request the rows as soon as the conversion has freed the registers -- was measured first: every pass then started by
    // waiting ~1.5 us for rows requested ~0.1 us earlier (46 % of the wave cycles in s_waitcnt: 0.19 ms, no faster than k_proj_pool).
    PP2Item it0, it1, it2;
    first_of((int)blockIdx.x, it0);
    next_of(it0, it1);
    next_of(it1, it2);
    f32x4 rawA[PP2_RAW], rawB[PP2_RAW];
    float smaxA = 0.f, smaxB = 0.f;
    issue(it0, rawA, smaxA);
    issue(it1, rawB, smaxB);
#pragma unroll
    for (int u = 0; u < PP2_RAW; ++u) convert(it0, rawA, smaxA, u, 0);
    issue(it2, rawA, smaxA);                                 // (rawA is free again)
    __syncthreads();
    auto copy_item = [](PP2Item& d, const PP2Item& x) { d.clip = x.clip; d.s = x.s; d.s_first = x.s_first; d.s_end = x.s_end; d.f0 = x.f0; d.valid = x.valid; };
    while (it0.valid) {
        pass(it0, 0, it1, rawB, smaxB);                      // it1: rawB -> buffer 1; it2 is loading into rawA
        PP2Item it3;
        next_of(it2, it3);
        issue(it3, rawB, smaxB);                             // (rawB is free again)
        __syncthreads();
        if (!it1.valid) break;
        pass(it1, 1, it2, rawA, smaxA);                      // it2: rawA -> buffer 0; it3 is loading into rawB
        PP2Item it4;
        next_of(it3, it4);
        issue(it4, rawA, smaxA);
        __syncthreads();
        copy_item(it0, it2); copy_item(it1, it3); copy_item(it2, it4);
    }
}

// finalise log-mel for the stage API: out[s][t][m] = (max(x, smax-8)+4)/4, frames >= nf are silence
__global__ void k_logmel_finalize(const float* __restrict__ logmel, const float* __restrict__ seg_max, int nf, int T,
                                  int64_t n_seg, float* __restrict__ out) {
    const int64_t total = n_seg * T * N_MELS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i % N_MELS);
        const int64_t st = i / N_MELS;
        const int t = (int)(st % T);
        const int64_t s = st / T;
        const float x = t < nf ? logmel[(s * nf + t) * N_MELS + m] : -10.f;
        out[i] = (fmaxf(x, seg_max[s] - 8.0f) + 4.0f) / 4.0f;
    }
}

__global__ void k_fill(float* p, float v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---- stand-alone temporal pyramid pooling (pool_features / pool_features_batch) --------------------------
// grid (items, ceil(F/256)); thread = one feature column; walks the T rows once per bin
__global__ __launch_bounds__(256) void k_tpp(const float* __restrict__ feats, const int64_t* __restrict__ row_off, int F,
                                             int n_levels, const int* __restrict__ levels_dev, int pool_mode, int nbins,
                                             float* __restrict__ out) {
    const int item = blockIdx.x;
    const int f = blockIdx.y * 256 + threadIdx.x;
    if (f >= F) return;
    const int64_t r0 = row_off[item];
    const int T = (int)(row_off[item + 1] - r0);
    const float* base = feats + r0 * F + f;
    int b = 0;
    for (int l = 0; l < n_levels; ++l) {
        const int lv = levels_dev[l];
        for (int i = 0; i < lv; ++i, ++b) {
            const int lo = (int)(((int64_t)i * T) / lv);
            const int hi = (int)(((int64_t)(i + 1) * T + lv - 1) / lv);
            float v;
            if (pool_mode == RADAD_POOL_MAX) {
                v = -INFINITY;
                for (int t = lo; t < hi; ++t) v = fmaxf(v, base[(int64_t)t * F]);
            } else {
                v = 0.f;
                for (int t = lo; t < hi; ++t) v += base[(int64_t)t * F];
                v = v / (float)(hi - lo);
            }
            out[((int64_t)item * nbins + b) * F + f] = v;
        }
    }
}

__global__ __launch_bounds__(256) void k_group_mean(const float* __restrict__ in, const int64_t* __restrict__ goff, int dim,
                                                    float* __restrict__ out, unsigned short* __restrict__ out_bf16) {
    const int g = blockIdx.x;
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= dim) return;
    const int64_t a = goff[g], b = goff[g + 1];
    float v = 0.f;
    for (int64_t r = a; r < b; ++r) v += in[r * dim + c];
    v = v / (float)(b - a);
    if (out_bf16) out_bf16[(int64_t)g * dim + c] = f32_to_bf16_rne(v);
    else out[(int64_t)g * dim + c] = v;
}

// ---- the segment plan on the device (segmenter.py:25-39) --------------------------------------------------------------
// clip b = samples [off[b], off[b+1]);  n_seg = max(1, (N - L) // hop + 1) (Python floor division);  segment i starts at
// off[b] + i hop and holds min(L, N - i hop) real samples (zero padded to L only when N < L; the tail past the last full
// window is dropped).  One workgroup: chunks of 1024 clips, an exclusive scan of n_seg with a running carry, then every
// thread writes its own clip's segments.  Outputs: clip_seg [B+1], seg_start [S], seg_valid [S], n_seg_out[0] = S.
// Nothing here needs the host: a batch with fresh offsets costs one tiny launch, no synchronisation.
// The offsets are DEVICE data nobody has looked at: every clip is clamped into [0, total) (total = samples in the wave buffer, or
// -1: unknown, only negative lengths are repaired) and the segment count to seg_cap, so that whatever the tensor holds the kernels
// behind this one read inside the wave buffer and inside their own scratch.  n_seg_out[1] reports what had to be repaired:
// bit 0 an offset outside [0, total], bit 1 a clip ending before it starts, bit 2 more segments than seg_cap.
// Chunk plan of k_logmel_h_clip / k_logmel_fft_clip (share_T > 0): a clip of S segments is cut into chunk_count_g(geom, S, T, H)
// chunks (logmel_h.inc, logmel_fft.inc); chunk_rec[] holds one record per chunk, clip after clip, n_seg_out[2] counts them.
__global__ __launch_bounds__(1024) void k_build_plan(const int64_t* __restrict__ off, int64_t n_clips, int L, int hop,
                                                     int64_t seg_cap, int64_t total, int64_t* __restrict__ clip_seg,
                                                     int64_t* __restrict__ seg_start, int* __restrict__ seg_valid, int* __restrict__ n_seg_out,
                                                     int share_T, int share_H, int64_t chunk_cap,
                                                     ChunkRec* __restrict__ chunk_rec, int geom) {
    __shared__ long long s_wave[16];
    __shared__ long long s_carry, s_carry_c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_carry = 0; s_carry_c = 0; }
    __syncthreads();
    int bad = 0;
    for (int64_t b0 = 0; b0 < n_clips; b0 += 1024) {
        const int64_t b = b0 + tid;
        long long n = 0, ns = 0, beg = 0;
        if (b < n_clips) {
            long long end = off[b + 1];
            beg = off[b];
            if (total >= 0) {
                const long long cb = beg < 0 ? 0 : (beg > total ? total : beg), ce = end < 0 ? 0 : (end > total ? total : end);
                if (cb != beg || ce != end) bad |= 1;
                beg = cb; end = ce;
            }
            n = end - beg;
            if (n < 0) { n = 0; bad |= 2; }
            const long long d = n - L;
            long long q = d / hop;
            if ((d % hop != 0) && (d < 0)) --q;          // floor division
            ns = q + 1 > 1 ? q + 1 : 1;
        }
        long long inc = ns;                              // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        long long base = s_carry;
        for (int w = 0; w < wave; ++w) base += s_wave[w];
        const long long first = base + inc - ns;         // exclusive prefix of this clip
        if (b < n_clips) {
            clip_seg[b] = first < seg_cap ? first : seg_cap;
            for (long long i = 0; i < ns; ++i) {
                if (first + i < seg_cap) {
                    seg_start[first + i] = beg + i * hop;
                    const long long v = n - i * hop;
                    seg_valid[first + i] = (int)(v < 0 ? 0 : (v > L ? L : v));
                }
            }
        }
        __syncthreads();
        if (tid == 1023) s_carry = base + inc;
        __syncthreads();
        if (share_T > 0) {                               // the same scan over the clips' chunk counts
            const long long first_seg = first < seg_cap ? first : seg_cap;
            const long long S = (b < n_clips) ? ((first + ns < seg_cap ? first + ns : seg_cap) - first_seg) : 0;     // segments that made it into the plan
            const long long nc = chunk_count_g(geom, (int)S, share_T, share_H);
            long long incc = nc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const long long t = __shfl_up(incc, o, 64);
                if (lane >= o) incc += t;
            }
            if (lane == 63) s_wave[wave] = incc;
            __syncthreads();
            long long basec = s_carry_c;
            for (int w = 0; w < wave; ++w) basec += s_wave[w];
            const long long firstc = basec + incc - nc;
            if (b < n_clips) {
                const long long last_valid = n - (S - 1) * hop;                      // samples of the clip's last segment
                ChunkRec r;
                r.beg = beg; r.seg0 = (int)first_seg; r.n_seg = (int)S;
                r.avail = (int)((S - 1) * hop + (last_valid < 0 ? 0 : (last_valid > L ? L : last_valid)));
                for (long long i = 0; i < nc; ++i)
                    if (firstc + i < chunk_cap) { r.cidx = (int)i; chunk_rec[firstc + i] = r; }
            }
            __syncthreads();
            if (tid == 1023) s_carry_c = basec + incc;
            __syncthreads();
        }
    }
    bad = (__syncthreads_or(bad & 1) ? 1 : 0) | (__syncthreads_or(bad & 2) ? 2 : 0);      // (a predicate per flag: the builtin ORs truth values)
    if (tid == 0) {
        clip_seg[n_clips] = s_carry < seg_cap ? s_carry : seg_cap;
        n_seg_out[0] = (int)(s_carry < seg_cap ? s_carry : seg_cap);
        n_seg_out[1] = bad | (s_carry > seg_cap ? 4 : 0);
        if (share_T > 0) n_seg_out[2] = (int)(s_carry_c < chunk_cap ? s_carry_c : chunk_cap);
    }
}

constexpr size_t logmel_lds_bytes() { return sizeof(float) * (SIG_FLOATS + 2 * CHUNK_FLOATS + 16); }
constexpr size_t projpool_lds_bytes(int waves) {
    return sizeof(_Float16) * 2 * PP_FB * PP_LDH + sizeof(float) * 2 * MAX_BINS_TOTAL * waves * 32;
}

// growable device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return RADAD_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t cap = need + need / 4;
        if (hipMalloc(&p, cap) != hipSuccess) { radad_set_error("hipMalloc of %zu bytes failed", cap); return RADAD_ENOMEM; }
        bytes = cap;
        return RADAD_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};


// Lane tables of k_logmel_fft_clip (layout: logmel_fft.inc; data flow checked in numpy by tools/fft_design.py and, driven by THESE
// tables, by tests/test_fft_tables.py).  Lane pl of an octet holds the residue class rr = pl < 4 ? pl : 11 - pl of the packed
// frame and ends with the bin block k2 = bitrev3(pl).  Returns false when the filter bank does not have the shape the sparse mel
// step needs (then the matrix-pipe kernels stay): every bin feeds at most two ADJACENT bands (b, b + 1), b never falls and rises
// by at most two from bin to bin (the Slaney bank's linear part has bands 37.2 Hz apart under bins 40 Hz apart: now and then b
// skips one), no band is fed by more than two lanes, bin 200 carries no weight, the band slots fit a row.  *adv2_mask: bit k1 is
// set when some lane's band index rises by two after its bin k1 (the kernel then stores both running sums at that step).
static bool fc_build_tables(const float* fb /* [201][80] */, float* tab /* [FC_TAB_FLOATS] */, int* adv2_mask) {
    const double PI = 3.14159265358979323846;
    for (int i = 0; i < FC_TAB_FLOATS; ++i) tab[i] = 0.f;
    *adv2_mask = 0;
    int bnd[200];                                                    // first band of every bin
    for (int k = 0; k < 200; ++k) {
        int first = -1, last = -1;
        for (int b = 0; b < N_MELS; ++b)
            if (fb[k * N_MELS + b] != 0.f) { if (first < 0) first = b; last = b; }
        if (first >= 0 && last - first > 1) return false;
        const int lo_ok = last < 0 ? 0 : std::max(0, last - 1), hi_ok = first < 0 ? N_MELS - 2 : std::min(first, N_MELS - 2);
        if (lo_ok > hi_ok) return false;
        if (k == 0) bnd[k] = lo_ok;
        else {
            if (bnd[k - 1] > hi_ok || lo_ok > bnd[k - 1] + 2) return false;       // the band index fell, or jumped by three
            bnd[k] = std::max(bnd[k - 1], lo_ok);
        }
    }
    for (int b = 0; b < N_MELS; ++b)
        if (fb[(N_BINS - 1) * N_MELS + b] != 0.f) return false;       // bin 200 (= Re Z[0] - Im Z[0]) is not computed
    int slot_base[8], n_slots[8], total = 0;                         // per bin block k2: its band slots in a row of the mel tile
    for (int k2 = 0; k2 < 8; ++k2) {
        slot_base[k2] = total;
        n_slots[k2] = bnd[25 * k2 + 24] - bnd[25 * k2] + 2;
        total += n_slots[k2];
    }
    if (total > FC_ROW_X1) return false;
    int* comb = reinterpret_cast<int*>(tab + FC_COMB);
    for (int m = 0; m < N_MELS; ++m) {
        int n = 0;
        comb[2 * m] = comb[2 * m + 1] = 4 * FC_ROW_ZERO;
        for (int k2 = 0; k2 < 8; ++k2) {
            const int j = m - bnd[25 * k2];
            if (j >= 0 && j < n_slots[k2]) {
                if (n == 2) return false;
                comb[2 * m + n++] = 4 * (slot_base[k2] + j);
            }
        }
    }
    for (int pl = 0; pl < 8; ++pl) {
        const int rr = pl < 4 ? pl : 11 - pl;
        const int k2 = ((pl & 1) << 2) | (pl & 2) | ((pl >> 2) & 1);
        for (int k1 = 0; k1 < 25; ++k1) {
            float* t = tab + k1 * FC_TABK;
            for (int e = 0; e < 2; ++e) {                                       // periodic hann at n = 16 m + 2 rr + e (m = k1's slot)
                const int n = 16 * k1 + 2 * rr + e;
                t[2 * pl + e] = (float)(0.5 - 0.5 * cos(2.0 * PI * n / N_FFT));
            }
            const int e200 = (rr * k1) % 200;                                   // W200^(rr k1) = cos - i sin
            t[16 + 2 * pl] = (float)cos(2.0 * PI * e200 / 200.0);
            t[16 + 2 * pl + 1] = (float)sin(2.0 * PI * e200 / 200.0);
            const int k = 25 * k2 + k1;                                         // the bin this lane ends with at k1
            t[32 + 2 * pl] = (float)cos(2.0 * PI * k / 400.0);
            t[32 + 2 * pl + 1] = (float)sin(2.0 * PI * k / 400.0);
            const int b = bnd[k];
            t[48 + 4 * pl] = 0.25f * fb[k * N_MELS + b];                        // the kernel forms 4 |X|^2
            t[48 + 4 * pl + 1] = 0.25f * fb[k * N_MELS + b + 1];
            const int soff = 4 * (slot_base[k2] + b - bnd[25 * k2]);            // byte offset of band b's slot in the row
            const int adv = k1 == 24 ? 1 : bnd[k + 1] - b;                      // how far the running sums shift after this bin (0, 1, 2)
            if (adv == 2) *adv2_mask |= 1 << k1;
            memcpy(&t[48 + 4 * pl + 2], &soff, sizeof(int));
            memcpy(&t[48 + 4 * pl + 3], &adv, sizeof(int));
        }
        // exchange stages: own <- (own + g partner)(c + i s).  Stage t pairs index i with i +- half inside the lane's branch; the lane
        // whose bit (2, 1, 0) of pl is clear keeps the sum, the other (x_small - x_big) W8^e.  A stage without multiply (the third) cannot
        // flip a sign, so lanes 1 and 3 -- which hold the larger index there and need partner - own -- get their stage-2 value negated.
        float* net = tab + FC_NET + 8 * pl;
        int idx = rr;
        for (int stage = 0; stage < 3; ++stage) {
            const int half = 4 >> stage, bit = 2 - stage;
            const bool hi = (pl >> bit) & 1, small = idx < half;
            const int e = (idx % half) * (4 / half);                            // twiddle exponent of W8
            double g = 1.0, c = 1.0, sn = 0.0;
            if (hi) {
                g = -1.0;
                const double sgn = small ? 1.0 : -1.0;
                c = sgn * cos(2.0 * PI * e / 8.0);
                sn = -sgn * sin(2.0 * PI * e / 8.0);
            }
            if (stage == 1 && (pl == 1 || pl == 3)) { c = -c; sn = -sn; }       // stored negated for stage 3
            if (stage == 2) {
                // inputs of lanes 1, 3 carry a minus sign: lane p < 4 odd wants partner - own = own' + partner; its partner own - own'
                if (pl < 4) g = (pl & 1) ? 1.0 : -1.0;
                else g = (pl & 1) ? -1.0 : 1.0;
                net[6] = (float)g;
            } else {
                net[3 * stage] = (float)g;
                net[3 * stage + 1] = (float)(fabs(c) < 1e-15 ? 0.0 : c);
                net[3 * stage + 2] = (float)(fabs(sn) < 1e-15 ? 0.0 : sn);
            }
            idx %= half;
        }
    }
    return true;
}

}  // namespace

struct radad_embed_s {
    radad_embed_cfg cfg;
    int device = 0;
    int nbins = 0, T = 0, nf = 0, padded = 0;
    unsigned nzmask = 0;
    float *basis = nullptr, *fbfrag = nullptr, *wscale = nullptr, *bias = nullptr;
    _Float16* wfrag_h = nullptr;
    _Float16* basis_h = nullptr;
    int logmel_f32 = 0;                  // RADAD_LOGMEL_F32=1: the fp32-MFMA kernel (k_logmel) instead of k_logmel_h
    // k_logmel_h_clip: frames shared by overlapping segments are transformed once (logmel_h.inc).  Possible when the segment hop is a
    // multiple of the frame hop and smaller than the segment, at most LH_MAX_OWNERS segments share a frame, and the filter bank
    // has the Slaney shape the mean correction relies on (bin 0 without weight, bin 1 only in bands 0 and 1)
    int share_frames = 0;                // 1: batches given as clips take k_logmel_h_clip (RADAD_LOGMEL_SHARED=0 turns it off)
    int share_H = 0;                     // segment hop in frames
    float fb1[2] = {0.f, 0.f};
    // k_logmel_fft_clip: the shared-frame work list as a radix FFT on the vector ALU (logmel_fft.inc).  Needs what share_frames needs
    // and a filter bank whose bins feed at most two adjacent bands (triangular filters); RADAD_EMBED_DFT_GEMM keeps k_logmel_h_clip
    int logmel_fft = 0;
    float* fft_tab = nullptr;
    int fft_adv2 = 0;                    // steps at which some lane's mel band index rises by two (fc_build_tables)
    int64_t plan_nchunks = 0;            // interior chunks of the cached plan (host-sized plans), or their upper bound (device-sized)
    bool plan_has_chunks = false;        // the cached plan carries the chunk arrays
    DevBuf seg_stats, chunk_rec;
    int last_logmel_kind = 0;            // 0 per segment, 1 shared clip frames (radad_embed_last_logmel_kind)
    int n_cus = 256;                     // compute units of the device (two chunk workgroups run on each)
    int* levels_dev = nullptr;
    // plan cache + scratch
    std::vector<int64_t> plan_key;       // the clip_offsets the cached plan was built from
    int64_t plan_nseg = 0;
    bool plan_uniform = true;            // every clip of the cached plan has the same number of segments
    DevBuf seg_start, seg_valid, clip_seg, seg_pool, logmel, seg_max, misc, clip_off, n_seg_dev;
    bool plan_on_device = false;         // the cached plan came from device-resident offsets: segment count only known there
    // two pinned staging buffers for clip offsets handed over in host memory (no stream synchronisation on the way)
    int64_t* pin[2] = {nullptr, nullptr};
    size_t pin_cap[2] = {0, 0};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    int pin_next = 0;
    // what k_build_plan repaired in the device-offset batches: a ring of pinned words, one per batch in flight, so that a caller can
    // ask without waiting (radad_embed_plan_flags_poll) and no batch's report is overwritten by the next one's
    static constexpr int PLAN_RING = 4;
    int* plan_flags_host = nullptr;      // [PLAN_RING] pinned
    hipEvent_t plan_flags_ev[PLAN_RING] = {nullptr, nullptr, nullptr, nullptr};
    bool plan_flags_pending[PLAN_RING] = {false, false, false, false};
    int plan_flags_next = 0, plan_flags_accum = 0;
    EventRing prof_logmel, prof_pool;
    std::mutex mu;
};

extern "C" int64_t radad_segment_count(int64_t n_samples, int32_t segment_length, int32_t hop_length) {
    if (segment_length <= 0 || hop_length <= 0 || n_samples < 0) return -1;
    // Python floor division (segmenter.py:25): (N-L)//hop is negative for N < L, max(1, .) then yields 1
    const int64_t d = n_samples - segment_length;
    int64_t q = d / hop_length;
    if ((d % hop_length != 0) && (d < 0)) --q;
    return std::max<int64_t>(1, q + 1);
}

static int upload(DevBuf& b, const void* host, size_t bytes, hipStream_t st) {
    int rc = b.ensure(bytes ? bytes : 1);
    if (rc) return rc;
    if (bytes) RADAD_HIP_CHECK(hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, st));
    return RADAD_OK;
}

// Launch k_build_plan over device-resident offsets.  seg_cap bounds the number of segments (buffers and grids).
// chunk_cap: interior chunks of k_logmel_h_clip to plan for (0: no chunk plan).
static int plan_on_device(radad_embed_t h, const int64_t* clip_off_dev, int64_t n_clips, int64_t seg_cap, int64_t total, hipStream_t st,
                          int64_t chunk_cap = 0) {
    int rc;
    if ((rc = h->seg_start.ensure((size_t)seg_cap * sizeof(int64_t)))) return rc;
    if ((rc = h->seg_valid.ensure((size_t)seg_cap * sizeof(int32_t)))) return rc;
    if ((rc = h->clip_seg.ensure((size_t)(n_clips + 1) * sizeof(int64_t)))) return rc;
    if ((rc = h->n_seg_dev.ensure(4 * sizeof(int)))) return rc;
    if (chunk_cap > 0 && (rc = h->chunk_rec.ensure((size_t)chunk_cap * sizeof(ChunkRec)))) return rc;
    hipLaunchKernelGGL(k_build_plan, dim3(1), dim3(1024), 0, st, clip_off_dev, n_clips, h->cfg.segment_length, h->cfg.hop_length,
                       seg_cap, total, (int64_t*)h->clip_seg.p, (int64_t*)h->seg_start.p, (int*)h->seg_valid.p, (int*)h->n_seg_dev.p,
                       chunk_cap > 0 ? h->nf : 0, h->share_H, chunk_cap, (ChunkRec*)h->chunk_rec.p, h->logmel_fft ? 1 : 0);
    RADAD_HIP_CHECK(hipGetLastError());
    h->plan_has_chunks = chunk_cap > 0;
    h->plan_nchunks = chunk_cap;
    return RADAD_OK;
}
static int64_t clip_chunks(const radad_embed_s* h, int64_t n_segments_of_clip) {       // chunks of one clip
    return chunk_count_g(h->logmel_fft ? 1 : 0, (int)n_segments_of_clip, h->nf, h->share_H);
}

// Plan for clip offsets in HOST memory (segmenter.py:25-39): the host only counts the segments (it needs the grid size);
// the offsets travel through a pinned staging buffer and the plan itself is built by k_build_plan -- no stream
// synchronisation, no pageable copies.  The plan of the previous call is reused when the offsets are the same.
static int build_plan(radad_embed_t h, const int64_t* clip_offsets, int64_t n_clips, hipStream_t st) {
    if (!h->plan_on_device && (int64_t)h->plan_key.size() == n_clips + 1 &&
        memcmp(h->plan_key.data(), clip_offsets, sizeof(int64_t) * (n_clips + 1)) == 0)
        return RADAD_OK;
    const int L = h->cfg.segment_length, hop = h->cfg.hop_length;
    int64_t n_seg = 0, first = -1, n_chunks = 0;
    bool uniform = true;
    for (int64_t b = 0; b < n_clips; ++b) {
        const int64_t n = clip_offsets[b + 1] - clip_offsets[b];
        RADAD_REQUIRE(n >= 0, "radad_embed_forward: clip_offsets must be non-decreasing");
        const int64_t ns = radad_segment_count(n, L, hop);
        if (first < 0) first = ns;
        uniform = uniform && ns == first;
        n_seg += ns;
        if (h->share_frames) n_chunks += clip_chunks(h, ns);
    }
    RADAD_REQUIRE(n_seg < (1ll << 31), "radad_embed_forward: too many segments in one batch");
    const int slot = h->pin_next;
    const size_t need = (size_t)(n_clips + 1) * sizeof(int64_t);
    if (!h->pin_ev[slot]) RADAD_HIP_CHECK(hipEventCreateWithFlags(&h->pin_ev[slot], hipEventDisableTiming));
    else RADAD_HIP_CHECK(hipEventSynchronize(h->pin_ev[slot]));       // the copy that last used this buffer (two calls ago) is done
    if (h->pin_cap[slot] < need) {
        if (h->pin[slot]) (void)hipHostFree(h->pin[slot]);
        h->pin[slot] = nullptr; h->pin_cap[slot] = 0;
        RADAD_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->pin[slot]), need + need / 2, hipHostMallocDefault));
        h->pin_cap[slot] = need + need / 2;
    }
    memcpy(h->pin[slot], clip_offsets, need);
    int rc;
    if ((rc = h->clip_off.ensure(need))) return rc;
    RADAD_HIP_CHECK(hipMemcpyAsync(h->clip_off.p, h->pin[slot], need, hipMemcpyHostToDevice, st));
    RADAD_HIP_CHECK(hipEventRecord(h->pin_ev[slot], st));
    h->pin_next ^= 1;
    if ((rc = plan_on_device(h, (const int64_t*)h->clip_off.p, n_clips, std::max<int64_t>(n_seg, 1), -1, st, n_chunks))) return rc;
    h->plan_uniform = uniform;
    h->plan_on_device = false;
    h->plan_key.assign(clip_offsets, clip_offsets + n_clips + 1);
    h->plan_nseg = n_seg;
    return RADAD_OK;
}

static int upload_plan_explicit(radad_embed_t h, const int64_t* seg_start, const int32_t* seg_valid, int64_t n_seg, hipStream_t st) {
    for (int64_t i = 0; i < n_seg; ++i)
        RADAD_REQUIRE(seg_valid[i] >= 0 && seg_valid[i] <= h->cfg.segment_length && seg_start[i] >= 0,
                      "segment %lld: bad start/valid", (long long)i);
    h->plan_key.clear();
    h->plan_on_device = false;
    h->plan_has_chunks = false;
    RADAD_HIP_CHECK(hipStreamSynchronize(st));
    int rc;
    if ((rc = upload(h->seg_start, seg_start, n_seg * sizeof(int64_t), st))) return rc;
    if ((rc = upload(h->seg_valid, seg_valid, n_seg * sizeof(int32_t), st))) return rc;
    std::vector<int64_t> cs(n_seg + 1);
    for (int64_t i = 0; i <= n_seg; ++i) cs[i] = i;            // one "clip" per segment
    if ((rc = upload(h->clip_seg, cs.data(), cs.size() * sizeof(int64_t), st))) return rc;
    RADAD_HIP_CHECK(hipStreamSynchronize(st));
    h->plan_nseg = n_seg;
    return RADAD_OK;
}

static int launch_logmel(radad_embed_t h, const float* wave_dev, int64_t n_seg, float* norm_out, hipStream_t st,
                         const int* n_seg_dev = nullptr) {
    int rc;
    if ((rc = h->logmel.ensure((size_t)n_seg * h->nf * N_MELS * sizeof(float)))) return rc;
    if ((rc = h->seg_max.ensure((size_t)n_seg * sizeof(float)))) return rc;
    const float init = h->T > h->nf ? -10.f : -INFINITY;   // padded mode: silence frames take part in the max
    const bool clip_frames = h->share_frames && h->plan_has_chunks && !h->logmel_f32 && !norm_out;
    if (!(clip_frames && h->cfg.normalize))                // (k_seg_stats starts the segment maxima itself)
        hipLaunchKernelGGL(k_fill, dim3((unsigned)std::min<int64_t>(1024, ceil_div64(n_seg, 256))), dim3(256), 0, st,
                           (float*)h->seg_max.p, init, n_seg);
    LogmelParams p;
    p.wave = wave_dev; p.seg_start = (const int64_t*)h->seg_start.p; p.seg_valid = (const int*)h->seg_valid.p;
    p.seg_len = h->cfg.segment_length; p.normalize = h->cfg.normalize; p.padded = h->padded; p.nf = h->nf;
    p.basis = h->basis; p.basis_h = h->basis_h; p.fbfrag = h->fbfrag; p.nzmask = h->nzmask; p.logmel = (float*)h->logmel.p;
    p.seg_max = (float*)h->seg_max.p; p.norm_out = norm_out; p.n_seg_dev = n_seg_dev;
    p.seg_stats = nullptr; p.clip_seg = nullptr; p.chunk_rec = nullptr; p.prefetch_dist = 0;
    p.seg_hop = 0; p.seg_hop_frames = 0; p.chunk_cap = 0; p.n_chunks = 0; p.fb1[0] = p.fb1[1] = 0.f;
    p.fft_tab = nullptr; p.fft_adv2 = 0;
    p.debug = 0;
#ifdef RADAD_DEBUG_HOOKS        // timing experiments only (tools/exp_logmel.sh builds with -DRADAD_DEBUG_HOOKS); never in the shipped library
    { const char* dbg = getenv("RADAD_DEBUG_LOGMEL"); p.debug = dbg ? atoi(dbg) : 0; }
#endif
    h->prof_logmel.begin(st);
    p.split_f0 = lh_split_f0(h->nf);
    p.n_seg = (int)n_seg;
    const int wg_frames = std::max(p.split_f0, h->nf - p.split_f0);
    p.plane_halfs = lh_plane_halfs(wg_frames);
    h->last_logmel_kind = clip_frames ? (h->logmel_fft ? 2 : 1) : 0;
    if (clip_frames) {
        // the segments' statistics first (the transform is shared, the normalisation is not), then chunks and edge frames in one grid
        if ((rc = h->seg_stats.ensure((size_t)n_seg * 4 * sizeof(float)))) return rc;
        if (h->cfg.normalize)
            hipLaunchKernelGGL(k_seg_stats, dim3((unsigned)n_seg), dim3(LH_THREADS), 0, st, wave_dev, p.seg_start, p.seg_valid, p.seg_len,
                               (int)n_seg, n_seg_dev, (float*)h->seg_stats.p, (float*)h->seg_max.p, init);
        p.seg_stats = (const float*)h->seg_stats.p;
        p.clip_seg = (const int64_t*)h->clip_seg.p;
        p.chunk_rec = (const ChunkRec*)h->chunk_rec.p;
        p.prefetch_dist = 2 * h->n_cus;
        p.seg_hop = h->cfg.hop_length;
        p.seg_hop_frames = h->share_H;
        p.chunk_cap = (int)h->plan_nchunks;
        p.n_chunks = (int)h->plan_nchunks;
        p.fb1[0] = h->fb1[0]; p.fb1[1] = h->fb1[1];
        p.plane_halfs = LH_CLIP_PLANE_HALFS;
        p.fft_tab = h->fft_tab; p.fft_adv2 = h->fft_adv2;
        if (h->logmel_fft)
            hipLaunchKernelGGL(k_logmel_fft_clip, dim3((unsigned)std::min<int64_t>(h->plan_nchunks, 2 * (int64_t)h->n_cus)), dim3(FC_THREADS),
                               logmel_fft_lds_bytes(), st, p);       // persistent: two workgroups per CU walk the chunk list
        else
            hipLaunchKernelGGL(k_logmel_h_clip, dim3((unsigned)h->plan_nchunks), dim3(LH_THREADS), logmel_h_clip_lds_bytes(), st, p);
    } else if (h->logmel_f32) hipLaunchKernelGGL(k_logmel, dim3((unsigned)n_seg), dim3(LM_THREADS), logmel_lds_bytes(), st, p);
    else hipLaunchKernelGGL(k_logmel_h, dim3((unsigned)(16 * ((n_seg + 7) / 8))), dim3(LH_THREADS), logmel_h_lds_bytes(wg_frames), st, p);
    h->prof_logmel.end(st);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

static void fill_projpool(radad_embed_t h, ProjPoolParams& p) {
    p.logmel = (const float*)h->logmel.p; p.seg_max = (const float*)h->seg_max.p; p.clip_seg = (const int64_t*)h->clip_seg.p;
    p.nf = h->nf; p.T = h->T; p.F = h->cfg.feat_dim; p.wfrag_h = h->wfrag_h; p.wscale = h->wscale; p.bias = h->bias; p.n_levels = h->cfg.n_levels;
    for (int i = 0; i < RADAD_MAX_LEVELS; ++i) p.levels[i] = i < h->cfg.n_levels ? h->cfg.levels[i] : 0;
    p.pool_mode = h->cfg.pool_mode; p.out = nullptr; p.out_bf16 = nullptr; p.frames_out = nullptr; p.n_groups_dev = nullptr;
    p.debug = 0;
#ifdef RADAD_DEBUG_HOOKS
    { const char* dbg = getenv("RADAD_DEBUG_LOGMEL"); p.debug = dbg ? atoi(dbg) : 0; }
#endif
}

extern "C" {

int radad_embed_create(const radad_embed_cfg* cfg, const float* mel_filters_host, const float* proj_w_host,
                       const float* proj_b_host, int device, radad_embed_t* out) {
    return radad_embed_create_ex(cfg, 0, mel_filters_host, proj_w_host, proj_b_host, device, out);
}

int radad_embed_create_ex(const radad_embed_cfg* cfg, int flags, const float* mel_filters_host, const float* proj_w_host,
                          const float* proj_b_host, int device, radad_embed_t* out) {
    RADAD_REQUIRE((flags & ~(RADAD_EMBED_NO_SHARED_FRAMES | RADAD_EMBED_LOGMEL_F32 | RADAD_EMBED_LOGMEL_DFT_GEMM)) == 0,
                  "radad_embed_create_ex: unknown flag bits 0x%x", flags);
    RADAD_REQUIRE(cfg && mel_filters_host && proj_w_host && proj_b_host && out, "radad_embed_create: NULL argument");
    RADAD_REQUIRE(cfg->n_fft == N_FFT && cfg->fft_hop == FFT_HOP && cfg->n_mels == N_MELS,
                  "radad_embed_create: only n_fft=400 / hop=160 / 80 mels (the HF Whisper front-end) is built");
    RADAD_REQUIRE(cfg->segment_length > 0 && cfg->segment_length <= MAX_SEG_LEN && cfg->segment_length % FFT_HOP == 0 &&
                      cfg->segment_length >= N_FFT,
                  "radad_embed_create: segment_length must be a multiple of 160 in [400, %d] (got %d)", MAX_SEG_LEN,
                  cfg->segment_length);
    RADAD_REQUIRE(cfg->hop_length > 0, "radad_embed_create: hop_length must be positive");
    RADAD_REQUIRE(cfg->feat_dim > 0 && cfg->feat_dim % 32 == 0, "radad_embed_create: feat_dim must be a multiple of 32");
    RADAD_REQUIRE(cfg->n_levels >= 1 && cfg->n_levels <= RADAD_MAX_LEVELS, "radad_embed_create: 1..%d pyramid levels", RADAD_MAX_LEVELS);
    RADAD_REQUIRE(cfg->pool_mode == RADAD_POOL_MAX || cfg->pool_mode == RADAD_POOL_AVG, "Unsupported pooling type: %d", cfg->pool_mode);
    int nbins = 0;
    for (int i = 0; i < cfg->n_levels; ++i) {
        RADAD_REQUIRE(cfg->levels[i] >= 1, "radad_embed_create: pyramid level must be >= 1");
        nbins += cfg->levels[i];
    }
    RADAD_REQUIRE(nbins <= MAX_BINS_TOTAL, "radad_embed_create: sum(levels) must be <= %d", MAX_BINS_TOTAL);
    const int L = cfg->segment_length;
    const int padded = cfg->padded_samples > 0 ? cfg->padded_samples : L;
    RADAD_REQUIRE(padded >= L && padded % FFT_HOP == 0, "radad_embed_create: padded_samples must be a multiple of 160 >= segment_length");
    const int T = padded / FFT_HOP;
    const int nf_sig = (L + 200 + FFT_HOP - 1) / FFT_HOP;   // frames whose window still touches the segment
    const int nf = std::min(T, nf_sig);
    RADAD_REQUIRE(nf <= LM_WAVES * 32, "radad_embed_create: %d frames per segment exceed the %d the kernel covers", nf, LM_WAVES * 32);
    RADAD_REQUIRE(padded == L || padded >= L + 2 * N_FFT, "radad_embed_create: padded_samples must be 0 or >= segment_length + 800");
    RADAD_REQUIRE(std::max(lh_split_f0(nf), nf - lh_split_f0(nf)) <= LH_WG_FRAMES,
                  "radad_create: %d frames per segment exceed the two workgroups of k_logmel_h", nf);
    int ndev = 0;
    RADAD_HIP_CHECK(hipGetDeviceCount(&ndev));
    RADAD_REQUIRE(device >= 0 && device < ndev, "radad_embed_create: device %d not in [0,%d)", device, ndev);

    radad_embed_s* h = new (std::nothrow) radad_embed_s();
    if (!h) { radad_set_error("out of host memory"); return RADAD_ENOMEM; }
    h->cfg = *cfg; h->device = device; h->nbins = nbins; h->T = T; h->nf = nf; h->padded = padded;
    DeviceGuard g(device);
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->n_cus = cus; }

    // folded, windowed DFT basis in MFMA fragment order
    std::vector<float> basis((size_t)NBT * NKK * 2 * 64 * 4);
    const double PI = 3.14159265358979323846;
    for (int bt = 0; bt < NBT; ++bt)
        for (int kk = 0; kk < NKK; ++kk)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int n = 8 * kk + 4 * (lane >> 5) + j + 1;     // 1..200
                    const int bin = 32 * bt + (lane & 31);
                    double c = 0.0, s = 0.0;
                    if (bin < N_BINS) {
                        const double w = 0.5 - 0.5 * cos(2.0 * PI * n / N_FFT);   // periodic hann
                        const int ph = (int)(((int64_t)bin * n) % N_FFT);
                        if (n < 200) { c = w * cos(2.0 * PI * ph / N_FFT); s = w * sin(2.0 * PI * ph / N_FFT); }
                        else { c = 0.5 * ((bin & 1) ? -1.0 : 1.0); s = 0.0; }      // n = 200 pairs with itself
                    }
                    const size_t base = ((((size_t)bt * NKK + kk) * 2) * 64 + lane) * 4 + j;
                    basis[base] = (float)c;
                    basis[base + 256] = (float)s;
                }
    // unfolded, windowed DFT basis x 2^11 as f16 hi/lo planes in MFMA fragment order (k_logmel_h)
    std::vector<_Float16> basis_h((size_t)LH_NG * 4 * 64 * 8);
    for (int bt = 0; bt < NBT; ++bt)
        for (int step = 0; step < LH_STEPS; ++step)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int n = 16 * step + 8 * (lane >> 5) + j;      // 0..399
                    const int bin = 32 * bt + (lane & 31);
                    double c = 0.0, sn = 0.0;
                    if (bin < N_BINS) {
                        const double w = 0.5 - 0.5 * cos(2.0 * PI * n / N_FFT);   // periodic hann
                        const int ph = (int)(((int64_t)bin * n) % N_FFT);
                        c = w * cos(2.0 * PI * ph / N_FFT);
                        sn = w * sin(2.0 * PI * ph / N_FFT);
                    }
                    const float cf = (float)(c * LH_BASIS_SCALE), sf = (float)(sn * LH_BASIS_SCALE);
                    const _Float16 chi = (_Float16)cf, shi_ = (_Float16)sf;
                    const size_t base = (((size_t)(bt * LH_GPB + step) * 4) * 64 + lane) * 8 + j;
                    basis_h[base] = chi;
                    basis_h[base + 1 * 512] = (_Float16)(cf - (float)chi);
                    basis_h[base + 2 * 512] = shi_;
                    basis_h[base + 3 * 512] = (_Float16)(sf - (float)shi_);
                }
    // mel filter bank in MFMA A-fragment order
    std::vector<float> fbfrag((size_t)NBT * NMT * 16 * 64, 0.f);
    unsigned nz = 0;
    for (int bt = 0; bt < NBT; ++bt)
        for (int mt = 0; mt < NMT; ++mt)
            for (int r = 0; r < 16; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    const int bin = 32 * bt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int mel = 32 * mt + (lane & 31);
                    float v = 0.f;
                    if (bin < N_BINS && mel < N_MELS) v = mel_filters_host[bin * N_MELS + mel];
                    if (v != 0.f) nz |= 1u << (bt * NMT + mt);
                    fbfrag[(((size_t)bt * NMT + mt) * 16 + r) * 64 + lane] = v;
                }
    h->nzmask = nz;
    // the same filter bank for k_logmel_h: f16 hi/lo of fb x 2^16, k' = 8 lh + j  <->  bin (j&3) + 16 h + 8 (j>>2) + 4 lh.  It rides in
    // the basis stream: groups LH_STEPS .. LH_STEPS + NMT - 1 of each bin tile (the rest of the round stays zero padding)
    for (int bt = 0; bt < NBT; ++bt)
        for (int mt = 0; mt < NMT; ++mt)
            for (int hh = 0; hh < 2; ++hh)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int bin = 32 * bt + (j & 3) + 16 * hh + 8 * (j >> 2) + 4 * (lane >> 5);
                        const int mel = 32 * mt + (lane & 31);
                        float v = 0.f;
                        if (bin < N_BINS && mel < N_MELS) v = mel_filters_host[bin * N_MELS + mel];
                        const float sc = ldexpf(v, LH_FB_SHIFT);
                        const _Float16 vh = (_Float16)sc;
                        const size_t base = ((((size_t)bt * LH_GPB + LH_STEPS + mt) * 4 + hh * 2) * 64 + lane) * 8 + j;
                        basis_h[base] = vh;
                        basis_h[base + 512] = (_Float16)(sc - (float)vh);
                    }
    const int F = cfg->feat_dim;
    // W x 2^e(feature) as f16 hi/lo in B-fragment order; e maps the column maximum into [2^13, 2^14)
    std::vector<_Float16> wfrag_h((size_t)(F / 32) * 5 * 2 * 64 * 8);
    std::vector<float> wscale(F);
    for (int f = 0; f < F; ++f) {
        float mx = 0.f;
        for (int k = 0; k < N_MELS; ++k) mx = std::max(mx, fabsf(proj_w_host[(size_t)k * F + f]));
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) {
            int fe;
            (void)frexpf(mx, &fe);
            e = std::max(-100, std::min(100, 14 - fe));
        }
        wscale[f] = ldexpf(1.0f, -e - PP_ACT_SHIFT);
        const int ft = f / 32, l31 = f % 32;
        for (int st = 0; st < 5; ++st)
            for (int lhh = 0; lhh < 2; ++lhh)
                for (int j = 0; j < 8; ++j) {
                    const int kidx = 16 * st + 8 * lhh + j;
                    const float sc = ldexpf(proj_w_host[(size_t)kidx * F + f], e);
                    const _Float16 vh = (_Float16)sc;
                    const size_t base = ((((size_t)ft * 5 + st) * 2) * 64 + lhh * 32 + l31) * 8 + j;
                    wfrag_h[base] = vh;
                    wfrag_h[base + 512] = (_Float16)(sc - (float)vh);
                }
    }
    auto put = [&](float** dst, const void* src, size_t bytes) -> int {
        if (hipMalloc((void**)dst, bytes) != hipSuccess) { radad_set_error("hipMalloc failed"); return RADAD_ENOMEM; }
        if (hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) { radad_set_error("H2D copy failed"); return RADAD_EHIP; }
        return RADAD_OK;
    };
    {
        // shared clip frames: the configuration decides (the data never does: padded and ragged clips are handled inside the kernel)
        const int hopf = cfg->hop_length / FFT_HOP;
        bool ok = padded == L && cfg->hop_length % FFT_HOP == 0 && cfg->hop_length < L && nf == T && T >= 8 && hopf >= 1 &&
                  (T - 4) / hopf + 1 <= LH_MAX_OWNERS;
        for (int mel = 0; mel < N_MELS && ok; ++mel) {
            if (mel_filters_host[0 * N_MELS + mel] != 0.f) ok = false;               // bin 0 must carry no weight
            if (mel >= 2 && mel_filters_host[1 * N_MELS + mel] != 0.f) ok = false;    // bin 1 only in bands 0 and 1
        }
        // (flags choose; the environment names are announced overrides for experiments only)
        const char* e = radad_env_override("RADAD_LOGMEL_SHARED", "0 keeps one log-mel transform per (segment, frame) even where overlapping segments share frames (k_logmel_h instead of the clip kernels), for extractors created from now on");
        if ((e && atoi(e) == 0) || (flags & RADAD_EMBED_NO_SHARED_FRAMES)) ok = false;
        h->share_frames = ok ? 1 : 0;
        h->share_H = ok ? hopf : 0;
        h->fb1[0] = mel_filters_host[1 * N_MELS + 0];
        h->fb1[1] = mel_filters_host[1 * N_MELS + 1];
    }
    { const char* e = radad_env_override("RADAD_LOGMEL_F32", "non-zero selects the fp32-MFMA log-mel kernel (k_logmel, ~2x slower) for extractors created from now on"); h->logmel_f32 = ((e && atoi(e) != 0) || (flags & RADAD_EMBED_LOGMEL_F32)) ? 1 : 0; }
    std::vector<float> fft_tab(FC_TAB_FLOATS);
    {
        // the shared-frame work list as a radix FFT on the vector ALU whenever the configuration allows sharing and the filter bank is triangular
        bool ok = h->share_frames && !h->logmel_f32 && fc_build_tables(mel_filters_host, fft_tab.data(), &h->fft_adv2);
        const char* e = radad_env_override("RADAD_LOGMEL_FFT", "0 keeps the DFT-as-GEMM log-mel kernel (k_logmel_h_clip) instead of the radix FFT (k_logmel_fft_clip) for extractors created from now on");
        if ((e && atoi(e) == 0) || (flags & RADAD_EMBED_LOGMEL_DFT_GEMM)) ok = false;
        h->logmel_fft = ok ? 1 : 0;
    }
    int rc = put(&h->basis, basis.data(), basis.size() * sizeof(float));
    if (!rc && h->logmel_fft) rc = put(&h->fft_tab, fft_tab.data(), fft_tab.size() * sizeof(float));
    if (!rc) rc = put((float**)&h->basis_h, basis_h.data(), basis_h.size() * sizeof(_Float16));
    if (!rc) rc = put(&h->fbfrag, fbfrag.data(), fbfrag.size() * sizeof(float));
    if (!rc) rc = put((float**)&h->wfrag_h, wfrag_h.data(), wfrag_h.size() * sizeof(_Float16));
    if (!rc) rc = put(&h->wscale, wscale.data(), wscale.size() * sizeof(float));
    if (!rc) rc = put(&h->bias, proj_b_host, (size_t)F * sizeof(float));
    if (!rc) rc = put((float**)&h->levels_dev, cfg->levels, sizeof(int32_t) * RADAD_MAX_LEVELS);
    if (!rc) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_logmel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)logmel_lds_bytes()) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_logmel_h), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)logmel_h_lds_bytes(LH_WG_FRAMES)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_logmel_h_clip), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)logmel_h_clip_lds_bytes()) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_logmel_fft_clip), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)logmel_fft_lds_bytes()) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_pool2<1, RADAD_POOL_MAX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)projpool2_lds_bytes(1)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_pool2<1, RADAD_POOL_AVG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)projpool2_lds_bytes(1)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_pool2<2, RADAD_POOL_MAX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)projpool2_lds_bytes(2)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_pool2<2, RADAD_POOL_AVG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)projpool2_lds_bytes(2)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_pool<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)projpool_lds_bytes(8)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_pool<false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)projpool_lds_bytes(16)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_pool<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)projpool_lds_bytes(8)) != hipSuccess) {
            radad_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
            rc = RADAD_EHIP;
        }
    }
    if (rc) { radad_embed_destroy(h); return rc; }
    *out = h;
    return RADAD_OK;
}

int radad_embed_destroy(radad_embed_t h) {
    if (!h) return RADAD_OK;
    {
        DeviceGuard g(h->device);
        if (h->basis) (void)hipFree(h->basis);
        if (h->basis_h) (void)hipFree(h->basis_h);
        if (h->fft_tab) (void)hipFree(h->fft_tab);
        if (h->fbfrag) (void)hipFree(h->fbfrag);
        if (h->wfrag_h) (void)hipFree(h->wfrag_h);
        if (h->wscale) (void)hipFree(h->wscale);
        if (h->bias) (void)hipFree(h->bias);
        if (h->levels_dev) (void)hipFree(h->levels_dev);
        h->seg_start.release(); h->seg_valid.release(); h->clip_seg.release(); h->logmel.release(); h->seg_max.release();
        h->seg_pool.release(); h->clip_off.release(); h->n_seg_dev.release();
        h->seg_stats.release(); h->chunk_rec.release();
        for (int i = 0; i < 2; ++i) {
            if (h->pin[i]) (void)hipHostFree(h->pin[i]);
            if (h->pin_ev[i]) (void)hipEventDestroy(h->pin_ev[i]);
        }
        if (h->plan_flags_host) (void)hipHostFree(h->plan_flags_host);
        for (int i = 0; i < radad_embed_s::PLAN_RING; ++i)
            if (h->plan_flags_ev[i]) (void)hipEventDestroy(h->plan_flags_ev[i]);
        h->misc.release();
        h->prof_logmel.destroy();
        h->prof_pool.destroy();
    }
    delete h;
    return RADAD_OK;
}

int radad_embed_output_dim(radad_embed_t h, int* dim) { RADAD_REQUIRE(h && dim, "NULL argument"); *dim = h->nbins * h->cfg.feat_dim; return RADAD_OK; }
int radad_embed_num_frames(radad_embed_t h, int* frames) { RADAD_REQUIRE(h && frames, "NULL argument"); *frames = h->T; return RADAD_OK; }

// log-mel + projection/pooling/segment mean over the plan held by the handle.  n_groups_x: clips (uniform batches: one
// workgroup pools all segments of a clip) or an upper bound on the segments (ragged: pooled per segment, then averaged).
static int embed_run(radad_embed_t h, const float* wave_dev, int64_t n_clips, int64_t n_seg_grid, bool per_clip, void* out_dev,
                     int out_dtype, const int* n_seg_dev, hipStream_t st) {
    int rc;
    if ((rc = launch_logmel(h, wave_dev, n_seg_grid, nullptr, st, n_seg_dev))) return rc;
    ProjPoolParams p;
    fill_projpool(h, p);
    const int out_dim = h->nbins * h->cfg.feat_dim;
    const bool bf16 = out_dtype == RADAD_OUT_BF16;
    int64_t n_groups = n_clips;           // workgroups along x: clips, or segments for a ragged batch
    if (per_clip) {
        if (bf16) p.out_bf16 = (unsigned short*)out_dev; else p.out = (float*)out_dev;
    } else {
        if ((rc = h->seg_pool.ensure((size_t)n_seg_grid * out_dim * sizeof(float)))) return rc;
        p.out = (float*)h->seg_pool.p;
        p.clip_seg = nullptr;             // one group per segment
        p.n_groups_dev = n_seg_dev;
        n_groups = n_seg_grid;
    }
    h->prof_pool.begin(st);
    // persistent workgroups (one per CU) walk the groups; the next pass's rows are in flight across the current pass's MFMAs
    const unsigned gx = (unsigned)std::min<int64_t>(n_groups, h->n_cus);
    const bool pmax = h->cfg.pool_mode == RADAD_POOL_MAX;
    if (h->cfg.feat_dim > 256) {
        const dim3 grid(gx, (unsigned)((h->cfg.feat_dim + 511) / 512));
        if (pmax) hipLaunchKernelGGL((k_proj_pool2<2, RADAD_POOL_MAX>), grid, dim3(PP2_THREADS), projpool2_lds_bytes(2), st, p, (int)n_groups);
        else hipLaunchKernelGGL((k_proj_pool2<2, RADAD_POOL_AVG>), grid, dim3(PP2_THREADS), projpool2_lds_bytes(2), st, p, (int)n_groups);
    } else {
        if (pmax) hipLaunchKernelGGL((k_proj_pool2<1, RADAD_POOL_MAX>), dim3(gx, 1), dim3(PP2_THREADS), projpool2_lds_bytes(1), st, p, (int)n_groups);
        else hipLaunchKernelGGL((k_proj_pool2<1, RADAD_POOL_AVG>), dim3(gx, 1), dim3(PP2_THREADS), projpool2_lds_bytes(1), st, p, (int)n_groups);
    }
    if (!per_clip)                        // pipeline.py:411: mean over each clip's segment vectors, in segment order
        hipLaunchKernelGGL(k_group_mean, dim3((unsigned)n_clips, (unsigned)((out_dim + 255) / 256)), dim3(256), 0, st,
                           (const float*)h->seg_pool.p, (const int64_t*)h->clip_seg.p, out_dim, bf16 ? nullptr : (float*)out_dev,
                           bf16 ? (unsigned short*)out_dev : nullptr);
    h->prof_pool.end(st);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

int radad_embed_forward_ex(radad_embed_t h, const float* wave_dev, const int64_t* clip_offsets_host, int64_t n_clips,
                           void* out_dev, int out_dtype, void* stream) {
    RADAD_REQUIRE(h, "NULL handle");
    RADAD_REQUIRE(n_clips >= 0, "radad_embed_forward: n_clips < 0");
    RADAD_REQUIRE(out_dtype == RADAD_OUT_F32 || out_dtype == RADAD_OUT_BF16, "radad_embed_forward: unsupported output dtype %d", out_dtype);
    if (n_clips == 0) return RADAD_OK;
    RADAD_REQUIRE(wave_dev && clip_offsets_host && out_dev, "radad_embed_forward: NULL buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = build_plan(h, clip_offsets_host, n_clips, st);
    if (rc) return rc;
    return embed_run(h, wave_dev, n_clips, h->plan_nseg, h->plan_uniform, out_dev, out_dtype, nullptr, st);
}

int radad_embed_forward(radad_embed_t h, const float* wave_dev, const int64_t* clip_offsets_host, int64_t n_clips,
                        float* out_dev, void* stream) {
    return radad_embed_forward_ex(h, wave_dev, clip_offsets_host, n_clips, out_dev, RADAD_OUT_F32, stream);
}

// 16-bit PCM -> float32 as the reference's loader hands it over (librosa / soundfile: sample / 32768, exact in fp32): audio files
// are int16, so a deployment that uploads the PCM and converts here moves half the bytes over PCIe (bench.py's pcie_inclusive leg)
__global__ __launch_bounds__(256) void k_pcm16_to_f32(const short* __restrict__ in, float* __restrict__ out, int64_t n) {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const s16x8 v = *reinterpret_cast<const s16x8*>(in + i);
        f32x4 a = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, b = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
        a *= 0x1p-15f; b *= 0x1p-15f;
        *reinterpret_cast<f32x4*>(out + i) = a;
        *reinterpret_cast<f32x4*>(out + i + 4) = b;
    } else {
        for (int64_t j = i; j < n && j < i + 8; ++j) out[j] = (float)in[j] * 0x1p-15f;
    }
}

int radad_pcm16_to_f32(const int16_t* pcm_dev, float* out_dev, int64_t n, int device, void* stream) {
    RADAD_REQUIRE(n >= 0, "radad_pcm16_to_f32: negative length");
    if (n == 0) return RADAD_OK;
    RADAD_REQUIRE(pcm_dev && out_dev, "radad_pcm16_to_f32: NULL buffer");
    DeviceGuard g(device);
    hipLaunchKernelGGL(k_pcm16_to_f32, dim3((unsigned)ceil_div64(n, 2048)), dim3(256), 0, (hipStream_t)stream, (const short*)pcm_dev, out_dev, n);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

int radad_embed_forward_dev(radad_embed_t h, const float* wave_dev, const int64_t* clip_offsets_dev, int64_t n_clips,
                            int64_t n_samples_total, void* out_dev, int out_dtype, void* stream) {
    RADAD_REQUIRE(h, "NULL handle");
    RADAD_REQUIRE(n_clips >= 0 && n_samples_total >= 0, "radad_embed_forward_dev: negative size");
    RADAD_REQUIRE(out_dtype == RADAD_OUT_F32 || out_dtype == RADAD_OUT_BF16, "radad_embed_forward_dev: unsupported output dtype %d", out_dtype);
    if (n_clips == 0) return RADAD_OK;
    RADAD_REQUIRE(wave_dev && clip_offsets_dev && out_dev, "radad_embed_forward_dev: NULL buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    // a clip of N samples has max(1, (N - L)//hop + 1) <= N/hop + 1 segments: the grids and buffers are sized for that
    // bound, the kernels read the true count from the device
    const int64_t seg_cap = n_samples_total / h->cfg.hop_length + n_clips;
    RADAD_REQUIRE(seg_cap < (1ll << 31), "radad_embed_forward_dev: too many segments in one batch");
    // chunks: a clip of S segments has <= ((S - 1) H + T) / 96 + 1 + 3 S / 32 + 1 of them
    const int64_t chunk_cap = !h->share_frames ? 0
                              : h->logmel_fft ? (seg_cap * h->share_H + n_clips * h->nf) / FC_SLOTS + (3 * seg_cap) / FC_EDGE_CHUNK + 2 * n_clips + 1
                                              : (seg_cap * h->share_H + n_clips * h->nf) / LH_CLIP_FRAMES + (3 * seg_cap) / LH_EDGE_CHUNK + 2 * n_clips + 1;
    RADAD_REQUIRE(chunk_cap < (1ll << 30), "radad_embed_forward_dev: too many frames in one batch");
    int rc = plan_on_device(h, clip_offsets_dev, n_clips, seg_cap, n_samples_total, st, chunk_cap);
    if (rc) return rc;
    h->plan_key.clear();
    h->plan_on_device = true;
    h->plan_nseg = seg_cap;
    if ((rc = embed_run(h, wave_dev, n_clips, seg_cap, false, out_dev, out_dtype, (const int*)h->n_seg_dev.p, st))) return rc;
    // what k_build_plan had to repair travels to pinned host memory behind the batch (radad_embed_plan_flags[_poll] read it)
    if (!h->plan_flags_host) {
        RADAD_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->plan_flags_host), radad_embed_s::PLAN_RING * sizeof(int), hipHostMallocDefault));
        for (int i = 0; i < radad_embed_s::PLAN_RING; ++i) RADAD_HIP_CHECK(hipEventCreateWithFlags(&h->plan_flags_ev[i], hipEventDisableTiming));
    }
    const int slot = h->plan_flags_next;
    if (h->plan_flags_pending[slot]) {           // four unread batches in flight: the oldest is folded into the running report first
        RADAD_HIP_CHECK(hipEventSynchronize(h->plan_flags_ev[slot]));
        h->plan_flags_accum |= h->plan_flags_host[slot];
        h->plan_flags_pending[slot] = false;
    }
    RADAD_HIP_CHECK(hipMemcpyAsync(h->plan_flags_host + slot, (const int*)h->n_seg_dev.p + 1, sizeof(int), hipMemcpyDeviceToHost, st));
    RADAD_HIP_CHECK(hipEventRecord(h->plan_flags_ev[slot], st));
    h->plan_flags_pending[slot] = true;
    h->plan_flags_next = (slot + 1) % radad_embed_s::PLAN_RING;
    return RADAD_OK;
}

static int plan_flags_collect(radad_embed_t h, bool wait, int* flags_out, int* pending_out) {
    int pending = 0;
    for (int i = 0; i < radad_embed_s::PLAN_RING; ++i) {
        if (!h->plan_flags_pending[i]) continue;
        if (wait) RADAD_HIP_CHECK(hipEventSynchronize(h->plan_flags_ev[i]));
        else if (hipEventQuery(h->plan_flags_ev[i]) != hipSuccess) { (void)hipGetLastError(); ++pending; continue; }
        h->plan_flags_accum |= h->plan_flags_host[i];
        h->plan_flags_pending[i] = false;
    }
    *flags_out = h->plan_flags_accum;
    h->plan_flags_accum = 0;                   // reported once
    if (pending_out) *pending_out = pending;
    return RADAD_OK;
}

int radad_embed_plan_flags(radad_embed_t h, int* flags_out) {
    RADAD_REQUIRE(h && flags_out, "radad_embed_plan_flags: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    return plan_flags_collect(h, true, flags_out, nullptr);
}

int radad_embed_plan_flags_poll(radad_embed_t h, int* flags_out, int* batches_pending_out) {
    RADAD_REQUIRE(h && flags_out, "radad_embed_plan_flags_poll: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    return plan_flags_collect(h, false, flags_out, batches_pending_out);
}

int radad_embed_clip_chunks(int n_segments, int frames_per_segment, int hop_frames, int32_t* out5) {
    RADAD_REQUIRE(out5 && n_segments >= 0 && frames_per_segment >= 8 && hop_frames >= 1, "radad_embed_clip_chunks: bad argument");
    const ClipChunks c = clip_chunking(n_segments, frames_per_segment, hop_frames);
    out5[0] = c.n_full; out5[1] = c.r; out5[2] = c.e_tail; out5[3] = c.n_edge_chunks;
    out5[4] = clip_chunk_count(n_segments, frames_per_segment, hop_frames);
    return RADAD_OK;
}

int radad_embed_fft_clip_chunks(int n_segments, int frames_per_segment, int hop_frames, int32_t* out5) {
    RADAD_REQUIRE(out5 && n_segments >= 0 && frames_per_segment >= 8 && hop_frames >= 1, "radad_embed_fft_clip_chunks: bad argument");
    const ClipChunks c = fc_chunking(n_segments, frames_per_segment, hop_frames);
    out5[0] = c.n_full; out5[1] = c.r; out5[2] = c.e_tail; out5[3] = c.n_edge_chunks;
    out5[4] = fc_chunk_count(n_segments, frames_per_segment, hop_frames);
    return RADAD_OK;
}

int radad_embed_fft_tables(const float* mel_filters_host, float* tab_out, int cap, int32_t* info4) {
    RADAD_REQUIRE(mel_filters_host && tab_out && info4 && cap >= FC_TAB_FLOATS, "radad_embed_fft_tables: bad argument (the table holds %d floats)", FC_TAB_FLOATS);
    int adv2 = 0;
    const bool ok = fc_build_tables(mel_filters_host, tab_out, &adv2);
    info4[0] = ok ? 1 : 0; info4[1] = FC_TAB_FLOATS; info4[2] = adv2; info4[3] = FC_COMB;
    return RADAD_OK;
}

int radad_embed_last_logmel_kind(radad_embed_t h, int* kind_out) {
    RADAD_REQUIRE(h && kind_out, "radad_embed_last_logmel_kind: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    *kind_out = h->last_logmel_kind;
    return RADAD_OK;
}

int radad_embed_profile(radad_embed_t h, int enable) {
    RADAD_REQUIRE(h, "NULL handle");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    int rc = h->prof_logmel.enable(enable != 0);
    if (!rc) rc = h->prof_pool.enable(enable != 0);
    if (rc) radad_set_error("hipEventCreate failed");
    return rc;
}

int radad_embed_profile_read(radad_embed_t h, float* logmel_ms_out, float* projpool_ms_out, int cap, int* n_out) {
    RADAD_REQUIRE(h && logmel_ms_out && projpool_ms_out && cap >= 0, "radad_embed_profile_read: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    int n1 = 0, n2 = 0;
    int rc = h->prof_logmel.read(logmel_ms_out, cap, &n1);
    if (!rc) rc = h->prof_pool.read(projpool_ms_out, cap, &n2);
    if (rc) { radad_set_error("reading profile events failed"); return rc; }
    if (n_out) *n_out = n1 < n2 ? n1 : n2;
    return RADAD_OK;
}

int radad_embed_normalize(radad_embed_t h, const float* wave_dev, const int64_t* seg_start_host, const int32_t* seg_valid_host,
                          int64_t n_seg, float* out_dev, void* stream) {
    RADAD_REQUIRE(h, "NULL handle");
    if (n_seg == 0) return RADAD_OK;
    RADAD_REQUIRE(wave_dev && seg_start_host && seg_valid_host && out_dev && n_seg > 0, "radad_embed_normalize: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = upload_plan_explicit(h, seg_start_host, seg_valid_host, n_seg, st);
    if (rc) return rc;
    return launch_logmel(h, wave_dev, n_seg, out_dev, st);
}

int radad_embed_logmel(radad_embed_t h, const float* wave_dev, const int64_t* seg_start_host, const int32_t* seg_valid_host,
                       int64_t n_seg, float* out_dev, void* stream) {
    RADAD_REQUIRE(h, "NULL handle");
    if (n_seg == 0) return RADAD_OK;
    RADAD_REQUIRE(wave_dev && seg_start_host && seg_valid_host && out_dev && n_seg > 0, "radad_embed_logmel: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = upload_plan_explicit(h, seg_start_host, seg_valid_host, n_seg, st);
    if (rc) return rc;
    if ((rc = launch_logmel(h, wave_dev, n_seg, nullptr, st))) return rc;
    hipLaunchKernelGGL(k_logmel_finalize, dim3(2048), dim3(256), 0, st, (const float*)h->logmel.p, (const float*)h->seg_max.p, h->nf,
                       h->T, n_seg, out_dev);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

int radad_embed_frame_features(radad_embed_t h, const float* wave_dev, const int64_t* seg_start_host,
                               const int32_t* seg_valid_host, int64_t n_seg, float* out_dev, void* stream) {
    RADAD_REQUIRE(h, "NULL handle");
    if (n_seg == 0) return RADAD_OK;
    RADAD_REQUIRE(wave_dev && seg_start_host && seg_valid_host && out_dev && n_seg > 0, "radad_embed_frame_features: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = upload_plan_explicit(h, seg_start_host, seg_valid_host, n_seg, st);
    if (rc) return rc;
    if ((rc = launch_logmel(h, wave_dev, n_seg, nullptr, st))) return rc;
    ProjPoolParams p;
    fill_projpool(h, p);
    p.frames_out = out_dev;
    hipLaunchKernelGGL((k_proj_pool<true, 8>), dim3((unsigned)n_seg, (unsigned)((h->cfg.feat_dim + 255) / 256)), dim3(512),
                       projpool_lds_bytes(8), st, p);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

int radad_tpp_forward(const float* feats_dev, const int64_t* row_offsets_host, int64_t n_items, int feat_dim,
                      const int32_t* levels, int n_levels, int pool_mode, float* out_dev, int device, void* stream) {
    RADAD_REQUIRE(n_items >= 0 && feat_dim > 0, "radad_tpp_forward: bad shape");
    RADAD_REQUIRE(levels && n_levels >= 1 && n_levels <= RADAD_MAX_LEVELS, "radad_tpp_forward: 1..%d levels", RADAD_MAX_LEVELS);
    RADAD_REQUIRE(pool_mode == RADAD_POOL_MAX || pool_mode == RADAD_POOL_AVG, "Unsupported pooling type: %d", pool_mode);
    if (n_items == 0) return RADAD_OK;
    RADAD_REQUIRE(feats_dev && row_offsets_host && out_dev, "radad_tpp_forward: NULL buffer");
    int nbins = 0;
    for (int i = 0; i < n_levels; ++i) { RADAD_REQUIRE(levels[i] >= 1, "radad_tpp_forward: level must be >= 1"); nbins += levels[i]; }
    for (int64_t i = 0; i < n_items; ++i)
        RADAD_REQUIRE(row_offsets_host[i + 1] > row_offsets_host[i], "radad_tpp_forward: item %lld has no rows", (long long)i);
    DeviceGuard g(device);
    hipStream_t st = (hipStream_t)stream;
    // small per-call upload (offsets + levels); freed after the launch is enqueued and the stream drained
    const size_t off_bytes = (size_t)(n_items + 1) * sizeof(int64_t);
    char* tmp = nullptr;
    if (hipMalloc((void**)&tmp, off_bytes + sizeof(int32_t) * RADAD_MAX_LEVELS) != hipSuccess) { radad_set_error("hipMalloc failed"); return RADAD_ENOMEM; }
    int rc = RADAD_OK;
    if (hipMemcpyAsync(tmp, row_offsets_host, off_bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(tmp + off_bytes, levels, sizeof(int32_t) * n_levels, hipMemcpyHostToDevice, st) != hipSuccess) {
        radad_set_error("H2D copy failed");
        rc = RADAD_EHIP;
    }
    if (!rc) {
        hipLaunchKernelGGL(k_tpp, dim3((unsigned)n_items, (unsigned)((feat_dim + 255) / 256)), dim3(256), 0, st, feats_dev,
                           (const int64_t*)tmp, feat_dim, n_levels, (const int*)(tmp + off_bytes), pool_mode, nbins, out_dev);
        if (hipGetLastError() != hipSuccess) { radad_set_error("k_tpp launch failed"); rc = RADAD_EHIP; }
    }
    if (hipStreamSynchronize(st) != hipSuccess && !rc) { radad_set_error("k_tpp failed"); rc = RADAD_EHIP; }
    (void)hipFree(tmp);
    return rc;
}

int radad_group_mean(const float* in_dev, const int64_t* group_offsets_host, int64_t n_groups, int dim, float* out_dev,
                     int device, void* stream) {
    RADAD_REQUIRE(n_groups >= 0 && dim > 0, "radad_group_mean: bad shape");
    if (n_groups == 0) return RADAD_OK;
    RADAD_REQUIRE(in_dev && group_offsets_host && out_dev, "radad_group_mean: NULL buffer");
    for (int64_t i = 0; i < n_groups; ++i)
        RADAD_REQUIRE(group_offsets_host[i + 1] > group_offsets_host[i], "radad_group_mean: group %lld is empty", (long long)i);
    DeviceGuard g(device);
    hipStream_t st = (hipStream_t)stream;
    const size_t off_bytes = (size_t)(n_groups + 1) * sizeof(int64_t);
    int64_t* tmp = nullptr;
    if (hipMalloc((void**)&tmp, off_bytes) != hipSuccess) { radad_set_error("hipMalloc failed"); return RADAD_ENOMEM; }
    int rc = RADAD_OK;
    if (hipMemcpyAsync(tmp, group_offsets_host, off_bytes, hipMemcpyHostToDevice, st) != hipSuccess) { radad_set_error("H2D copy failed"); rc = RADAD_EHIP; }
    if (!rc) {
        hipLaunchKernelGGL(k_group_mean, dim3((unsigned)n_groups, (unsigned)((dim + 255) / 256)), dim3(256), 0, st, in_dev, tmp, dim, out_dev, (unsigned short*)nullptr);
        if (hipGetLastError() != hipSuccess) { radad_set_error("k_group_mean launch failed"); rc = RADAD_EHIP; }
    }
    if (hipStreamSynchronize(st) != hipSuccess && !rc) { radad_set_error("k_group_mean failed"); rc = RADAD_EHIP; }
    (void)hipFree(tmp);
    return rc;
}

}  // extern "C"
