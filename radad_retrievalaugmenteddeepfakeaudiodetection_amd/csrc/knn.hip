// knn.hip -- HBM-resident vector store + brute-force top-k for gfx950 (MI355X).
//
// Replaces the faiss IndexFlatL2 / IndexFlatIP object RADAD keeps in VectorDatabase.index
// (vector_database.py:56-97 create, :138 add, :181 search; pipeline.py:503 reconstruct).
//
// Kernels
//   k_rows_prepare   : append path -- copy (IP), copy + |y|^2 (L2) or normalise (cosine) incoming rows.
//   k_knn_f32        : the scan.  S = Y . Q^T on v_mfma_f32_32x32x2_f32 (exact f32) with a fused
//                      per-workgroup top-k: 128 store rows x 128 queries per tile, K stepped 32 at a
//                      time through double-buffered LDS; the store rows are the MFMA "A" side so that a
//                      lane owns ONE query column and its running k-th-best threshold lives in a register.
//   k_merge_refine   : picks k+6 fp32 candidates per query from the partial lists, re-scores them in float64 and
//                      ranks by (distance, id); k_merge_lists: P-way merge of final per-shard lists.
//   k_gather_rows    : batched reconstruct.
//
// Layout in HBM: rows [capacity, dim] fp32 row-major (insertion order), ynorm [capacity] fp32 (L2 only).
// Partial results: score [nq, n_splits, k] fp32 + local row index int32, same shape.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <type_traits>
#include <mutex>
#include <new>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {

constexpr int KT_M = 128;       // store rows per tile (MFMA A rows)
constexpr int KT_N = 128;       // queries per tile (MFMA B columns)
constexpr int KT_K = 32;        // K step
constexpr int KT_LD = KT_K + 4; // padded LDS row (floats): 36-dword stride is conflict-free for ds_read_b128
constexpr int KNN_THREADS = 256;
constexpr int KS_LD = KT_M + 4;  // score-tile row (floats): 132-dword stride keeps b128 writes and reads conflict-free
constexpr int IDX_SENTINEL = 0x7fffffff;

struct KnnParams {
    const void* db;       // [n, dim] fp32, or fp16 when db_f16
    const float* ynorm;   // [n] (L2) or nullptr
    const void* q;        // [nq, dim] (already normalised for cosine); fp16 for the fp16 tile kernel, fp32 otherwise
    int db_f16;
    int64_t n;
    int nq;
    int dim;
    int k;
    int l2;               // 1: score = 2*dot - |y|^2 ; 0: score = dot
    int n_qtiles;
    int n_splits;         // multiple of 8
    int64_t chunk_rows;   // rows per split, multiple of KT_M
    float* part_score;    // [nq, n_splits, k]
    int* part_idx;        // [nq, n_splits, k]
    int debug;            // timing experiments only (RADAD_DEBUG_KNN): 1 = skip the top-k epilogue
};

__device__ __forceinline__ bool better(float s, int i, float ws, int wi) {
    return s > ws || (s == ws && i < wi);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One 128 x 128 tile of S = Y . Q^T over the full depth: acc[mt][nt] (wave sub-tile 64 x 64 as 2 x 2 MFMA 32x32 blocks).
// Global -> LDS staging uses buffer loads: the descriptors are wave-uniform (SGPRs), rows past the end of the
// chunk / of the query block fall outside num_records and read as zero, and a thread needs only four 32-bit
// offsets for all of its loads (the K step goes into the scalar offset).
struct TileCtx {
    int dim, nk, debug;
    unsigned voff[4];     // byte offset of (row ld_r + 32 i, col ld_c) inside a tile
    int ld_r, ld_c;
    int wm, wn, l31, lh;
};

template <bool A_F16>
__device__ __forceinline__ void knn_tile_gemm(const TileCtx& c, __amdgpu_buffer_rsrc_t ra_desc, __amdgpu_buffer_rsrc_t rq_desc,
                                              float* sA, float* sB, f32x16 (&acc)[2][2]) {
    f32x4 ra[4], rb[4];
    auto gload = [&](int kc) {
        // columns past `dim` (last K step when dim % 32 != 0) must read as zero: push the OFFSET out of range rather
        // than selecting on the loaded data -- a select would make the compiler wait for the loads right here
        const bool kin = kc * KT_K + c.ld_c < c.dim;
        const int soff = kc * KT_K * (int)sizeof(float);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned vo = kin ? c.voff[i] : 0x7FFF0000u;
            if (A_F16) {      // fp16 store: 4 halfs per lane, decoded to fp32 on the way to LDS (exact)
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 hv = __builtin_amdgcn_raw_buffer_load_b64(ra_desc, kin ? (c.voff[i] >> 1) : 0x7FFF0000u, soff >> 1, 0);
                const f16x4 h4 = __builtin_bit_cast(f16x4, hv);
                ra[i] = f32x4{(float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]};
            } else {
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_desc, vo, soff, 0));
            }
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rq_desc, vo, soff, 0));
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = c.ld_r + 32 * i;
            *reinterpret_cast<f32x4*>(sA + (buf * KT_M + r) * KT_LD + c.ld_c) = ra[i];
            *reinterpret_cast<f32x4*>(sB + (buf * KT_N + r) * KT_LD + c.ld_c) = rb[i];
        }
    };
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    gload(0);
    swrite(0);
    __syncthreads();
    for (int kc = 0; kc < c.nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < c.nk) gload(kc + 1);
        const float* a_base = sA + (buf * KT_M + c.wm * 64 + c.l31) * KT_LD + 4 * c.lh;
        const float* b_base = sB + (buf * KT_N + c.wn * 64 + c.l31) * KT_LD + 4 * c.lh;
#pragma unroll
        for (int kk = 0; kk < KT_K / 8; ++kk) {
            // lanes 0-31 hold k = 8kk+j, lanes 32-63 k = 8kk+4+j of their row: one ds_read_b128 feeds 4 MFMAs
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(a_base + kk * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(a_base + 32 * KT_LD + kk * 8);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b_base + kk * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(b_base + 32 * KT_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
        }
        if (kc + 1 < c.nk) swrite(buf ^ 1);
        __syncthreads();
    }
}

// Same tile product, software-pipelined ACROSS tiles: on entry chunk 0 of this tile already sits in LDS buffer
// (gbuf & 1); during the last K step the first chunk of the NEXT tile is fetched, so no tile starts by waiting
// for HBM.  On exit buffer (gbuf & 1) holds the next tile's chunk 0 (if has_next) and the other one is free.
__device__ __forceinline__ void knn_tile_gemm_pipelined(const TileCtx& c, __amdgpu_buffer_rsrc_t cur_desc,
                                                        __amdgpu_buffer_rsrc_t next_desc, bool has_next,
                                                        __amdgpu_buffer_rsrc_t rq_desc, float* sA, float* sB, int& gbuf,
                                                        f32x16 (&acc)[2][2]) {
    f32x4 ra[4], rb[4];
    auto gload = [&](__amdgpu_buffer_rsrc_t a_desc, int kc) {
        const bool kin = kc * KT_K + c.ld_c < c.dim;
        const int soff = kc * KT_K * (int)sizeof(float);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned vo = kin ? c.voff[i] : 0x7FFF0000u;
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_desc, vo, soff, 0));
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rq_desc, vo, soff, 0));
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = c.ld_r + 32 * i;
            *reinterpret_cast<f32x4*>(sA + (buf * KT_M + r) * KT_LD + c.ld_c) = ra[i];
            *reinterpret_cast<f32x4*>(sB + (buf * KT_N + r) * KT_LD + c.ld_c) = rb[i];
        }
    };
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int kc = 0; kc < c.nk; ++kc) {
        const int buf = gbuf & 1;
        const bool more = kc + 1 < c.nk;
        if (!RADAD_DBG(c.debug, 2)) {
            if (more) gload(cur_desc, kc + 1);
            else if (has_next) gload(next_desc, 0);
        }
        const float* a_base = sA + (buf * KT_M + c.wm * 64 + c.l31) * KT_LD + 4 * c.lh;
        const float* b_base = sB + (buf * KT_N + c.wn * 64 + c.l31) * KT_LD + 4 * c.lh;
#pragma unroll
        for (int kk = 0; kk < KT_K / 8; ++kk) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(a_base + kk * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(a_base + 32 * KT_LD + kk * 8);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b_base + kk * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(b_base + 32 * KT_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
        }
        if ((more || has_next) && !RADAD_DBG(c.debug, 4)) swrite(buf ^ 1);
        if (!RADAD_DBG(c.debug, 8)) __syncthreads();
        ++gbuf;
    }
}

// (re)load chunk 0 of a tile into buffer `buf`: block prologue, and after the rare overflow path clobbered the tiles
__device__ __forceinline__ void knn_tile_prime(const TileCtx& c, __amdgpu_buffer_rsrc_t a_desc, __amdgpu_buffer_rsrc_t rq_desc,
                                               float* sA, float* sB, int buf) {
    const bool kin = c.ld_c < c.dim;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned vo = kin ? c.voff[i] : 0x7FFF0000u;
        const int r = c.ld_r + 32 * i;
        *reinterpret_cast<f32x4*>(sA + (buf * KT_M + r) * KT_LD + c.ld_c) =
            __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_desc, vo, 0, 0));
        *reinterpret_cast<f32x4*>(sB + (buf * KT_N + r) * KT_LD + c.ld_c) =
            __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rq_desc, vo, 0, 0));
    }
    __syncthreads();
}

// ---- LDS-DMA variant of the tile product (register-list kernel) -------------------------------------------------
// The ablation of the register-staged loop showed the ds_write_b128 pass costing 8 % of the scan (7.84 -> 7.20 ms
// without it): staging now goes HBM/L2 -> LDS directly with `buffer_load_dwordx4 ... lds`, no VGPR round trip, no
// ds_write, 32 VGPRs freed.  LDS-DMA writes each wave-instruction's 64 x 16 B linearly, so the tiles are UNPADDED
// (128-byte rows) and bank conflicts are avoided by an XOR swizzle applied on the SOURCE side and on the reads:
// 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7).  A 16-lane ds_read_b128 group then covers all 64 banks.
constexpr int KD_ROW_BYTES = KT_K * 4;            // 128
constexpr int KD_TILE_BYTES = KT_M * KD_ROW_BYTES;   // 16 KB per operand per buffer
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct DmaCtx {
    int dim, nk, debug;
    unsigned voff[4];        // byte offset in the tile's global rows of this lane's 16 B, per DMA instruction
    unsigned dst[4];         // wave-uniform LDS byte offset (within one operand tile) of each DMA instruction
    unsigned rd_a, rd_b;     // LDS byte offset of this lane's A / B row (mt = 0 / nt = 0)
    unsigned rd_x;           // (lh ^ swizzle) << 4: XOR-ed with 32*kk gives the chunk offset inside the row
};

__device__ __forceinline__ DmaCtx make_dma_ctx(int dim, int tid, int debug, int esize) {
    DmaCtx c;
    c.dim = dim; c.nk = (dim * esize + KD_ROW_BYTES - 1) / KD_ROW_BYTES; c.debug = debug;   // 128 bytes of every row per K step
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int grp = wave * 4 + i;                 // 8-row group written by this instruction
        const int r = grp * 8 + (lane >> 3);
        const int pc = lane & 7;                      // physical chunk this lane fills
        const int lc = pc ^ ((r >> 1) & 7);           // logical chunk it must fetch
        c.voff[i] = (unsigned)(r * dim * esize + lc * 16);
        c.dst[i] = (unsigned)(grp * 1024);
    }
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    c.rd_a = (unsigned)((wm * 64 + l31) * KD_ROW_BYTES);
    c.rd_b = (unsigned)((wn * 64 + l31) * KD_ROW_BYTES);
    c.rd_x = (unsigned)((lh ^ ((l31 >> 1) & 7)) << 4);
    return c;
}

// issue the 8 DMA loads of one K chunk of (A tile, Q tile) into buffer `buf`
__device__ __forceinline__ void dma_issue(const DmaCtx& c, __amdgpu_buffer_rsrc_t a_desc, __amdgpu_buffer_rsrc_t q_desc, char* sA,
                                          char* sB, int buf, int kc) {
    const int soff = kc * KD_ROW_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_desc, (lds_ptr_t)(sA + buf * KD_TILE_BYTES + c.dst[i]), 16, c.voff[i], soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(q_desc, (lds_ptr_t)(sB + buf * KD_TILE_BYTES + c.dst[i]), 16, c.voff[i], soff, 0, 0);
    }
}

template <bool F16, typename Hook>
__device__ __forceinline__ void knn_tile_gemm_dma(const DmaCtx& c, __amdgpu_buffer_rsrc_t cur_desc, __amdgpu_buffer_rsrc_t next_desc,
                                                  bool has_next, __amdgpu_buffer_rsrc_t q_desc, char* sA, char* sB, int& gbuf,
                                                  f32x16 (&acc)[2][2], Hook&& after_first_barrier) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int kc = 0; kc < c.nk; ++kc) {
        const int buf = gbuf & 1;
        if (!RADAD_DBG(c.debug, 2)) {
            if (kc + 1 < c.nk) dma_issue(c, cur_desc, q_desc, sA, sB, buf ^ 1, kc + 1);
            else if (has_next) dma_issue(c, next_desc, q_desc, sA, sB, buf ^ 1, 0);
        }
        const char* a_row = sA + buf * KD_TILE_BYTES + c.rd_a;
        const char* b_row = sB + buf * KD_TILE_BYTES + c.rd_b;
#pragma unroll
        for (int kk = 0; kk < KT_K / 8; ++kk) {
            const unsigned off = c.rd_x ^ (unsigned)(32 * kk);      // chunk (2kk + lh) ^ swizzle, in bytes
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(a_row + off);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(a_row + 32 * KD_ROW_BYTES + off);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b_row + off);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(b_row + 32 * KD_ROW_BYTES + off);
            if constexpr (F16) {
                // the same 16 bytes are 8 halfs: k = 16kk + 8lh + j of the row -- exactly the A/B lane map of
                // v_mfma_f32_32x32x16_f16, so one MFMA (16x the fp32 rate) replaces the four below
                const f16x8 ha0 = __builtin_bit_cast(f16x8, a0), ha1 = __builtin_bit_cast(f16x8, a1);
                const f16x8 hb0 = __builtin_bit_cast(f16x8, b0), hb1 = __builtin_bit_cast(f16x8, b1);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha0, hb0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha0, hb1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb1, acc[1][1], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
                }
            }
        }
        if (!RADAD_DBG(c.debug, 8)) __syncthreads();      // waits vmcnt(0): the DMA issued above has landed for everyone
        ++gbuf;
        if (kc == 0) after_first_barrier();       // work deferred from the previous tile's epilogue (no barriers inside)
    }
}

__device__ __forceinline__ TileCtx make_tile_ctx(int dim, int tid) {
    TileCtx c;
    c.debug = 0;
    c.dim = dim;
    c.nk = (dim + KT_K - 1) / KT_K;
    c.ld_r = tid >> 3;
    c.ld_c = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) c.voff[i] = (unsigned)(((c.ld_r + 32 * i) * dim + c.ld_c) * (int)sizeof(float));
    const int lane = tid & 63, wave = tid >> 6;
    c.wm = wave & 1; c.wn = wave >> 1; c.l31 = lane & 31; c.lh = lane >> 5;
    return c;
}
// descriptor over `rows` rows of `dim` floats starting at `base` (wave-uniform inputs only)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_desc(const void* base, int rows, int dim, int esize = 4) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, rows * dim * esize, 0x00020000);
}

template <bool A_F16>
__global__ __launch_bounds__(KNN_THREADS, 2) void k_knn_f32(KnnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sA = reinterpret_cast<float*>(smem);          // [2][KT_M][KT_LD]
    float* sB = sA + 2 * KT_M * KT_LD;                   // [2][KT_N][KT_LD]
    float* sS = sA;                                      // [KT_N][KS_LD] score tile, aliases both (epilogue only)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1;    // which 64-row half of the store tile
    const int wn = wave >> 1;   // which 64-query half
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    // XCD-aware mapping: blocks b and b+8 share an XCD (round-robin dispatch), so the n_qtiles blocks
    // that stream the SAME store chunk are placed on one XCD and run together -> the chunk is pulled
    // from HBM once and re-served from that XCD's L2.  Speed only; any placement is correct.
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int slot = bid >> 3;
    const int qt = slot % p.n_qtiles;
    const int split = (slot / p.n_qtiles) * 8 + xcd;

    const int q0 = qt * KT_N;
    const int64_t chunk_begin = (int64_t)split * p.chunk_rows;
    const int64_t chunk_end = min(chunk_begin + p.chunk_rows, p.n);
    const int k = p.k;

    // this block's output lists double as the running top-k lists (rarely touched; L2-resident)
    // thread t < 128 owns query q0+t for the whole scan: its threshold stays in a register.
    float my_thr = -INFINITY;
    float* my_ls = nullptr;
    int* my_li = nullptr;
    if (tid < KT_N && q0 + tid < p.nq) {
        my_ls = p.part_score + ((int64_t)(q0 + tid) * p.n_splits + split) * k;
        my_li = p.part_idx + ((int64_t)(q0 + tid) * p.n_splits + split) * k;
        for (int j = 0; j < k; ++j) { my_ls[j] = -INFINITY; my_li[j] = IDX_SENTINEL; }
    }
    if (chunk_begin >= chunk_end) return;

    const TileCtx tc = make_tile_ctx(p.dim, tid);
    const __amdgpu_buffer_rsrc_t q_desc =
        rows_desc(reinterpret_cast<const float*>(p.q) + (int64_t)q0 * p.dim, min(KT_N, p.nq - q0), p.dim);
    constexpr int AES = A_F16 ? 2 : 4;      // bytes per stored element

    for (int64_t row0 = chunk_begin; row0 < chunk_end; row0 += KT_M) {
        f32x16 acc[2][2];
        knn_tile_gemm<A_F16>(tc, rows_desc(reinterpret_cast<const char*>(p.db) + row0 * p.dim * AES,
                                           (int)min((int64_t)KT_M, chunk_end - row0), p.dim, AES), q_desc, sA, sB, acc);

        // ---- fused top-k epilogue -------------------------------------------------------------
        // acc[mt][nt][r] = dot(store row row0 + wm*64 + mt*32 + (r&3) + 8*(r>>2) + 4*lh,
        //                      query     q0   + wn*64 + nt*32 + l31)
        // The tile buffers are free now: park the 128x128 score tile in them, query-major
        // (S[q][row], row stride KS_LD), then ONE thread per query filters its 128 scores against the
        // k-th best it carries in a register and inserts the rare survivors into its list.
        const int rowlimit = (int)min((int64_t)KT_M, chunk_end - row0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rl = wm * 64 + mt * 32 + 8 * g + 4 * lh;   // rows rl..rl+3 = registers 4g..4g+3
                f32x4 v0, v1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float s0 = acc[mt][0][4 * g + i], s1 = acc[mt][1][4 * g + i];
                    if (p.l2) {
                        const float yn = p.ynorm[min(row0 + rl + i, p.n - 1)];
                        s0 = 2.f * s0 - yn;
                        s1 = 2.f * s1 - yn;
                    }
                    v0[i] = s0;
                    v1[i] = s1;
                }
                *reinterpret_cast<f32x4*>(sS + (wn * 64 + l31) * KS_LD + rl) = v0;
                *reinterpret_cast<f32x4*>(sS + (wn * 64 + 32 + l31) * KS_LD + rl) = v1;
            }
        __syncthreads();
        if (tid < KT_N && q0 + tid < p.nq) {
            const float* srow = sS + tid * KS_LD;
            for (int c = 0; c < rowlimit; c += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(srow + c);
                const float vm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                if (vm >= my_thr) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float sc = v[i];
                        const int id = (int)(row0 + c + i);
                        if (c + i < rowlimit && sc >= my_thr && better(sc, id, my_ls[k - 1], my_li[k - 1])) {
                            int pos = k - 1;
                            while (pos > 0 && better(sc, id, my_ls[pos - 1], my_li[pos - 1])) {
                                my_ls[pos] = my_ls[pos - 1];
                                my_li[pos] = my_li[pos - 1];
                                --pos;
                            }
                            my_ls[pos] = sc;
                            my_li[pos] = id;
                            my_thr = my_ls[k - 1];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- scan kernel, register-list variant (k + margin <= KSEL) ---------------------------------------------------
// Same MFMA main loop as k_knn_f32; the fused top-k differs:
//   * thread t < 128 owns query q0+t and keeps its KSEL best (score, row) pairs in REGISTERS as 64-bit sortable
//     keys (ordered-float score in the high word, ~row in the low word: one u64 compare == (score, -row) order);
//     an insertion is a fully unrolled compare/select ladder -- no memory traffic, no pointer chasing.
//   * every lane filters its 64 accumulators against its two queries' thresholds (LDS, 1 float each) and pushes the
//     rare survivors into a per-query LDS slot buffer (CAND_CAP entries, ds_add_rtn for the slot);
//   * a query that overflows its buffer in one tile (always the first tile of a chunk, or adversarially ordered
//     data) takes the exact slow path: the whole 128x128 score tile is parked in LDS and its owner re-reads its row.
constexpr int CAND_CAP = 8;
typedef unsigned long long u64;

__device__ __forceinline__ u64 pack_key(float s, int id) {
    unsigned u = __float_as_uint(s);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ((u64)u << 32) | (u64)(0xFFFFFFFFu - (unsigned)id);
}
__device__ __forceinline__ float key_score(u64 k) {
    unsigned u = (unsigned)(k >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}
__device__ __forceinline__ int key_id(u64 k) { return (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)); }

template <int KSEL>
__device__ __forceinline__ void list_insert(u64 (&l)[KSEL], u64 key) {
    if (key <= l[KSEL - 1]) return;
#pragma unroll
    for (int j = KSEL - 1; j >= 1; --j) {
        const bool up = key > l[j - 1];            // key ranks before entry j-1: that entry moves down to j
        const bool here = !up && key > l[j];       // key lands at j
        l[j] = up ? l[j - 1] : (here ? key : l[j]);
    }
    l[0] = key > l[0] ? key : l[0];
}

template <int KSEL, bool F16>
__global__ __launch_bounds__(KNN_THREADS, 2) void k_knn_f32_reg(KnnParams p) {
    constexpr int ES = F16 ? 2 : 4;          // bytes per element of BOTH operands (fp16 store is scanned with fp16 queries)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                                  // [2][128 rows][128 B], swizzled
    char* sB = smem + 2 * KD_TILE_BYTES;                              // [2][128 rows][128 B], swizzled
    float2* s_cand = reinterpret_cast<float2*>(smem + 4 * KD_TILE_BYTES);   // [KT_N][CAND_CAP] (score, row bits), NOT aliased:
                                                                      // it outlives the tile (drained during the next K loop)
    float* s_thr = reinterpret_cast<float*>(s_cand + KT_N * CAND_CAP);   // [KT_N] current KSEL-th best score of each query
    int* s_cnt = reinterpret_cast<int*>(s_thr + KT_N);                // [KT_N] candidates pushed this tile
    int* s_flag = s_cnt + KT_N;                                       // [1] some query overflowed its slots this tile
    char* sS = smem;                                                  // [KT_N][512 B] swizzled score tile of the overflow path

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const int bid = blockIdx.x;              // XCD-aware mapping: see k_knn_f32
    const int xcd = bid & 7;
    const int slot_b = bid >> 3;
    const int qt = slot_b % p.n_qtiles;
    const int split = (slot_b / p.n_qtiles) * 8 + xcd;

    const int q0 = qt * KT_N;
    const int64_t chunk_begin = (int64_t)split * p.chunk_rows;
    const int64_t chunk_end = min(chunk_begin + p.chunk_rows, p.n);
    const bool owner = tid < KT_N && q0 + tid < p.nq;

    const u64 SENT = pack_key(-INFINITY, IDX_SENTINEL);
    u64 lst[KSEL];
#pragma unroll
    for (int j = 0; j < KSEL; ++j) lst[j] = SENT;
    if (tid < KT_N) { s_thr[tid] = -INFINITY; s_cnt[tid] = 0; }
    if (tid == 0) *s_flag = 0;
    __syncthreads();

    const DmaCtx dc = make_dma_ctx(p.dim, tid, p.debug, ES);
    const __amdgpu_buffer_rsrc_t q_desc =
        rows_desc(reinterpret_cast<const char*>(p.q) + (int64_t)q0 * p.dim * ES, min(KT_N, p.nq - q0), p.dim, ES);
    const int ql[2] = {wn * 64 + l31, wn * 64 + 32 + l31};
    const bool qvalid[2] = {q0 + ql[0] < p.nq, q0 + ql[1] < p.nq};

    auto tile_desc = [&](int64_t r0) {
        return rows_desc(reinterpret_cast<const char*>(p.db) + r0 * p.dim * ES,
                         (int)max((int64_t)0, min((int64_t)KT_M, chunk_end - r0)), p.dim, ES);
    };
    int gbuf = 0;
    if (chunk_begin < chunk_end) {
        dma_issue(dc, tile_desc(chunk_begin), q_desc, sA, sB, 0, 0);
        __syncthreads();
    }
    // owners fold their slot buffer into the register list; no barrier inside, so it can run in the shadow of the
    // next tile's MFMA work (the pushes it reads were published by the epilogue's barrier)
    auto drain = [&]() {
        if (owner) {
            const int c = min(s_cnt[tid], CAND_CAP);
            if (c > 0) {
                for (int i = 0; i < c; ++i) {
                    const float2 cv = s_cand[tid * CAND_CAP + i];
                    list_insert<KSEL>(lst, pack_key(cv.x, __float_as_int(cv.y)));
                }
                s_thr[tid] = key_score(lst[KSEL - 1]);
            }
        }
        if (tid < KT_N) s_cnt[tid] = 0;
    };
    bool pending = false;
    for (int64_t row0 = chunk_begin; row0 < chunk_end; row0 += KT_M) {
        f32x16 acc[2][2];
        const bool has_next = row0 + KT_M < chunk_end;
        knn_tile_gemm_dma<F16>(dc, tile_desc(row0), tile_desc(row0 + KT_M), has_next, q_desc, sA, sB, gbuf, acc,
                          [&]() { if (pending) { drain(); pending = false; } });
        if (RADAD_DBG(p.debug, 1)) {      // timing experiment: keep the accumulators alive, skip the epilogue
            if (acc[0][0][0] + acc[0][1][5] + acc[1][0][9] + acc[1][1][15] == 12345.678f) s_thr[tid & 127] = 1.f;
            continue;
        }
        // ---- fused top-k epilogue -------------------------------------------------------------------------
        const int rowlimit = (int)min((int64_t)KT_M, chunk_end - row0);
        if (p.l2) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float yn = p.ynorm[min(row0 + rl, p.n - 1)];
                    acc[mt][0][r] = 2.f * acc[mt][0][r] - yn;
                    acc[mt][1][r] = 2.f * acc[mt][1][r] - yn;
                }
        }
        // phase 1: every lane filters its 2 x 32 scores and pushes survivors to its queries' slot buffers
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float thr = s_thr[ql[nt]];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float gm = acc[mt][nt][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) gm = fmaxf(gm, acc[mt][nt][r]);
                if (qvalid[nt] && gm >= thr) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (rl < rowlimit && acc[mt][nt][r] >= thr) {
                            const int sl = atomicAdd(&s_cnt[ql[nt]], 1);
                            if (sl < CAND_CAP)
                                s_cand[ql[nt] * CAND_CAP + sl] = make_float2(acc[mt][nt][r], __int_as_float((int)(row0 + rl)));
                            else
                                *s_flag = 1;
                        }
                    }
                }
            }
        }
        __syncthreads();                 // pushes, counters and the overflow flag are visible to everyone
        if (*s_flag == 0) {
            // common case: nothing overflowed.  The drain is deferred into the next tile's K loop (or done right here
            // for the last tile); no second barrier, the next K loop's own barriers order everything.
            if (has_next && dc.nk >= 2) pending = true;     // nk == 1: the only K-loop barrier precedes the drain hook
            else drain();
            continue;
        }
        // overflow path (exact, slower; always the first tile of a chunk): owners whose slots sufficed drain them, the
        // others re-read their whole row from the score tile, parked query-major over the tile buffers (512-byte rows,
        // 16-byte chunk c of query q at chunk c ^ (q & 15): conflict-free b128 writes and reads).
        const bool over = owner && s_cnt[tid] > CAND_CAP;
        if (!over) drain();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = (wm * 64 + mt * 32 + 8 * g + 4 * lh) >> 2;   // rows 4ch..4ch+3 = registers 4g..4g+3
                f32x4 v0, v1;
#pragma unroll
                for (int i = 0; i < 4; ++i) { v0[i] = acc[mt][0][4 * g + i]; v1[i] = acc[mt][1][4 * g + i]; }
                *reinterpret_cast<f32x4*>(sS + ql[0] * 512 + ((ch ^ (ql[0] & 15)) << 4)) = v0;
                *reinterpret_cast<f32x4*>(sS + ql[1] * 512 + ((ch ^ (ql[1] & 15)) << 4)) = v1;
            }
        __syncthreads();
        if (tid == 0) *s_flag = 0;
        if (over) {
            for (int c = 0; c < rowlimit; c += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sS + tid * 512 + (((c >> 2) ^ (tid & 15)) << 4));
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (c + i < rowlimit) list_insert<KSEL>(lst, pack_key(v[i], (int)(row0 + c + i)));
            }
            s_thr[tid] = key_score(lst[KSEL - 1]);
            s_cnt[tid] = 0;
        }
        __syncthreads();
        if (has_next) {                  // the score tile overwrote the prefetched chunk: fetch it again
            dma_issue(dc, tile_desc(row0 + KT_M), q_desc, sA, sB, gbuf & 1, 0);
            __syncthreads();
        }
    }
    // the lists leave the registers once, at the end of the chunk
    if (owner) {
        float* ls = p.part_score + ((int64_t)(q0 + tid) * p.n_splits + split) * p.k;
        int* li = p.part_idx + ((int64_t)(q0 + tid) * p.n_splits + split) * p.k;
#pragma unroll
        for (int j = 0; j < KSEL; ++j)
            if (j < p.k) { ls[j] = key_score(lst[j]); li[j] = key_id(lst[j]); }
    }
}

#include "knn_hi.inc"

// ---- small-batch scan (nq <= 16: the online `predict` case, pipeline.py:1038-1054) ------------------------------
// With a handful of queries the scan is HBM-bound (2 flop per stored byte per query), so the 128-query tile above
// would burn 8x-128x the needed MFMA time.  Here every WAVE streams its own slice of the store, 16 rows per step,
// whole 2 KB rows in one burst of 16-byte loads straight from HBM into registers (no LDS for the store: it is read exactly once), against the <=16 queries parked in LDS.
// v_mfma_f32_16x16x4_f32 with the store rows as A and the queries as B leaves a lane with ONE query column
// (lane & 15) and 4 rows per group, so the filter threshold is a register.  Each wave keeps private top-k lists
// (lanes 0-15 own one query each, lists in registers); survivors travel through a per-wave LDS slot buffer that can
// never overflow (16 slots per query = everything one step can produce).  No workgroup barrier inside the scan.
constexpr int SQ_THREADS = 256;
constexpr int SQ_NQ = 16;
constexpr int SQ_SLOTS = 16;
constexpr size_t SQ_LDS_BUDGET = 160 * 1024;      // the query block [nq][dim + pad] + slot buffers must fit one CU's LDS

struct SmallQParams {
    const float* db; const float* ynorm; const float* q;      // fp32 store only
    int64_t n;
    int nq, dim, k, l2;
    int rows_per_wave;          // multiple of 16
    int n_parts;                // workgroups = lists per query
    float* part_score; int* part_idx;   // [nq, n_parts, k]
};

template <int KSEL>
__global__ __launch_bounds__(SQ_THREADS, 2) void k_knn_f32_smallq(SmallQParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int qld = p.dim + 4;                                    // padded query row: conflict-free b128 reads
    float* sQ = reinterpret_cast<float*>(smem);                   // [nq][dim + 4]: only the queries there are (the reference's
                                                                  // predict() search is ONE query of dim 5376 / 3584)
    float2* sCand = reinterpret_cast<float2*>(sQ + p.nq * qld);   // [4 waves][16 q][SQ_SLOTS]
    int* sCnt = reinterpret_cast<int*>(sCand + 4 * SQ_NQ * 24);         // [4 waves][16]; the slot area is sized for the final
                                                                        // hand-over of 3 x 16 x 32 keys (12 KB) as well

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < p.nq * (p.dim >> 2); i += SQ_THREADS) {
        const int qq = i / (p.dim >> 2), c4 = i % (p.dim >> 2);
        *reinterpret_cast<f32x4*>(sQ + qq * qld + c4 * 4) = *reinterpret_cast<const f32x4*>(p.q + (int64_t)qq * p.dim + c4 * 4);
    }
    if (tid < 4 * SQ_NQ) sCnt[tid] = 0;
    __syncthreads();

    const int part = blockIdx.x * 4 + wave;      // this wave's slice of the store
    const int64_t w_begin = (int64_t)part * p.rows_per_wave;
    const int64_t w_end = min(w_begin + p.rows_per_wave, p.n);
    float2* myCand = sCand + wave * SQ_NQ * SQ_SLOTS;
    int* myCnt = sCnt + wave * SQ_NQ;

    const u64 SENT = pack_key(-INFINITY, IDX_SENTINEL);
    u64 lst[KSEL];                       // meaningful in lanes 0..15 (lane = query)
#pragma unroll
    for (int j = 0; j < KSEL; ++j) lst[j] = SENT;
    float thr = -INFINITY;               // threshold of query (lane & 15), refreshed from the owner lane
    // K mapping inside a 32-float block: load h (0/1), element j of lane group g is k = 16h + 4g + j, so one load
    // instruction reads 64 CONTIGUOUS bytes of each of its 16 rows and the pair covers the whole 128-byte line.
    const float* qrow = sQ + min(r16, p.nq - 1) * qld + 4 * g;     // columns >= nq repeat the last query (never admitted below)
    const int nkb = p.dim >> 5;

    for (int64_t row0 = w_begin; row0 < w_end; row0 += 16) {
        // 16 rows per step; a lane issues ALL its loads of a 512-float panel (32 x 16 B = the whole 2 KB row at
        // dim 512) before the first MFMA, so each row is fetched in one burst and 32 loads per lane are in flight.
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};       // even / odd K blocks: two MFMA chains
        const int64_t ra = min(row0 + r16, p.n - 1);                             // clamp: masked below
        const float* pa = p.db + ra * p.dim + 4 * g;
        constexpr int PKB = KSEL <= 16 ? 16 : 8;      // K blocks per panel (the 32-entry lists leave fewer registers)
        for (int kp = 0; kp < nkb; kp += PKB) {
            const int nb = min(PKB, nkb - kp);
            f32x4 v[2 * PKB];
#pragma unroll
            for (int kb = 0; kb < PKB; ++kb)
                if (kb < nb) {
                    v[2 * kb] = *reinterpret_cast<const f32x4*>(pa + (kp + kb) * 32);
                    v[2 * kb + 1] = *reinterpret_cast<const f32x4*>(pa + (kp + kb) * 32 + 16);
                }
#pragma unroll
            for (int kb = 0; kb < PKB; ++kb)
                if (kb < nb) {
                    const f32x4 q0 = *reinterpret_cast<const f32x4*>(qrow + (kp + kb) * 32);
                    const f32x4 q1 = *reinterpret_cast<const f32x4*>(qrow + (kp + kb) * 32 + 16);
                    if (kb & 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[2 * kb][j], q0[j], acc1, 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[2 * kb + 1][j], q1[j], acc1, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[2 * kb][j], q0[j], acc0, 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[2 * kb + 1][j], q1[j], acc0, 0, 0, 0);
                    }
                }
        }
        // acc0[e] + acc1[e]: row row0 + 4g + e, query r16
        bool any = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t row = row0 + 4 * g + e;
            float sc = acc0[e] + acc1[e];
            if (p.l2) sc = 2.f * sc - p.ynorm[min(row, p.n - 1)];
            if (row < w_end && r16 < p.nq && sc >= thr) {
                const int sl = atomicAdd(&myCnt[r16], 1);          // < SQ_SLOTS by construction (4 x 4 lanes)
                myCand[r16 * SQ_SLOTS + sl] = make_float2(sc, __int_as_float((int)row));
                any = true;
            }
        }
        if (__any(any)) {                                              // wave-uniform; LDS ops of one wave are ordered
            if (lane < SQ_NQ) {
                const int c = myCnt[lane];
                for (int i = 0; i < c; ++i) {
                    const float2 cv = myCand[lane * SQ_SLOTS + i];
                    list_insert<KSEL>(lst, pack_key(cv.x, __float_as_int(cv.y)));
                }
                myCnt[lane] = 0;
            }
            thr = __shfl(key_score(lst[KSEL - 1]), r16, 64);
        }
    }
    // one list per WORKGROUP leaves the kernel: waves 1-3 hand their lists to wave 0 through the slot buffer
    __syncthreads();
    u64* sKeys = reinterpret_cast<u64*>(sCand);                     // [3][16][KSEL] (<= the slot buffer's 16 KB)
    if (wave > 0 && lane < SQ_NQ) {
#pragma unroll
        for (int j = 0; j < KSEL; ++j) sKeys[((wave - 1) * SQ_NQ + lane) * KSEL + j] = lst[j];
    }
    __syncthreads();
    if (wave == 0 && lane < p.nq) {
        for (int w = 0; w < 3; ++w)
            for (int j = 0; j < KSEL; ++j) list_insert<KSEL>(lst, sKeys[(w * SQ_NQ + lane) * KSEL + j]);
        float* ls = p.part_score + ((int64_t)lane * p.n_parts + blockIdx.x) * p.k;
        int* li = p.part_idx + ((int64_t)lane * p.n_parts + blockIdx.x) * p.k;
#pragma unroll
        for (int j = 0; j < KSEL; ++j)
            if (j < p.k) { ls[j] = key_score(lst[j]); li[j] = key_id(lst[j]); }
    }
}

// ---- a SMALL store (<= RF_STAGE_MAX rows: the IVF index's centroids, vector_database.py:65-70; a database of a few thousand files):
// all scores, no lists.  The register-list kernels keep k + margin <= 32 candidates per (query, row slice) in sorted lists -- with
// 4096 rows over 16 workgroups nearly every row is an insertion (k_knn_f32_smallq<32>: 0.14 ms for ONE query against 4096 x 512,
// of which the 8 MB of rows are 1 us).  Here every (row, query) score is written out -- v_mfma_f32_16x16x4_f32 on the fp32
// operands themselves, a wave per 16 rows x 16 queries -- as the query's candidate buffer in the emit-mode layout of the
// certified tile scan (score [nq][plen], idx [nq][plen], cnt [nq] = n), and k_merge_refine<true> selects on its staged copy,
// re-scores in float64 and certifies against eps(q) of the fp32 products (k_hi_rows, exact_ops).
struct DenseParams {
    const float* db; const float* ynorm; const float* q;      // fp32 store only
    int64_t n;
    int nq, dim, l2;
    int plen;                   // row stride of score / idx: a multiple of 4, >= n
    float* score; int* idx; int* cnt;
};

// <RT, QT>: 16-row x 16-query tiles per wave.  <1, 1> for the batches of <= 16 queries (one wave per 16 rows: parallel over the rows);
// <2, 4> otherwise: 32 rows x 64 queries per wave -- every operand a wave loads feeds 4 resp. 2 MFMAs (<1, 1> on 1024 queries x 4096
// centroids pulled 1 GB through the L2s for 4 GFLOP: 0.122 ms, the L2 bandwidth).
template <int RT, int QT>
__global__ __launch_bounds__(256) void k_knn_dense(DenseParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * (64 * RT) + wave * (16 * RT);
    const int q0 = blockIdx.y * (16 * QT);
    if (row0 >= p.plen) return;
    const float* pa[RT];
    const float* pb[QT];
#pragma unroll
    for (int r = 0; r < RT; ++r) pa[r] = p.db + min(row0 + 16 * r + r16, p.n - 1) * p.dim + 4 * g;
#pragma unroll
    for (int t = 0; t < QT; ++t) pb[t] = p.q + (int64_t)min(q0 + 16 * t + r16, p.nq - 1) * p.dim + 4 * g;
    constexpr int NA = RT * QT == 1 ? 2 : 1;           // accumulators per tile (one tile alone: two chains hide the MFMA latency)
    constexpr int KB = RT * QT == 1 ? 4 : 2;           // 16-element K blocks per panel; two panels in registers: the next one's loads
                                                       // are in flight while the current one's MFMAs issue
    f32x4 acc[RT][QT][NA];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int c = 0; c < NA; ++c) acc[r][t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkb = p.dim >> 4;
    f32x4 a0[RT][KB], b0[QT][KB], a1[RT][KB], b1[QT][KB];
    auto load = [&](f32x4 (&a)[RT][KB], f32x4 (&b)[QT][KB], int kp) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int kb = min(kp + u, nkb - 1);       // (a panel past the end re-reads the last block; its products are not used)
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r][u] = *reinterpret_cast<const f32x4*>(pa[r] + kb * 16);
#pragma unroll
            for (int t = 0; t < QT; ++t) b[t][u] = *reinterpret_cast<const f32x4*>(pb[t] + kb * 16);
        }
    };
    auto mma = [&](const f32x4 (&a)[RT][KB], const f32x4 (&b)[QT][KB], int kp) {
#pragma unroll
        for (int u = 0; u < KB; ++u)
            if (kp + u < nkb) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int t = 0; t < QT; ++t)
                            acc[r][t][u % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][u][j], b[t][u][j], acc[r][t][u % NA], 0, 0, 0);
            }
    };
    load(a0, b0, 0);
    for (int kp = 0; kp < nkb; kp += 2 * KB) {
        load(a1, b1, kp + KB);
        mma(a0, b0, kp);
        load(a0, b0, kp + 2 * KB);
        mma(a1, b1, kp + KB);
    }
    // acc[r][t][.][e]: row row0 + 16 r + 4 g + e, query q0 + 16 t + r16
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qq = q0 + 16 * t + r16;
        if (qq >= p.nq) continue;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int64_t rb = row0 + 16 * r + 4 * g;
            f32x4 sc;
            int id[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = rb + e;
                float d = acc[r][t][0][e];
                if (NA == 2) d += acc[r][t][NA - 1][e];
                const bool on = row < p.n;
                sc[e] = on ? (p.l2 ? 2.f * d - p.ynorm[row] : d) : -INFINITY;
                id[e] = on ? (int)row : IDX_SENTINEL;
            }
            if (rb < p.plen) {
                *reinterpret_cast<f32x4*>(p.score + (int64_t)qq * p.plen + rb) = sc;
                *reinterpret_cast<int4*>(p.idx + (int64_t)qq * p.plen + rb) = make_int4(id[0], id[1], id[2], id[3]);
            }
        }
        if (blockIdx.x == 0 && wave == 0 && g == 0) p.cnt[qq] = (int)p.n;
    }
}

// ---- the same small-batch scan over the f16 plane (certified mode) -----------------------------------------------------
// Half the bytes: the hi plane of an fp32 store (or an fp16 store itself) is streamed instead of the fp32 rows, one
// v_mfma_f32_16x16x32_f16 per 32 elements of 16 rows against the f16-rounded queries in LDS.  The scores carry the error bound
// eps(q) of the certified scan (k_hi_rows), and k_merge_refine certifies or rejects every query exactly as it does for the
// tile kernel (complete top-(k + margin) lists per workgroup, no admission floor): the online search reads 1 GB instead of 2.
struct SmallQHiParams {
    const _Float16* db;         // [n][dim] f16 rows
    const float* rscale;        // [n] per-row scale, or nullptr: uscale for every row
    float uscale;
    const float* rbias;         // [n] per-row bias magnitude (|y'|^2 for L2, mu.y for a centred IP / cosine plane), or nullptr
    float bias_sign, mult;      // score = mult a + bias_sign rbias[row] + qconst[query]   (knn_hi.inc, RSC 2)
    const float* qconst;        // [nq] or nullptr
    const _Float16* q;          // [nq][dim] f16 queries (scaled per query; centred when the plane is)
    const float* qscale;        // [nq]
    int64_t n;
    int nq, dim, k;
    int rows_per_wave;          // multiple of 16
    int n_parts;                // workgroups = lists per query
    float* part_score; int* part_idx;   // [nq, n_parts, k]
};

// (launch bounds: 155 VGPRs = 3 waves per SIMD for KSEL 16; forcing 4 (128 VGPRs) spills the 16-load panel: 0.41 instead of 0.21 ms)
template <int KSEL>
__global__ __launch_bounds__(SQ_THREADS, 2) void k_knn_hi_smallq(SmallQHiParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int qld = p.dim + 8;                                    // padded query row (halfs): 16-byte chunks of the 16 rows fall in distinct banks
    _Float16* sQ = reinterpret_cast<_Float16*>(smem);             // [nq][dim + 8]
    float2* sCand = reinterpret_cast<float2*>(sQ + p.nq * qld);   // [4 waves][16 q][SQ_SLOTS]
    int* sCnt = reinterpret_cast<int*>(sCand + 4 * SQ_NQ * 24);   // [4 waves][16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < p.nq * (p.dim >> 3); i += SQ_THREADS) {
        const int qq = i / (p.dim >> 3), c8 = i % (p.dim >> 3);
        *reinterpret_cast<f16x8*>(sQ + qq * qld + c8 * 8) = *reinterpret_cast<const f16x8*>(p.q + (int64_t)qq * p.dim + c8 * 8);
    }
    if (tid < 4 * SQ_NQ) sCnt[tid] = 0;
    __syncthreads();

    const int part = blockIdx.x * 4 + wave;      // this wave's slice of the store
    const int64_t w_begin = (int64_t)part * p.rows_per_wave;
    const int64_t w_end = min(w_begin + p.rows_per_wave, p.n);
    float2* myCand = sCand + wave * SQ_NQ * SQ_SLOTS;
    int* myCnt = sCnt + wave * SQ_NQ;

    const u64 SENT = pack_key(-INFINITY, IDX_SENTINEL);
    u64 lst[KSEL];                       // meaningful in lanes 0..15 (lane = query)
#pragma unroll
    for (int j = 0; j < KSEL; ++j) lst[j] = SENT;
    float thr = -INFINITY;               // threshold of query (lane & 15), refreshed from the owner lane
    const float qs = r16 < p.nq ? p.qscale[r16] * p.mult : 0.f;
    const float qc = (r16 < p.nq && p.qconst) ? p.qconst[r16] : 0.f;
    // lane (r16, g) holds elements 32 kb + 8 g .. + 7 of row r16 (A) and of query r16 (B): one load instruction reads 64
    // contiguous bytes of each of its 16 rows, two consecutive ones a whole 128-byte line
    const _Float16* qrow = sQ + min(r16, p.nq - 1) * qld + 8 * g;   // columns >= nq repeat the last query (never admitted below)
    const int nkb = p.dim >> 5;

    for (int64_t row0 = w_begin; row0 < w_end; row0 += 16) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};       // even / odd K blocks: two MFMA chains
        const int64_t ra = min(row0 + r16, p.n - 1);                             // clamp: masked below
        const _Float16* pa = p.db + ra * p.dim + 8 * g;
        float rs[4], yn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t row = min(row0 + 4 * g + e, p.n - 1);
            rs[e] = p.rscale ? p.rscale[row] : p.uscale;
            yn[e] = p.rbias ? p.bias_sign * p.rbias[row] + qc : qc;
        }
        constexpr int PKB = KSEL <= 16 ? 16 : 8;      // K blocks per panel (the 32-entry lists leave fewer registers)
        for (int kp = 0; kp < nkb; kp += PKB) {
            const int nb = min(PKB, nkb - kp);
            f16x8 v[PKB];
#pragma unroll
            for (int kb = 0; kb < PKB; ++kb)
                if (kb < nb) v[kb] = *reinterpret_cast<const f16x8*>(pa + (kp + kb) * 32);
#pragma unroll
            for (int kb = 0; kb < PKB; ++kb)
                if (kb < nb) {
                    const f16x8 b = *reinterpret_cast<const f16x8*>(qrow + (kp + kb) * 32);
                    if (kb & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v[kb], b, acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v[kb], b, acc0, 0, 0, 0);
                }
        }
        // acc0[e] + acc1[e]: row row0 + 4g + e, query r16
        bool any = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t row = row0 + 4 * g + e;
            const float sc = fmaf((acc0[e] + acc1[e]) * rs[e], qs, yn[e]);
            if (row < w_end && r16 < p.nq && sc >= thr) {
                const int sl = atomicAdd(&myCnt[r16], 1);          // < SQ_SLOTS by construction (4 x 4 lanes)
                myCand[r16 * SQ_SLOTS + sl] = make_float2(sc, __int_as_float((int)row));
                any = true;
            }
        }
        if (__any(any)) {                                              // wave-uniform; LDS ops of one wave are ordered
            if (lane < SQ_NQ) {
                const int c = myCnt[lane];
                for (int i = 0; i < c; ++i) {
                    const float2 cv = myCand[lane * SQ_SLOTS + i];
                    list_insert<KSEL>(lst, pack_key(cv.x, __float_as_int(cv.y)));
                }
                myCnt[lane] = 0;
            }
            thr = __shfl(key_score(lst[KSEL - 1]), r16, 64);
        }
    }
    // one list per WORKGROUP leaves the kernel: waves 1-3 hand their lists to wave 0 through the slot buffer
    __syncthreads();
    u64* sKeys = reinterpret_cast<u64*>(sCand);                     // [3][16][KSEL] (<= the slot buffer's 16 KB)
    if (wave > 0 && lane < SQ_NQ) {
#pragma unroll
        for (int j = 0; j < KSEL; ++j) sKeys[((wave - 1) * SQ_NQ + lane) * KSEL + j] = lst[j];
    }
    __syncthreads();
    if (wave == 0 && lane < p.nq) {
        for (int w = 0; w < 3; ++w)
            for (int j = 0; j < KSEL; ++j) list_insert<KSEL>(lst, sKeys[(w * SQ_NQ + lane) * KSEL + j]);
        float* ls = p.part_score + ((int64_t)lane * p.n_parts + blockIdx.x) * p.k;
        int* li = p.part_idx + ((int64_t)lane * p.n_parts + blockIdx.x) * p.k;
#pragma unroll
        for (int j = 0; j < KSEL; ++j)
            if (j < p.k) { ls[j] = key_score(lst[j]); li[j] = key_id(lst[j]); }
    }
}

// ---- the same streaming scan, K SPLIT over the four waves of a workgroup (small stores of wide rows) ---------------------------
// The reference's own online search is ONE query of dim 5376 against 25 423 rows (pipeline.py:1038-1054, config.py:48-56): 1 589
// 16-row steps in all.  With a wave per row slice that is at most 1 589 waves (6 per CU) each streaming 10.5 KB rows: too few loads
// in flight to cover the HBM latency (measured 2.0 TB/s).  Here the four waves of a workgroup take a quarter of the row's elements
// each for the SAME 16 rows, so every 16-row step keeps 4 waves busy (6 356 wave-steps over the store); their partial products
// meet in LDS, are added in a fixed order (deterministic) by wave 0, which also keeps the workgroup's top-k lists.
template <int KSEL>
__global__ __launch_bounds__(SQ_THREADS, KSEL <= 16 ? 3 : 2) void k_knn_hi_smallq_ksplit(SmallQHiParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int qld = p.dim + 8;
    _Float16* sQ = reinterpret_cast<_Float16*>(smem);             // [nq][dim + 8]
    float2* sCand = reinterpret_cast<float2*>(sQ + p.nq * qld);   // [16 q][SQ_SLOTS] (wave 0's)
    int* sCnt = reinterpret_cast<int*>(sCand + 4 * SQ_NQ * 24);   // [16]
    f32x4* sPart = reinterpret_cast<f32x4*>(sCand + SQ_NQ * SQ_SLOTS);   // [2 parities][3 waves][64 lanes]: partial sums of waves 1-3 (inside the slot area)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < p.nq * (p.dim >> 3); i += SQ_THREADS) {
        const int qq = i / (p.dim >> 3), c8 = i % (p.dim >> 3);
        *reinterpret_cast<f16x8*>(sQ + qq * qld + c8 * 8) = *reinterpret_cast<const f16x8*>(p.q + (int64_t)qq * p.dim + c8 * 8);
    }
    if (tid < SQ_NQ) sCnt[tid] = 0;
    __syncthreads();

    const int64_t w_begin = (int64_t)blockIdx.x * p.rows_per_wave;      // (here: rows per WORKGROUP)
    const int64_t w_end = min(w_begin + p.rows_per_wave, p.n);
    const u64 SENT = pack_key(-INFINITY, IDX_SENTINEL);
    u64 lst[KSEL];                       // wave 0, lanes 0..15 (lane = query)
#pragma unroll
    for (int j = 0; j < KSEL; ++j) lst[j] = SENT;
    float thr = -INFINITY;
    const float qs = r16 < p.nq ? p.qscale[r16] * p.mult : 0.f;
    const float qc = (r16 < p.nq && p.qconst) ? p.qconst[r16] : 0.f;
    const int nkb = p.dim >> 5;
    const int kb0 = (nkb * wave) >> 2, kb1 = (nkb * (wave + 1)) >> 2;       // this wave's K blocks
    const _Float16* qrow = sQ + min(r16, p.nq - 1) * qld + 8 * g;
    int par = 0;
    for (int64_t row0 = w_begin; row0 < w_end; row0 += 16, par ^= 1) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const int64_t ra = min(row0 + r16, p.n - 1);
        const _Float16* pa = p.db + ra * p.dim + 8 * g;
        constexpr int PKB = 16;
        for (int kp = kb0; kp < kb1; kp += PKB) {
            const int nb = min(PKB, kb1 - kp);
            f16x8 v[PKB];
#pragma unroll
            for (int kb = 0; kb < PKB; ++kb)
                if (kb < nb) v[kb] = *reinterpret_cast<const f16x8*>(pa + (kp + kb) * 32);
#pragma unroll
            for (int kb = 0; kb < PKB; ++kb)
                if (kb < nb) {
                    const f16x8 b = *reinterpret_cast<const f16x8*>(qrow + (kp + kb) * 32);
                    if (kb & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v[kb], b, acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v[kb], b, acc0, 0, 0, 0);
                }
        }
        f32x4 sum = acc0 + acc1;
        if (wave > 0) sPart[(par * 3 + wave - 1) * 64 + lane] = sum;
        __syncthreads();                 // (one barrier per step: the partial buffers alternate, so step t + 1's writes cannot
                                         // overtake wave 0's reads of step t)
        if (wave == 0) {
            sum = ((sum + sPart[(par * 3 + 0) * 64 + lane]) + sPart[(par * 3 + 1) * 64 + lane]) + sPart[(par * 3 + 2) * 64 + lane];
            bool any = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = row0 + 4 * g + e;
                const int64_t rc = min(row, p.n - 1);
                const float rs = p.rscale ? p.rscale[rc] : p.uscale;
                const float yn = p.rbias ? p.bias_sign * p.rbias[rc] + qc : qc;
                const float sc = fmaf(sum[e] * rs, qs, yn);
                if (row < w_end && r16 < p.nq && sc >= thr) {
                    const int sl = atomicAdd(&sCnt[r16], 1);
                    sCand[r16 * SQ_SLOTS + sl] = make_float2(sc, __int_as_float((int)row));
                    any = true;
                }
            }
            if (__any(any)) {
                if (lane < SQ_NQ) {
                    const int c = sCnt[lane];
                    for (int i = 0; i < c; ++i) {
                        const float2 cv = sCand[lane * SQ_SLOTS + i];
                        list_insert<KSEL>(lst, pack_key(cv.x, __float_as_int(cv.y)));
                    }
                    sCnt[lane] = 0;
                }
                thr = __shfl(key_score(lst[KSEL - 1]), r16, 64);
            }
        }
    }
    if (wave == 0 && lane < p.nq) {
        float* ls = p.part_score + ((int64_t)lane * p.n_parts + blockIdx.x) * p.k;
        int* li = p.part_idx + ((int64_t)lane * p.n_parts + blockIdx.x) * p.k;
#pragma unroll
        for (int j = 0; j < KSEL; ++j)
            if (j < p.k) { ls[j] = key_score(lst[j]); li[j] = key_id(lst[j]); }
    }
}

constexpr size_t knn_reg_lds_bytes() { return 4 * KD_TILE_BYTES + sizeof(float2) * KT_N * CAND_CAP + sizeof(float) * KT_N + sizeof(int) * KT_N + 16; }

// ---- merge of sorted partial lists -----------------------------------------------------------------------
// element j of partial list `part` of query q sits at [(q*n_parts + part)*part_len + j]; lists are sorted best-first by
// (score, -row) and padded with sentinels.  One wave per query; each lane walks the heads of lists lane, lane+64, ...
//
// k_merge_refine (inside radad_knn_search): the scan is a FILTER, this kernel decides.  With a = the scan's score and
//   eps(q) a bound on |a - exact score| (k_hi_rows), a_k the k-th best a over all lists:
//     * every row of the exact top-k has a >= tau = a_k - 2 eps   (the k rows with the best a have exact scores >= a_k - eps,
//       so the exact k-th score is >= a_k - eps, and a row that reaches it has a >= a_k - 2 eps);
//     * so all listed rows with a >= tau are RE-SCORED in float64 straight from the stored rows (L2 as sum (q-y)^2, no
//       cancellation) and ranked by (float64 distance, id);
//     * the result is CERTIFIED to be the exact float64 brute force when no unlisted row can have a >= tau: no list was
//       used up by the selection while full (its tail could hide such rows), the scan's admission floor thr_init is
//       <= tau (rows below the floor were never listed), the scan dropped nothing for this query (qflag), and the
//       candidate buffer (cap) sufficed.  Uncertified queries are appended to flag_sel for the exact kernel below.
//   eps == nullptr (IVF list scans, k > 128): legacy behaviour -- the `cap` best candidates are re-scored, no certificate.
// k_merge_lists<KeyT> (radad_topk_merge / _f64): plain P-way merge of final per-shard lists, no rescoring.
constexpr int KNN_MARGIN = 6;            // spare entries of a (query, chunk) list on the fp32 tile kernels
constexpr int KNN_CERT_EXTRA = 32;       // candidates beyond k the certified re-rank can take before it gives up, at least ...
constexpr int KNN_CERT_CAP = 512;        // ... and this many in all: stores of near-duplicates (the benchmark plants 2048 rows
                                         // within 2e-2 of every query) put hundreds of rows within 2 eps of the k-th
constexpr int KNN_CERT_MAX_K = 128;      // largest k the certificate + exact kernel cover
constexpr int KNN_PROG_OFF = 64;         // the tile scan's progress counters (one per query tile) sit behind the 256-byte flag_count block ...
constexpr int KNN_PROG_MAX = 4096;       // ... at most this many query tiles (1 M queries) raise their floors inside one launch
constexpr int KW_SAMPLE_SPLITS = 64;     // one-tile splits of the threshold pre-pass (<= 16384 rows)

struct RefineParams {
    const float* score;       // [nq, n_parts, part_len] fp32 scan scores (larger is better), each list sorted
    const int* idx;           // [nq, n_parts, part_len] local row or IDX_SENTINEL
    int n_parts, k, dim, l2;
    int part_len;             // entries per partial list
    int cap;                  // candidates that can be re-scored per query (>= k)
    const float* eps;         // [nq] error bound of the scan scores, or nullptr = legacy mode (exactly `cap` candidates)
    const float* thr_init;    // optional [nq]: admission floor of the scan (rows below it were never listed)
    const int* qflag;         // optional [nq]: the scan dropped a candidate of this query
    const int* part_cnt;      // optional [nq] (emit-mode scan, n_parts == 1): entries filled in the query's buffer; > part_len = overflow
    const float* global_lb;   // optional [nq] (sharded search): the best lower bound any shard has of the exact k-th best score
    const float* ak_in;       // optional [nq] (emit-mode scan, two-half search): a_k as k_kth_floor found it for the bounds
    int* flag_count;          // certified mode: [1] number of uncertified queries (atomicAdd) ...
    int* flag_sel;            // ... and their indices, in arrival order
    int* stats;               // optional [5]: sum of candidates re-scored, queries rejected for: buffer full / list used up /
                              // floor above tau / dropped by the scan
    int64_t nq;
    const void* db;           // stored rows (normalised for cosine); fp16 when db_f16
    int db_f16;
    const float* q;           // the fp32 queries (normalised for cosine)
    const int64_t* id_map;    // optional: reported id = id_map[row] (IVF: list-sorted position -> insertion id), ties by it
    int64_t id_base;
    float* out_dist;          // [nq, k]
    int64_t* out_idx;         // [nq, k]
    double* out_key;          // optional [nq, k] float64 distances
    int debug;                // timing experiments only (-DRADAD_DEBUG_HOOKS, RADAD_DEBUG_KNN): 128 no statistics atomics, 256 no re-score, 512 no ranking, 1024 no fp32 funnel
    int* qflag_out = nullptr;           // optional [nq] (certified mode): the certificate's verdict per query, 0 = certified (IVF: who takes the fp32 pass)
    const int* only_flagged = nullptr;  // optional [nq]: workgroups of queries whose entry is 0 leave at once (IVF: the fp32 pass of the rejected)
};

// One workgroup (256 threads) per query: thread t looks after lists t, t + 256, ... (<= RF_MAXL of them); the k rounds
// are block-wide arg-max reductions over cached list heads (only the winning list re-reads its head), the walk to tau and
// the append are parallel over the lists, the float64 re-score runs 16 candidates per pass (16 lanes each, all loads of a
// pass in flight), the ranking is one thread per candidate.
constexpr int RF_THREADS = 256;
constexpr int RF_MAXL = 4;               // lists per thread: n_parts <= 1024
constexpr int RF_STAGE_MAX = 6144;       // list entries per query that k_merge_refine<true> stages in LDS (48 KB + candidates)
constexpr int RF_STAGE_MAX_SMALLQ = 16384;   // ... for batches of <= 16 queries (128 KB: occupancy does not matter there)

constexpr size_t refine_lds_bytes(int cap) { return (size_t)cap * 20 + 256; }      // candidates + per-wave scratch of k_merge_refine

// STAGED (certified mode, n_parts * part_len <= RF_STAGE_MAX): all of the query's list entries are first copied to LDS in ONE
// round of independent loads, and the k selection rounds, the gathering of everything >= tau and the used-up test run on that
// copy.  The unstaged form walks the lists in global memory: every round's winner re-reads its list head and every list is
// walked entry by entry, i.e. 30-50 DEPENDENT trips to L2/HBM per query -- 0.17 ms for 1024 queries that re-score 150 rows
// each, where the re-scoring itself needs a tenth of that.
// The k-th largest of the NE scores staged in LDS (e_sc; -inf = "no entry", the smallest key), by a 4-pass radix select on the
// order-preserving image of the float bits.  Only the VALUE is found.  Called by all NT (>= 256 = the bins) threads of the
// block; hist [256] and xchg [2] are LDS scratch; e_sc needs no barrier of its own (the first pass's publishes it; so does the
// early exit).  Returns -inf when NE < k.
template <int NT = 256>
__device__ __forceinline__ float radix_select_kth(const float* e_sc, int NE, int k, int* hist, int* xchg) {
    static_assert(NT >= 256 && NT % 64 == 0, "one thread per bin at least");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto okey = [](float v) { const unsigned u = __float_as_uint(v); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    unsigned prefix = 0, pmask = 0;
    int krem = k;
    const bool have_k = NE >= k;                                         // (block-uniform)
    for (int shift = 24; shift >= 0 && have_k; shift -= 8) {
        if (NT == 256 || tid < 256) hist[tid] = 0;                       // 256 bins
        __syncthreads();                                                 // (first pass: also publishes e_sc)
        for (int i = tid; i < NE; i += NT) {
            const unsigned key = okey(e_sc[i]);
            if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        if (wave == 0) {                                                 // the digit that holds the krem-th largest matching key
            int c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = hist[4 * lane + j];
            const int s4 = c[0] + c[1] + c[2] + c[3];
            int incl = s4;                                               // matching keys in this lane's bins and all higher ones
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_down(incl, o, 64);
                if (lane + o < 64) incl += t;
            }
            int above = incl - s4;
            if (above < krem && krem <= incl) {                          // exactly one lane
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    if (above + c[j] >= krem) { xchg[0] = 4 * lane + j; xchg[1] = krem - above; break; }
                    above += c[j];
                }
            }
        }
        __syncthreads();
        prefix |= (unsigned)xchg[0] << shift;
        pmask |= 255u << shift;
        krem = xchg[1];
    }
    if (!have_k) { __syncthreads(); return -INFINITY; }                  // publishes e_sc all the same
    const unsigned u = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;
    return __uint_as_float(u);
}

// k_kth_floor: between the two phases of the certified tile scan (and, for a sharded search, after it).  One workgroup per query:
// a_k = the k-th best score among the candidates emitted so far (-inf when there are fewer than k, or the buffer overflowed).
//   floor_io  (optional) [nq]: raised to a_k - 2 eps -- a score k rows are KNOWN to reach, lowered by the scan's error on both
//             sides: every row of the exact top-k scores at least that, so the rest of the store is scanned with it;
//   lb_out    (optional) [nq][k]: the k best scan scores, each lowered by eps -- lower bounds of the exact scores of k DIFFERENT rows
//             of this store (unordered; -inf where there are fewer).  Shards exchange them: the k-th largest of all shards' values
//             is a lower bound of the exact k-th best score of the whole store.
struct KthParams {
    const float* score;        // [nq][cap]
    const int* cnt;            // [nq]
    int cap, k;
    const float* eps;          // [nq]
    float* floor_io;
    float* lb_out;
    float* ak_out;             // optional [nq]: a_k itself (the re-rank of a two-half search takes it instead of selecting again)
};
__global__ __launch_bounds__(RF_THREADS) void k_kth_floor(KthParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_k[];
    float* e_sc = reinterpret_cast<float*>(smem_k);                      // [cap]
    int* hist = reinterpret_cast<int*>(e_sc + p.cap);                    // [256]
    int* xchg = hist + 256;
    const int64_t q = blockIdx.x;
    const int c = p.cnt[q];
    const int NE = c <= p.cap ? c : 0;                                   // an overflowed buffer proves nothing (its query is rejected)
    for (int i = threadIdx.x; i < NE; i += RF_THREADS) e_sc[i] = p.score[q * p.cap + i];
    const float a_k = radix_select_kth(e_sc, NE, p.k, hist, xchg);
    const float e = p.eps[q];
    if (threadIdx.x == 0 && p.ak_out) p.ak_out[q] = a_k;
    if (threadIdx.x == 0 && p.floor_io && a_k > -INFINITY) p.floor_io[q] = fmaxf(p.floor_io[q], a_k - 2.f * e);
    if (p.lb_out) {
        float* out = p.lb_out + q * p.k;
        if (!(a_k > -INFINITY)) {
            for (int j = threadIdx.x; j < p.k; j += RF_THREADS) out[j] = -INFINITY;
            return;                                                          // (block-uniform)
        }
        if (threadIdx.x == 0) xchg[0] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < NE; i += RF_THREADS)                   // fewer than k entries are above the k-th best
            if (e_sc[i] > a_k) { const int pos = atomicAdd(&xchg[0], 1); if (pos < p.k) out[pos] = e_sc[i] - e; }
        __syncthreads();
        for (int j = min(xchg[0], p.k) + threadIdx.x; j < p.k; j += RF_THREADS) out[j] = a_k - e;
    }
}

// the scan's first admission floor: the rank-th best of the sample pre-pass's entries (n_ent per query, unsorted; a sentinel
// index = no entry), lowered by 2 eps -- a_k over the WHOLE store is >= the rank-th best of any subset for rank >= k, so the
// floor is <= tau = a_k - 2 eps always.  -inf when the sample holds fewer entries.  One workgroup per query (k_thr_from_parts, its
// predecessor, walked the 512 list heads `rank` times with a wave: 19 us against 6).
// clear / clear_n (optional): the query's candidate-score buffer, set to -inf -- the scan that raises its floors inside the launch reads
// other workgroups' candidates while they are being written: a slot reserved but not written yet must read as "no entry".
__global__ __launch_bounds__(RF_THREADS) void k_floor_from_sample(const float* __restrict__ score, const int* __restrict__ idx, int n_ent, int rank,
                                                                 const float* __restrict__ eps, float* __restrict__ thr,
                                                                 float* __restrict__ clear, int clear_n, int* __restrict__ zero, int zero_n) {
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    float* e_sc = reinterpret_cast<float*>(smem_f);                      // [n_ent]
    int* hist = reinterpret_cast<int*>(e_sc + n_ent);                    // [256]
    int* xchg = hist + 256;
    const int64_t q = blockIdx.x;
    for (int i = threadIdx.x; i < n_ent; i += RF_THREADS)
        e_sc[i] = idx[q * n_ent + i] != IDX_SENTINEL ? score[q * n_ent + i] : -INFINITY;
    if (clear) for (int i = threadIdx.x; i < clear_n; i += RF_THREADS) clear[q * clear_n + i] = -INFINITY;
    if (zero && q == 0) for (int i = threadIdx.x; i < zero_n; i += RF_THREADS) zero[i] = 0;      // (the scan's per-query-tile progress counters)
    const float a = radix_select_kth(e_sc, n_ent, rank, hist, xchg);
    if (threadIdx.x == 0) thr[q] = a > -INFINITY ? a - 2.f * eps[q] : -INFINITY;
}

// (A one-wave-per-query form -- the <= 1024 sample entries in 16 registers per lane, the rank by a 32-step binary search on the keys with
// ballot counts, no LDS -- was built in round 5 and measured at 15.0 us against this kernel's 11.6 for 1024 queries: one wave walks
// its dependent chain of loads and 32 search steps alone, where this one puts four waves on a query.  Removed.)

template <bool STAGED>
__global__ __launch_bounds__(RF_THREADS) void k_merge_refine(RefineParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_m[];
    double* c_key = reinterpret_cast<double*>(smem_m);                   // [cap] float64 distance of a candidate
    int64_t* c_gid = reinterpret_cast<int64_t*>(c_key + p.cap);          // [cap] its reported id (before id_base)
    int* c_id = reinterpret_cast<int*>(c_gid + p.cap);                   // [cap] its local row
    float* w_sc = reinterpret_cast<float*>(c_id + p.cap);                // [2][4] per-wave winners of a round (double-buffered)
    int* w_id = reinterpret_cast<int*>(w_sc + 8);                        // [2][4]
    int* w_pt = w_id + 8;                                                // [2][4]
    int* w_cnt = w_pt + 8;                                               // [4] per-wave counts of the append
    float* e_sc = reinterpret_cast<float*>(smem_m + refine_lds_bytes(p.cap));      // STAGED: [n_parts * part_len] scores ...
    int* e_id = reinterpret_cast<int*>(e_sc + (STAGED ? p.n_parts * p.part_len : 0));   // ... and rows; then the radix select's 256 bins
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = blockIdx.x;
    if (p.only_flagged && !p.only_flagged[q]) return;                     // (workgroup-uniform)
    const bool cert = p.eps != nullptr;
    const float two_eps = cert ? 2.f * p.eps[q] : 0.f;
    const int64_t qbase = q * p.n_parts;
    float tau = -INFINITY;
    int nsel = 0;
    int why = 0;                             // bit 0: candidate buffer full, 1: a full list used up, 2: admission floor above tau, 3: dropped
    const float floor_q = (cert && p.thr_init) ? p.thr_init[q] : -INFINITY;     // (read here: off the dependent chain below)
    const int dropped_q = (cert && p.qflag) ? p.qflag[q] : 0;

  if constexpr (STAGED) {
    // (emit-mode scan: ONE unsorted buffer of part_len entries per query, of which part_cnt[q] are filled -- more means overflow)
    const int filled = p.part_cnt ? p.part_cnt[q] : 0;
    const int NE = p.part_cnt ? min(filled, p.part_len) : p.n_parts * p.part_len;
    const int64_t ebase = qbase * p.part_len;
    int* hist = e_id + p.n_parts * p.part_len;                           // [256] digit histogram of the radix select
    for (int i = tid; i < NE; i += RF_THREADS) {
        const int id = p.idx[ebase + i];
        const float sc = p.score[ebase + i];
        e_id[i] = id;
        e_sc[i] = id != IDX_SENTINEL ? sc : -INFINITY;
    }
    // a_k = the k-th largest scan score (radix_select_kth: only the VALUE is needed -- every entry >= tau = a_k - 2 eps is re-scored,
    // the k best among them; the first version found the k best one by one, k rounds of a block-wide arg-max with two barriers
    // each: 40 % of this kernel)
    {
        float a_k;
        if (p.ak_in) { a_k = p.ak_in[q]; __syncthreads(); }             // (the barrier publishes e_sc, as the select's would)
        else a_k = radix_select_kth(e_sc, NE, p.k, hist, w_pt);
        if (a_k > -INFINITY) tau = a_k - two_eps;                        // (fewer than k listed rows: tau stays -inf, all of them go on)
    }
    // (Round 5, measured and reverted: halving the band.  tau = a_k - 2 eps pays eps twice; scoring the <= 32 rows with a >= a_k in fp32
    // first gives L = the k-th largest of their s - e, a lower bound of the exact k-th best, and only rows with a + eps >= L can be
    // among the exact k best.  On the benchmark's store that cut the re-scored rows from 151 to 105 per query -- and the kernel went
    // from 73 to 79 us: one more dependent stage per workgroup costs more than the 30 % of row reads it saves; this kernel is bound
    // by its chain of dependent round trips, not by bytes.)
    // sharded search: no row whose score is below (the best lower bound any shard has of the exact k-th best) - eps can be in the
    // GLOBAL top-k, whatever this shard's own k-th best is -- the re-rank takes only what can still matter
    if (cert && p.global_lb) tau = fmaxf(tau, p.global_lb[q] - 0.5f * two_eps);
    // everything >= tau goes on to the re-score; the used-up test: a FULL list whose last (= smallest) entry is >= tau
    int cnt = 0, used_up = 0;
    for (int i = tid; i < NE; i += RF_THREADS) cnt += (e_id[i] != IDX_SENTINEL && e_sc[i] >= tau) ? 1 : 0;
    if (p.part_cnt) used_up = filled > p.part_len ? 1 : 0;             // the scan emitted more than the buffer holds
    else
        for (int l = tid; l < p.n_parts; l += RF_THREADS) {
            const int last = l * p.part_len + p.part_len - 1;
            if (e_id[last] != IDX_SENTINEL && e_sc[last] >= tau) used_up = 1;
        }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) w_cnt[wave] = incl;
    used_up = __syncthreads_or(used_up);
    int base_slot = incl - cnt, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) base_slot += w_cnt[w]; total += w_cnt[w]; }
    if (cnt > 0)
        for (int i = tid; i < NE; i += RF_THREADS) {
            const int id = e_id[i];
            if (id != IDX_SENTINEL && e_sc[i] >= tau) {
                if (base_slot < p.cap) c_id[base_slot] = id;
                ++base_slot;
            }
        }
    nsel = total;
    if (nsel > p.cap) { nsel = p.cap; why |= 1; }
    if (used_up) why |= 2;
  } else {

    // cached heads of this thread's lists
    int pos[RF_MAXL];
    float hsc[RF_MAXL];
    int hid[RF_MAXL];
    auto fetch = [&](int part, int at, float& sc, int& id) {
        id = IDX_SENTINEL; sc = -INFINITY;
        if (part < p.n_parts && at < p.part_len) {
            const int64_t off = (qbase + part) * p.part_len + at;
            id = p.idx[off];
            if (id != IDX_SENTINEL) sc = p.score[off];
        }
    };
#pragma unroll
    for (int i = 0; i < RF_MAXL; ++i) { pos[i] = 0; fetch(tid + RF_THREADS * i, 0, hsc[i], hid[i]); }

    // 1) the k best by scan score (legacy mode: `cap` of them, and that is all): fixes a_k and tau = a_k - 2 eps
    const int rounds = cert ? min(p.k, p.cap) : p.cap;
    for (int o = 0; o < rounds; ++o) {
        float bs = -INFINITY;
        int bi = IDX_SENTINEL, bp = -1;
#pragma unroll
        for (int i = 0; i < RF_MAXL; ++i)
            if (hid[i] != IDX_SENTINEL && (bp < 0 || hsc[i] > bs || (hsc[i] == bs && hid[i] < bi))) { bs = hsc[i]; bi = hid[i]; bp = tid + RF_THREADS * i; }
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) {
            const float os = __shfl_xor(bs, ofs, 64);
            const int oi = __shfl_xor(bi, ofs, 64);
            const int op = __shfl_xor(bp, ofs, 64);
            if (op >= 0 && (bp < 0 || os > bs || (os == bs && oi < bi))) { bs = os; bi = oi; bp = op; }
        }
        const int slot = (o & 1) * 4;
        if (lane == 0) { w_sc[slot + wave] = bs; w_id[slot + wave] = bi; w_pt[slot + wave] = bp; }
        __syncthreads();
        bs = w_sc[slot]; bi = w_id[slot]; bp = w_pt[slot];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float os = w_sc[slot + w];
            const int oi = w_id[slot + w], op = w_pt[slot + w];
            if (op >= 0 && (bp < 0 || os > bs || (os == bs && oi < bi))) { bs = os; bi = oi; bp = op; }
        }
        if (bp < 0) break;                                   // every list exhausted (block-uniform)
        if ((bp & (RF_THREADS - 1)) == tid) {
#pragma unroll
            for (int i = 0; i < RF_MAXL; ++i)
                if (bp == tid + RF_THREADS * i) { pos[i] += 1; fetch(bp, pos[i], hsc[i], hid[i]); }
        }
        if (tid == 0) c_id[nsel] = bi;
        ++nsel;
        if (cert && nsel == p.k) tau = bs - two_eps;
    }
    if (cert && p.global_lb && nsel == p.k) tau = fmaxf(tau, p.global_lb[q] - 0.5f * two_eps);     // (sharded search: see the staged form)
    if (cert) {
        int cnt = 0, used_up = 0;
        int endp[RF_MAXL];
#pragma unroll
        for (int i = 0; i < RF_MAXL; ++i) {
            const int part = tid + RF_THREADS * i;
            int e = pos[i];
            if (nsel == p.k && part < p.n_parts) {           // walk on while the entries are >= tau (sorted lists)
                const int64_t lb = (qbase + part) * p.part_len;
                while (e < p.part_len) {
                    const int id = p.idx[lb + e];
                    if (id == IDX_SENTINEL || p.score[lb + e] < tau) break;
                    ++e;
                }
            }
            endp[i] = e;
            cnt += e - pos[i];
            // a full list whose entries are ALL >= tau may hide more such rows; everything else a list hides is below its
            // first entry < tau (or, for a list that never filled up, below the admission floor)
            if (part < p.n_parts && e >= p.part_len && p.idx[(qbase + part) * p.part_len + p.part_len - 1] != IDX_SENTINEL) used_up = 1;
        }
        // slots by a prefix sum over the block (lists in thread order)
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) w_cnt[wave] = incl;
        used_up = __syncthreads_or(used_up);
        int base_slot = nsel + incl - cnt, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wave) base_slot += w_cnt[w]; total += w_cnt[w]; }
#pragma unroll
        for (int i = 0; i < RF_MAXL; ++i) {
            const int part = tid + RF_THREADS * i;
            const int n_i = endp[i] - pos[i];
            if (n_i > 0) {
                const int64_t lb = (qbase + part) * p.part_len;
                for (int j = 0; j < n_i; ++j)
                    if (base_slot + j < p.cap) c_id[base_slot + j] = p.idx[lb + pos[i] + j];
                base_slot += n_i;
            }
        }
        nsel += total;
        if (nsel > p.cap) { nsel = p.cap; why |= 1; }
        if (used_up) why |= 2;
    }
  }
    if (cert) {
        // rows below the floor are unlisted: the floor must not exceed tau (tau = -inf when fewer than k rows are listed, unless a
        // global bound set it)
        if (floor_q > -INFINITY && !(floor_q <= tau)) why |= 4;
        if (dropped_q) why |= 8;
        if (tid == 0) {
            if (why) p.flag_sel[atomicAdd(p.flag_count, 1)] = (int)q;
            if (p.qflag_out) p.qflag_out[q] = why;
            if (p.stats && !RADAD_DBG(p.debug, 128)) {
                atomicAdd(&p.stats[0], nsel);
                for (int b2 = 0; b2 < 4; ++b2) if (why & (1 << b2)) atomicAdd(&p.stats[1 + b2], 1);
            }
        }
    }
    __syncthreads();                         // c_id complete
    // 1b) fp32 funnel (certified mode, more candidates than results): every candidate is scored ONCE in fp32 -- s = q.y, or
    //     (2q - y).y = |q|^2 - |q - y|^2 for L2 -- together with e = gamma sum |products|, a rigorous bound on that sum's
    //     rounding error (every term passes through at most dim/16 + 4 fused accumulations; gamma has slack for the
    //     accumulation of e itself and for forming s -+ e).  A candidate with at least k others CERTAINLY above it
    //     (s_j - e_j > s_c + e_c) cannot be among the exact k best and is dropped; only the survivors -- k plus the few within
    //     ~1e-6 of the k-th -- go through the float64 re-score and the ranking, whose cost is linear resp. quadratic in the
    //     count (with 150 candidates per query they were two thirds of this kernel).
    //     (a handful beyond k -- the fp32 scans' candidates, the IVF index's coarse step -- go straight to float64: the funnel's pass over
    //     their rows costs what it saves)
    if (cert && nsel > p.k + 16 && !RADAD_DBG(p.debug, 1024)) {
        float2* f_lh = reinterpret_cast<float2*>(c_key);                 // (s - e, s + e) per candidate; c_key is free until 2)
        int* tmp_id = reinterpret_cast<int*>(c_gid);                     // survivors' rows; c_gid is free until 3)
        const float gamma = (float)(p.dim / 16 + 8) * 5.9604645e-08f * 1.01f;
        const float* qrow = p.q + q * p.dim;
        const int sub = tid >> 4, l16 = tid & 15;
        for (int c0 = 0; c0 < nsel; c0 += 32) {
            int cc[2];
            bool on[2];
            int64_t rid[2];
            float sdot[2] = {0.f, 0.f}, sabs[2] = {0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                cc[h] = c0 + 16 * h + sub;
                on[h] = cc[h] < nsel;
                rid[h] = on[h] ? (int64_t)c_id[cc[h]] : 0;
            }
            for (int i0 = 0; i0 < p.dim; i0 += 512) {
                f32x4 b[2][8];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + (u * 16 + l16) * 4;
                        b[h][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (on[h] && i < p.dim) {
                            if (p.db_f16) {
                                const f16x4 h4 = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(p.db) + rid[h] * p.dim + i);
                                b[h][u] = f32x4{(float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]};
                            } else {
                                b[h][u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.db) + rid[h] * p.dim + i);
                            }
                        }
                    }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + (u * 16 + l16) * 4;
                    if (i < p.dim) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(qrow + i);
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float y = b[h][u][e];
                                const float x = p.l2 ? 2.f * a[e] - y : a[e];      // (one more rounding per term for L2: inside gamma's slack)
                                sdot[h] = fmaf(x, y, sdot[h]);
                                sabs[h] = fmaf(fabsf(x), fabsf(y), sabs[h]);
                            }
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int ofs = 8; ofs > 0; ofs >>= 1) {
                    sdot[h] += __shfl_xor(sdot[h], ofs, 64);
                    sabs[h] += __shfl_xor(sabs[h], ofs, 64);
                }
                if (on[h] && l16 == 0) {
                    const float e = gamma * sabs[h] + 4.8e-7f * fabsf(sdot[h]);      // + 4 ulp for forming s -+ e
                    f_lh[cc[h]] = make_float2(sdot[h] - e, sdot[h] + e);
                }
            }
        }
        __syncthreads();
        // survivors: fewer than k candidates certainly above (T threads per candidate share the count)
        const int T = nsel <= 64 ? 4 : (nsel <= 128 ? 2 : 1);
        const int part = tid & (T - 1);
        int n_keep = 0;                                                  // survivors so far (block-uniform)
        for (int c0 = 0; c0 < nsel; c0 += RF_THREADS / T) {
            const int c = c0 + tid / T;
            const bool onc = c < nsel;
            const float hi_c = onc ? f_lh[c].y : 0.f;
            int above = 0;
            for (int j = part; j < nsel; j += T) above += (int)(f_lh[j].x > hi_c);
            if (T >= 2) above += __shfl_xor(above, 1, 64);
            if (T == 4) above += __shfl_xor(above, 2, 64);
            const bool keep = onc && part == 0 && !(above >= p.k);       // (NaN scores compare false everywhere: kept)
            const unsigned long long m = __ballot(keep);
            if (lane == 0) w_cnt[wave] = __popcll(m);
            __syncthreads();
            int base = n_keep, total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { if (w < wave) base += w_cnt[w]; total += w_cnt[w]; }
            if (keep) tmp_id[base + __popcll(m & ((1ull << lane) - 1ull))] = c_id[c];
            n_keep += total;
            __syncthreads();
        }
        for (int c = tid; c < n_keep; c += RF_THREADS) c_id[c] = tmp_id[c];
        nsel = n_keep;
        __syncthreads();
    }
    // 2) float64 re-score: 16 lanes per candidate, 32 candidates per pass (two per 16-lane group), all row loads of a pass
    //    issued before the first multiply (the stored rows are random 2 KB reads: latency-bound unless many are in flight)
    {
        const float* qrow = p.q + q * p.dim;
        const int sub = tid >> 4, l16 = tid & 15;
        for (int c0 = 0; c0 < (RADAD_DBG(p.debug, 256) ? 0 : nsel); c0 += 32) {
            int cc[2];
            bool on[2];
            int64_t rid[2];
            double acc[2] = {0.0, 0.0};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                cc[h] = c0 + 16 * h + sub;
                on[h] = cc[h] < nsel;
                rid[h] = on[h] ? (int64_t)c_id[cc[h]] : 0;
            }
            for (int i0 = 0; i0 < p.dim; i0 += 512) {                 // 16 lanes x 8 x 4 elements per sweep
                f32x4 b[2][8];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + (u * 16 + l16) * 4;
                        b[h][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (on[h] && i < p.dim) {
                            if (p.db_f16) {
                                const f16x4 h4 = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(p.db) + rid[h] * p.dim + i);
                                b[h][u] = f32x4{(float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]};
                            } else {
                                b[h][u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.db) + rid[h] * p.dim + i);
                            }
                        }
                    }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + (u * 16 + l16) * 4;
                    if (i < p.dim) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(qrow + i);
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (p.l2) { const double d = (double)a[e] - (double)b[h][u][e]; acc[h] += d * d; }
                                else acc[h] += (double)a[e] * (double)b[h][u][e];
                            }
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int ofs = 8; ofs > 0; ofs >>= 1) acc[h] += __shfl_xor(acc[h], ofs, 64);
                if (on[h] && l16 == 0) c_key[cc[h]] = acc[h];
            }
        }
    }
    __syncthreads();
    // 3) rank by (distance, id): L2 ascending, IP descending; ids are unique so ranks are a permutation.  Rank = number of
    //    candidates that come first, counted branch-free over "larger is better" keys by 1, 2 or 4 threads per candidate
    //    (the first version -- one thread per candidate, short-circuit compares, the id map looked up inside the loop --
    //    was a third of this kernel's time)
    const int nrank = RADAD_DBG(p.debug, 512) ? 0 : nsel;
    for (int c = tid; c < nrank; c += RF_THREADS) {
        c_gid[c] = p.id_map ? p.id_map[c_id[c]] : (int64_t)c_id[c];
        double kc = p.l2 ? -c_key[c] : c_key[c];
        // a NaN key (non-finite query or row) compares false with everything: several candidates would take the same rank and
        // leave output slots unwritten.  It ranks last instead, among its kind by id: the order stays total.
        if (kc != kc) kc = -(double)INFINITY;
        c_key[c] = kc;
    }
    __syncthreads();
    {
        const int T = nrank <= 64 ? 4 : (nrank <= 128 ? 2 : 1);
        const int part = tid & (T - 1);
        for (int c0 = 0; c0 < nrank; c0 += RF_THREADS / T) {
            const int c = c0 + tid / T;
            const bool onc = c < nrank;
            const double kc = onc ? c_key[c] : 0.0;
            const int64_t ic = onc ? c_gid[c] : 0;
            int rank = 0;
            for (int j = part; j < nrank; j += T) {
                const double kj = c_key[j];
                const int64_t ij = c_gid[j];
                rank += (int)((kj > kc) | ((kj == kc) & (ij < ic)));
            }
            if (T >= 2) rank += __shfl_xor(rank, 1, 64);
            if (T == 4) rank += __shfl_xor(rank, 2, 64);
            if (onc && part == 0 && rank < p.k) {
                const double kd = p.l2 ? -kc : kc;
                p.out_dist[q * p.k + rank] = (float)kd;
                p.out_idx[q * p.k + rank] = ic + p.id_base;
                if (p.out_key) p.out_key[q * p.k + rank] = kd;
            }
        }
    }
    for (int o = nsel + tid; o < p.k; o += RF_THREADS) {   // faiss fills what it cannot find with -1 / +-inf
        p.out_dist[q * p.k + o] = p.l2 ? INFINITY : -INFINITY;
        p.out_idx[q * p.k + o] = -1;
        if (p.out_key) p.out_key[q * p.k + o] = p.l2 ? (double)INFINITY : -(double)INFINITY;
    }
}

// ---- the re-rank of a SMALL batch (<= 16 queries: the online predict() search, pipeline.py:1038-1054) -------------------------
// k_merge_refine is one 256-thread workgroup per query: with one query that is ONE workgroup walking 500-800 partial lists (the
// streaming kernels write one per workgroup) and re-scoring its candidates 16 lanes at a time -- 0.123 ms of the reference's own
// search (25 423 x 5376, k 15), twice its scan; 45 us on the 1 M x 512 store.  Where the time went: the radix select's LDS-atomic
// histogram over all 16 K staged entries (~30 us), then 11 dependent sweeps of row loads per re-score pass, twice (fp32 funnel,
// float64).  This form:
//   * RS_SPLIT workgroups of 1024 threads per query, thread t <-> partial list t (<= RS_MAX_PARTS lists, sorted best-first);
//   * tau from the list HEADS only: h_k = the k-th largest head is a lower bound of a_k (k different lists reach it), so
//     tau = h_k - 2 eps admits every row the certificate needs (knn.hip: k_merge_refine's header) -- and, the store being split
//     over hundreds of lists, hardly any more (h_k = a_k unless two of the k best rows share a list);
//   * the lists are walked while their entries are >= tau (sorted: usually 0 or 1 step), slots by a block prefix sum -- every
//     workgroup of the query computes the same candidate list;
//   * float64 re-score of ALL candidates (no fp32 funnel: there are few), one WAVE per candidate with the row's loads of a sweep
//     (4096 elements) in flight together, the candidates dealt round-robin to the RS_SPLIT x 16 waves of the query;
//   * the keys meet in global memory; the workgroup that arrives last (device-scope counter, as k_exact_scan's slices) ranks
//     them, writes the result and the certificate's verdict.
constexpr int RS_THREADS = 1024;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_SPLIT = 8;
constexpr int RS_MAX_PARTS = RS_THREADS;
constexpr size_t refine_small_lds_bytes(int cap) { return (size_t)cap * 20 + (size_t)RS_MAX_PARTS * 4 + 256 * 4 + 256; }

__global__ __launch_bounds__(RS_THREADS) void k_refine_small(RefineParams p, double* __restrict__ wkey, int* __restrict__ wcount) {
    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    double* c_key = reinterpret_cast<double*>(smem_s);                   // [cap]
    int64_t* c_gid = reinterpret_cast<int64_t*>(c_key + p.cap);          // [cap]
    int* c_id = reinterpret_cast<int*>(c_gid + p.cap);                   // [cap]
    float* heads = reinterpret_cast<float*>(c_id + p.cap);               // [RS_MAX_PARTS]
    int* hist = reinterpret_cast<int*>(heads + RS_MAX_PARTS);            // [256]
    int* xch = hist + 256;                                               // [2] select, [16] wave counts, [1] arrival
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x;
    const int64_t q = blockIdx.y;
    const float two_eps = 2.f * p.eps[q];
    const float floor_q = p.thr_init ? p.thr_init[q] : -INFINITY;
    const int dropped_q = p.qflag ? p.qflag[q] : 0;
    const int64_t lb = (q * p.n_parts + tid) * p.part_len;               // this thread's list
    const bool have = tid < p.n_parts;

    int hid = IDX_SENTINEL;
    float hs = -INFINITY;
    if (have) { hid = p.idx[lb]; const float s0 = p.score[lb]; if (hid != IDX_SENTINEL) hs = s0; }
    heads[tid] = hs;
    const float h_k = radix_select_kth<RS_THREADS>(heads, p.n_parts, p.k, hist, xch);
    float tau = h_k > -INFINITY ? h_k - two_eps : -INFINITY;             // (fewer than k lists: everything listed goes on)
    if (p.global_lb) tau = fmaxf(tau, p.global_lb[q] - 0.5f * two_eps);  // (sharded search: see k_merge_refine)
    int cnt = 0, used_up = 0;
    if (have && hid != IDX_SENTINEL && hs >= tau) {
        cnt = 1;
        while (cnt < p.part_len) {
            const int id = p.idx[lb + cnt];
            if (id == IDX_SENTINEL || !(p.score[lb + cnt] >= tau)) break;
            ++cnt;
        }
        if (cnt == p.part_len) used_up = 1;                              // a FULL list of entries >= tau may hide more such rows
    }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) xch[2 + wave] = incl;
    used_up = __syncthreads_or(used_up);
    int base_slot = incl - cnt, total = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) { const int c = xch[2 + w]; if (w < wave) base_slot += c; total += c; }
    for (int j = 0; j < cnt; ++j)
        if (base_slot + j < p.cap) c_id[base_slot + j] = j == 0 ? hid : p.idx[lb + j];
    int nsel = total, why = 0;
    if (nsel > p.cap) { nsel = p.cap; why |= 1; }
    if (used_up) why |= 2;
    if (floor_q > -INFINITY && !(floor_q <= tau)) why |= 4;
    if (dropped_q) why |= 8;
    __syncthreads();                                                     // c_id complete

    // float64 re-score: wave (r, wave) takes candidates r * 16 + wave, + RS_SPLIT * 16, ...
    {
        const float* qrow = p.q + q * p.dim;
        for (int c = r * RS_WAVES + wave; c < nsel; c += RS_SPLIT * RS_WAVES) {
            const int64_t rid = (int64_t)c_id[c];
            double acc = 0.0;
            for (int i0 = 0; i0 < p.dim; i0 += 4096) {                   // 64 lanes x 16 x 4 elements per sweep
                f32x4 b[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + (u * 64 + lane) * 4;
                    b[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (i < p.dim) {
                        if (p.db_f16) {
                            const f16x4 h4 = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(p.db) + rid * p.dim + i);
                            b[u] = f32x4{(float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]};
                        } else {
                            b[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.db) + rid * p.dim + i);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + (u * 64 + lane) * 4;
                    if (i < p.dim) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(qrow + i);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (p.l2) { const double d = (double)a[e] - (double)b[u][e]; acc += d * d; }
                            else acc += (double)a[e] * (double)b[u][e];
                        }
                    }
                }
            }
#pragma unroll
            for (int ofs = 32; ofs > 0; ofs >>= 1) acc += __shfl_xor(acc, ofs, 64);
            if (lane == 0) wkey[q * p.cap + c] = acc;
        }
    }
    __threadfence();                                     // release: this workgroup's keys are visible device-wide before the counter moves
    __syncthreads();
    if (tid == 0) {
        const int old = atomicAdd(&wcount[q], 1);
        xch[18] = old == RS_SPLIT - 1;
        if (old == RS_SPLIT - 1) wcount[q] = 0;          // ready for the next search
    }
    __syncthreads();
    if (!xch[18]) return;                                // (workgroup-uniform)
    __threadfence();                                     // acquire
    if (tid == 0) {
        if (why) p.flag_sel[atomicAdd(p.flag_count, 1)] = (int)q;
        if (p.stats) {
            atomicAdd(&p.stats[0], nsel);
            for (int b2 = 0; b2 < 4; ++b2) if (why & (1 << b2)) atomicAdd(&p.stats[1 + b2], 1);
        }
    }
    // rank by (distance, id), as k_merge_refine's step 3
    for (int c = tid; c < nsel; c += RS_THREADS) {
        c_gid[c] = p.id_map ? p.id_map[c_id[c]] : (int64_t)c_id[c];
        double kc = __builtin_nontemporal_load(&wkey[q * p.cap + c]);
        if (p.l2) kc = -kc;
        if (kc != kc) kc = -(double)INFINITY;            // a NaN key ranks last, among its kind by id
        c_key[c] = kc;
    }
    __syncthreads();
    {
        const int T = nsel <= 256 ? 4 : 2;               // (nsel <= cap <= 512)
        const int part = tid & (T - 1);
        for (int c0 = 0; c0 < nsel; c0 += RS_THREADS / T) {
            const int c = c0 + tid / T;
            const bool onc = c < nsel;
            const double kc = onc ? c_key[c] : 0.0;
            const int64_t ic = onc ? c_gid[c] : 0;
            int rank = 0;
            for (int j = part; j < nsel; j += T) {
                const double kj = c_key[j];
                const int64_t ij = c_gid[j];
                rank += (int)((kj > kc) | ((kj == kc) & (ij < ic)));
            }
            rank += __shfl_xor(rank, 1, 64);
            if (T == 4) rank += __shfl_xor(rank, 2, 64);
            if (onc && part == 0 && rank < p.k) {
                const double kd = p.l2 ? -kc : kc;
                p.out_dist[q * p.k + rank] = (float)kd;
                p.out_idx[q * p.k + rank] = ic + p.id_base;
                if (p.out_key) p.out_key[q * p.k + rank] = kd;
            }
        }
    }
    for (int o = nsel + tid; o < p.k; o += RS_THREADS) {   // faiss fills what it cannot find with -1 / +-inf
        p.out_dist[q * p.k + o] = p.l2 ? INFINITY : -INFINITY;
        p.out_idx[q * p.k + o] = -1;
        if (p.out_key) p.out_key[q * p.k + o] = p.l2 ? (double)INFINITY : -(double)INFINITY;
    }
}

// ---- exact float64 search of the queries the certificate rejected --------------------------------------------------
// Driven entirely from the device: the number of queries (*count) and their indices (sel) were written by
// k_merge_refine; the launch geometry is fixed, workgroups with nothing to do leave at once.  Slice s of the store
// (n_slices row ranges) is scanned by workgroups (s, y); workgroup (s, y) takes the query groups y, y + gridDim.y, ...
// of `group` (<= 8) queries each.  A wave reads one row per step (coalesced), every lane multiplies its elements with
// the group's queries from LDS in float64, a butterfly sum leaves the (row, query) scores in all lanes; each wave keeps
// its exact top-k per query as a sorted LDS list (wave-cooperative insertion), the workgroup merges its waves' lists
// and writes one list per (query slot, slice); the slice that arrives last for a query group merges the slices (exact_merge_slot)
// and overwrites the queries' results.
// Keys are "larger is better" doubles (inner product, or minus the squared distance); order (key desc, id asc).
constexpr int KX_THREADS = 512;
constexpr int KX_WAVES = KX_THREADS / 64;
constexpr int KX_SLICES = 64;
constexpr int KX_GROUPS_Y = 4;

struct ExactParams {
    const void* db; int db_f16;
    const float* q;            // [nq][dim] fp32 queries (normalised for cosine)
    const int* sel;            // [*count] query indices
    const int* count;
    int64_t n; int dim, k, l2, group;
    int64_t slice_rows;
    double* pkey;              // [nq][KX_SLICES][k]
    int* pidx;
    int64_t id_base;
    float* out_dist; int64_t* out_idx; double* out_key;
    int* host_stats;           // pinned host memory (device-visible): the kernel leaves the search's 6 counters and its batch size ([7]) there, then ...
    int stamp;                 // ... this value in host_stats[6] (the search's sequence number + 1): the host reads a report only when its stamp is there
    int nq_report;             // the search's batch size (travels with the report: the host may be many searches ahead when it reads it)
    int* arrive;               // [query groups] arrival counters of the slices (zero between launches: the last arrival resets its own)
};

__device__ __forceinline__ bool kx_better(double ka, int ia, double kb, int ib) { return ka > kb || (ka == kb && ia < ib); }
__device__ __forceinline__ void exact_merge_slot(const ExactParams& p, int slot, int lane);

__global__ __launch_bounds__(KX_THREADS) void k_exact_scan(ExactParams p) {
    const int count = *p.count;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && p.host_stats) {
        *reinterpret_cast<volatile int*>(&p.host_stats[6]) = 0;          // the slot is being rewritten: no report until the new stamp is there
        __threadfence_system();
        for (int i = 0; i < 6; ++i) p.host_stats[i] = p.count[i];       // count + 5 statistics
        p.host_stats[7] = p.nq_report;
        __threadfence_system();
        *reinterpret_cast<volatile int*>(&p.host_stats[6]) = p.stamp;
    }
    if (count <= 0) return;
    __shared__ int s_last;
    extern __shared__ __attribute__((aligned(16))) char smem_x[];
    const int G = p.group;
    float* sQ = reinterpret_cast<float*>(smem_x);                                   // [G][dim]
    double* sKey = reinterpret_cast<double*>(smem_x + (size_t)G * p.dim * 4);       // [KX_WAVES][G][k]
    int* sId = reinterpret_cast<int*>(sKey + (size_t)KX_WAVES * G * p.k);           // [KX_WAVES][G][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r_begin = (int64_t)blockIdx.x * p.slice_rows;
    const int64_t r_end = min(r_begin + p.slice_rows, p.n);
    const int n_groups = (count + G - 1) / G;
    const int nv = p.dim >> 2;
    for (int g = blockIdx.y; g < n_groups; g += gridDim.y) {
        const int ng = min(G, count - g * G);
        __syncthreads();
        for (int i = tid; i < ng * nv; i += KX_THREADS) {
            const int qq = i / nv, c4 = i % nv;
            reinterpret_cast<f32x4*>(sQ + (size_t)qq * p.dim)[c4] =
                reinterpret_cast<const f32x4*>(p.q + (int64_t)p.sel[g * G + qq] * p.dim)[c4];
        }
        double* wKey = sKey + (size_t)wave * G * p.k;
        int* wId = sId + (size_t)wave * G * p.k;
        for (int i = lane; i < G * p.k; i += 64) { wKey[i] = -INFINITY; wId[i] = IDX_SENTINEL; }
        __syncthreads();
        for (int64_t row = r_begin + wave; row < r_end; row += KX_WAVES) {
            double part[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) part[j] = 0.0;
            for (int i = lane; i < nv; i += 64) {
                f32x4 y;
                if (p.db_f16) {
                    const f16x4 h4 = reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(p.db) + row * p.dim)[i];
                    y = f32x4{(float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]};
                } else {
                    y = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.db) + row * p.dim)[i];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < ng) {
                        const f32x4 x = reinterpret_cast<const f32x4*>(sQ + (size_t)j * p.dim)[i];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (p.l2) { const double d = (double)x[e] - (double)y[e]; part[j] -= d * d; }
                            else part[j] += (double)x[e] * (double)y[e];
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < ng) {
                    double v = part[j];
#pragma unroll
                    for (int ofs = 32; ofs > 0; ofs >>= 1) v += __shfl_xor(v, ofs, 64);
                    // wave-uniform from here: insert (v, row) into the sorted list of query j when it beats the k-th entry
                    double* lk = wKey + (size_t)j * p.k;
                    int* li = wId + (size_t)j * p.k;
                    if (kx_better(v, (int)row, lk[p.k - 1], li[p.k - 1])) {
                        // lane l looks after entries l, l + 64 (k <= 128): read all, count the better ones, shift, write
                        double e0 = -INFINITY, e1 = -INFINITY;
                        int i0 = IDX_SENTINEL, i1 = IDX_SENTINEL;
                        if (lane < p.k) { e0 = lk[lane]; i0 = li[lane]; }
                        if (lane + 64 < p.k) { e1 = lk[lane + 64]; i1 = li[lane + 64]; }
                        const bool b0 = lane < p.k && kx_better(e0, i0, v, (int)row);
                        const bool b1 = lane + 64 < p.k && kx_better(e1, i1, v, (int)row);
                        const int pos = __popcll(__ballot(b0)) + __popcll(__ballot(b1));
                        if (lane < p.k && !b0 && lane + 1 < p.k) { lk[lane + 1] = e0; li[lane + 1] = i0; }
                        if (lane + 64 < p.k && !b1 && lane + 65 < p.k) { lk[lane + 65] = e1; li[lane + 65] = i1; }
                        if (lane == (pos & 63)) { lk[pos] = v; li[pos] = (int)row; }
                    }
                }
            }
        }
        __syncthreads();
        // wave j merges the KX_WAVES lists of query j (lane 0, serial: k * 8 comparisons) into the slice's partial list
        if (wave < ng && lane == 0) {
            int pos[KX_WAVES];
#pragma unroll
            for (int w = 0; w < KX_WAVES; ++w) pos[w] = 0;
            const int64_t ob = ((int64_t)(g * G + wave) * KX_SLICES + blockIdx.x) * p.k;
            for (int o = 0; o < p.k; ++o) {
                double bk = -INFINITY; int bi = IDX_SENTINEL, bw = -1;
#pragma unroll
                for (int w = 0; w < KX_WAVES; ++w) {
                    if (pos[w] < p.k) {
                        const double kk = sKey[((size_t)w * G + wave) * p.k + pos[w]];
                        const int ii = sId[((size_t)w * G + wave) * p.k + pos[w]];
                        if (ii != IDX_SENTINEL && (bw < 0 || kx_better(kk, ii, bk, bi))) { bk = kk; bi = ii; bw = w; }
                    }
                }
                p.pkey[ob + o] = bw < 0 ? -INFINITY : bk;
                p.pidx[ob + o] = bw < 0 ? IDX_SENTINEL : bi;
                if (bw >= 0) {
#pragma unroll
                    for (int w = 0; w < KX_WAVES; ++w) if (w == bw) ++pos[w];
                }
            }
        }
        // the slice that arrives LAST for this query group merges the group's KX_SLICES partial lists (it used to be a second launch --
        // and the usual search, with nothing rejected, paid two empty launches and the gap between them)
        __threadfence();                                 // release: this slice's lists are visible device-wide before the counter moves
        __syncthreads();
        if (tid == 0) {
            const int old = atomicAdd(&p.arrive[g], 1);
            s_last = old == KX_SLICES - 1;
            if (s_last) p.arrive[g] = 0;                 // ready for the next launch
        }
        __syncthreads();
        if (s_last) {                                    // (workgroup-uniform)
            __threadfence();                             // acquire
            if (wave < ng) exact_merge_slot(p, g * G + wave, lane);
        }
    }
}

// one wave: merge the KX_SLICES partial lists of rejected-query slot `slot`, overwrite the query's output rows
__device__ __forceinline__ void exact_merge_slot(const ExactParams& p, int slot, int lane) {
    static_assert(KX_SLICES == 64, "one list per lane");
    {
    const int64_t q = p.sel[slot];
    const double* lk = p.pkey + ((int64_t)slot * KX_SLICES + lane) * p.k;
    const int* li = p.pidx + ((int64_t)slot * KX_SLICES + lane) * p.k;
    int pos = 0;
    for (int o = 0; o < p.k; ++o) {
        double hk = -INFINITY; int hi = IDX_SENTINEL;
        if (pos < p.k) { hk = lk[pos]; hi = li[pos]; }
        double bk = hk; int bi = hi;
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) {
            const double ok = __shfl_xor(bk, ofs, 64);
            const int oi = __shfl_xor(bi, ofs, 64);
            if (oi != IDX_SENTINEL && (bi == IDX_SENTINEL || kx_better(ok, oi, bk, bi))) { bk = ok; bi = oi; }
        }
        if (bi != IDX_SENTINEL && hi == bi) ++pos;          // ids are unique: exactly one lane advances
        if (lane == 0) {
            const bool none = bi == IDX_SENTINEL;
            const double d = none ? (p.l2 ? (double)INFINITY : -(double)INFINITY) : (p.l2 ? -bk : bk);
            p.out_dist[q * p.k + o] = (float)d;
            p.out_idx[q * p.k + o] = none ? -1 : (int64_t)bi + p.id_base;
            if (p.out_key) p.out_key[q * p.k + o] = d;
        }
    }
    }
}

template <typename KeyT>
struct ListMergeParams {
    const KeyT* key;          // [n_parts, nq, k] final distances of each shard
    const int64_t* idx;       // [n_parts, nq, k] global ids, -1 = unfilled
    int n_parts, k, l2;
    int64_t nq;
    float* out_dist;
    int64_t* out_idx;
    double* out_key;          // optional (KeyT == double)
};

template <typename KeyT>
__global__ __launch_bounds__(256) void k_merge_lists(ListMergeParams<KeyT> p) {
    extern __shared__ int s_pos_all[];   // [4 waves][n_parts]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= p.nq) return;
    int* s_pos = s_pos_all + wave * p.n_parts;
    for (int i = lane; i < p.n_parts; i += 64) s_pos[i] = 0;
    for (int o = 0; o < p.k; ++o) {
        KeyT bs = 0;             // "larger is better" key
        int64_t bi = INT64_MAX;
        int bp = -1;
        for (int part = lane; part < p.n_parts; part += 64) {
            const int pos = s_pos[part];
            if (pos >= p.k) continue;
            const int64_t off = ((int64_t)part * p.nq + q) * p.k + pos;
            const int64_t id = p.idx[off];
            if (id < 0) continue;
            const KeyT key = p.l2 ? -p.key[off] : p.key[off];
            if (bp < 0 || key > bs || (key == bs && id < bi)) { bs = key; bi = id; bp = part; }
        }
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) {
            const KeyT os = __shfl_xor(bs, ofs, 64);
            const int64_t oi = __shfl_xor(bi, ofs, 64);
            const int op = __shfl_xor(bp, ofs, 64);
            if (op >= 0 && (bp < 0 || os > bs || (os == bs && oi < bi))) { bs = os; bi = oi; bp = op; }
        }
        if (bp >= 0 && (bp & 63) == lane) s_pos[bp] += 1;
        if (lane == 0) {
            const KeyT d = bp < 0 ? (p.l2 ? (KeyT)INFINITY : -(KeyT)INFINITY) : (p.l2 ? -bs : bs);
            p.out_dist[q * p.k + o] = (float)d;
            p.out_idx[q * p.k + o] = bp < 0 ? -1 : bi;
            if (p.out_key) p.out_key[q * p.k + o] = (double)d;
        }
    }
}

// ---- append path ----------------------------------------------------------------------------------
// one wave per row.  mode 0: copy; 1: copy + ynorm = sum x^2; 2: x / (sqrt(sum x^2) + 1e-12)
// OutT = float: as above.  OutT = _Float16 (fp16 store / fp16 scan queries): values are rounded to nearest-even on the
// way out and |y|^2 is taken of the ROUNDED row, so the L2 score 2 q.y - |y|^2 is consistent with what is stored.
template <typename OutT>
__global__ __launch_bounds__(256) void k_rows_prepare(const float* in, OutT* out, float* ynorm, int64_t n, int dim, int mode) {
    constexpr bool H = sizeof(OutT) == 2;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(in + row * dim);
    const int nv = dim >> 2;
    auto put = [&](int i, f32x4 v) {
        if constexpr (H) {
            const f16x4 h4 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            reinterpret_cast<f16x4*>(out + row * dim)[i] = h4;
        } else {
            reinterpret_cast<f32x4*>(out + row * dim)[i] = v;
        }
    };
    float ss = 0.f;
    if (mode != 0) {
        for (int i = lane; i < nv; i += 64) {
            f32x4 v = src[i];
            if (H && mode == 1) { v[0] = (float)(_Float16)v[0]; v[1] = (float)(_Float16)v[1]; v[2] = (float)(_Float16)v[2]; v[3] = (float)(_Float16)v[3]; }
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        ss = wave_sum(ss);
    }
    if (mode == 2) {
        const float den = sqrtf(ss) + 1e-12f;
        for (int i = lane; i < nv; i += 64) {
            f32x4 v = src[i];
            v[0] = v[0] / den; v[1] = v[1] / den; v[2] = v[2] / den; v[3] = v[3] / den;
            put(i, v);
        }
    } else {
        if (H || reinterpret_cast<const void*>(in) != reinterpret_cast<const void*>(out))
            for (int i = lane; i < nv; i += 64) put(i, src[i]);
        if (mode == 1 && lane == 0) ynorm[row] = ss;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_gather_rows(const T* __restrict__ db, const int64_t* __restrict__ idx,
                                                     int64_t n_out, int64_t ntotal, int64_t id_base, int dim,
                                                     float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= n_out) return;
    const int64_t r = idx[o] - id_base;
    const bool ok = idx[o] >= 0 && r >= 0 && r < ntotal;
    f32x4* dst = reinterpret_cast<f32x4*>(out + o * dim);
    const T* src = db + (ok ? r : 0) * dim;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < (dim >> 2); i += 64) {
        f32x4 v = z;
        if (ok) {
            if constexpr (sizeof(T) == 2) {
                const f16x4 h4 = reinterpret_cast<const f16x4*>(src)[i];
                v = f32x4{(float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]};       // fp16 -> fp32 is exact
            } else {
                v = reinterpret_cast<const f32x4*>(src)[i];
            }
        }
        dst[i] = v;
    }
}

// |y|^2 of stored rows (fp32 or fp16): used when a snapshot is loaded into an L2 store
template <typename T>
__global__ __launch_bounds__(256) void k_row_sqnorm(const T* __restrict__ rows, int64_t n, int dim, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float ss = 0.f;
    for (int i = lane; i < dim; i += 64) { const float v = (float)rows[r * dim + i]; ss += v * v; }
    ss = wave_sum(ss);
    if (lane == 0) out[r] = ss;
}

// exclusion + compaction of search hits (retrieve_similar_vectors, pipeline.py:491-515): one thread per query row
__global__ __launch_bounds__(256) void k_filter_topk(const float* __restrict__ in_dist, const int64_t* __restrict__ in_idx,
                                                     int64_t nq, int k_in, int k_keep, const int64_t* __restrict__ tags,
                                                     int64_t ntotal, int64_t id_base, const int64_t* __restrict__ excl,
                                                     int64_t n_excl, float* __restrict__ out_dist, int64_t* __restrict__ out_idx) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    int kept = 0;
    for (int j = 0; j < k_in && kept < k_keep; ++j) {
        const int64_t id = in_idx[q * k_in + j];
        const int64_t r = id - id_base;
        if (id < 0 || r < 0 || r >= ntotal) continue;
        bool drop = false;
        if (n_excl > 0) {
            const int64_t t = tags[r];
            int64_t lo = 0, hi = n_excl;                    // binary search in the sorted exclusion set
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (excl[mid] < t) lo = mid + 1; else hi = mid;
            }
            drop = lo < n_excl && excl[lo] == t;
        }
        if (drop) continue;
        out_idx[q * k_keep + kept] = id;
        out_dist[q * k_keep + kept] = in_dist[q * k_in + j];
        ++kept;
    }
    for (; kept < k_keep; ++kept) {
        out_idx[q * k_keep + kept] = -1;
        out_dist[q * k_keep + kept] = __int_as_float(0x7fc00000);   // NaN, as pipeline.py:515
    }
}

// out[r] = the k-th largest of in[r][0..m) (NaN ranks lowest).  One wave per row, a lane holds up to KTH_PER_LANE values; k - 1 rounds
// of "wave-wide maximum, its first holder drops it".  The sharded search's bound: m = G k lower bounds per query.
constexpr int KTH_PER_LANE = 20;         // m <= 1280 (8 shards x k = 128, or 40 x 32)
// Value j = g per_group + i of row r sits at in[(g n + r) per_group + i]: [groups][n][per_group], the layout an all-gather of the
// shards' [n][k] bound blocks produces (groups = 1: plain [n][m]).
__global__ __launch_bounds__(256) void k_kth_largest(const float* __restrict__ in, int64_t n, int groups, int per_group, int k,
                                                     float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int m = groups * per_group;
    float v[KTH_PER_LANE];
#pragma unroll
    for (int i = 0; i < KTH_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float x = c < m ? in[((int64_t)(c / per_group) * n + r) * per_group + c % per_group] : -INFINITY;
        v[i] = x == x ? x : -INFINITY;
    }
    float best = -INFINITY;
    for (int round = 0; round < k; ++round) {
        float mine = v[0];
#pragma unroll
        for (int i = 1; i < KTH_PER_LANE; ++i) mine = fmaxf(mine, v[i]);
        best = wave_max(mine);
        if (round + 1 == k) break;
        // the lowest lane holding `best` drops one copy of it
        const unsigned long long holders = __ballot(mine == best);
        if (holders == 0ull) break;                                   // (only when everything left is -inf)
        if (lane == __ffsll((long long)holders) - 1) {
            bool done = false;
#pragma unroll
            for (int i = 0; i < KTH_PER_LANE; ++i)
                if (!done && v[i] == best) { v[i] = -INFINITY; done = true; }
        }
    }
    if (lane == 0) out[r] = best;
}

constexpr size_t knn_lds_bytes() { return sizeof(float) * (2 * KT_M * KT_LD + 2 * KT_N * KT_LD); }
static_assert(KT_N * KS_LD <= 2 * KT_M * KT_LD + 2 * KT_N * KT_LD, "score tile must fit in the tile buffers");

}  // namespace

template <typename KeyT>
static int merge_lists(int metric, const KeyT* in_key, const int64_t* in_idx, int n_parts, int64_t nq, int k, float* out_dist,
                       int64_t* out_idx, double* out_key, int device, void* stream) {
    RADAD_REQUIRE(metric >= 0 && metric <= 2, "radad_topk_merge: bad metric");
    RADAD_REQUIRE(n_parts >= 1 && n_parts <= 4096 && k >= 1 && k <= RADAD_KNN_MAX_K && nq >= 0, "radad_topk_merge: bad shape");
    if (nq == 0) return RADAD_OK;
    RADAD_REQUIRE(in_key && in_idx && out_dist && out_idx, "radad_topk_merge: NULL buffer");
    DeviceGuard g(device);
    ListMergeParams<KeyT> m;
    m.key = in_key; m.idx = in_idx; m.n_parts = n_parts; m.k = k; m.l2 = metric == RADAD_METRIC_L2; m.nq = nq;
    m.out_dist = out_dist; m.out_idx = out_idx; m.out_key = out_key;
    hipLaunchKernelGGL(k_merge_lists<KeyT>, dim3((unsigned)ceil_div64(nq, 4)), dim3(256), 4 * n_parts * sizeof(int),
                       (hipStream_t)stream, m);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

// what phase 2 of a search needs from phase 1 (knn_search_phase1 / _phase2 below)
struct SearchCtx {
    bool valid = false;
    int64_t nq = 0;
    int k = 0, l2 = 0, cslot = 0, n_parts = 0, plen = 0, cap = 0, xgroup = 1;
    bool cert = false, emit = false, use_floor = false;
    bool small_lists = false;          // the streaming kernels' output: one SORTED list per workgroup and query (k_refine_small)
    bool canonical = false;            // the scan's scores estimate q.y / -|q - y|^2 themselves (comparable across shards); the fp32
                                       // kernels' L2 score 2 q.y - |y|^2 lacks the -|q|^2: a cross-shard bound does not apply to it
    const float* q_use = nullptr;       // fp32 queries as the re-rank reads them (caller's buffer, or the workspace's normalised copy)
    bool have_ak = false;              // k_kth_floor left a_k in the workspace (o_ak)
    bool hi_tile = false;              // the certified f16 tile scan took this search
    size_t o_ak = 0;
    size_t o_eps = 0, o_thr = 0, o_cnt = 0, o_fcount = 0, o_fsel = 0, o_ps = 0, o_pi = 0, o_xk = 0, o_xi = 0;
};

// ---- handle -------------------------------------------------------------------------------------------
struct radad_knn_s {
    int dim = 0, metric = 0, device = 0;
    int64_t id_base = 0;
    int64_t ntotal = 0, capacity = 0;
    int f16 = 0;              // 1: rows are stored as IEEE fp16 (config.use_float16, vector_database.py:80)
    char* rows = nullptr;     // [capacity, dim] fp32 or fp16
    float* ynorm = nullptr;
    // f16 "hi plane" of an fp32 store for the certified scan (knn_hi.inc): built lazily, rows [0, hi_rows) are current
    _Float16* hi = nullptr;      // [hi_cap][dim]
    float* rscale = nullptr;     // [hi_cap] 2^-e per row (nullptr for cosine stores: one scale, 2^-14)
    int64_t hi_rows = 0, hi_cap = 0;
    int hi_off = 0;              // 1: disabled (RADAD_KNN_OPT_HI_PLANE 0 / RADAD_KNN_HI=0, or its allocation failed)
    // kernel choices of this handle (radad_knn_set_option; the environment names are announced overrides of the defaults set at creation)
    int opt_centre = -1;         // -1: decided per store from |mean|^2 / mean |y|^2; 0 never, 1 always
    int opt_smallq_hi = 1;       // small batches stream the f16 plane (0: the fp32 rows)
    int opt_wide_min_q = 17;     // smallest batch that takes the 256-query tile scan
    int opt_dense = 1;           // stores of <= RF_STAGE_MAX rows: all scores + select (k_knn_dense) instead of the register-list kernels
    size_t last_o_cnt = 0, last_o_thr = 0; int64_t last_emit_nq = 0;   // the last tile-scan search's candidate counters / floors in the workspace
    int opt_live_floor = 0;      // 1: the tile scan covers the store in ONE launch and raises its admission floors inside it; 0 (default): one
                                 // launch per phase.  Measured on one box, three alternations (profiles/r5_ab_*.txt): the one-launch form is
                                 // 1 % slower in the scan at 1 M x 512 (0.862-0.872 against 0.853-0.860 ms; the step equal within noise: it
                                 // saves k_kth_floor), 1.4 % at 10 M x 512, 4 % at 50 M x 256 -- ~0.2-0.3 us per tile for the progress atomic
                                 // and the floor reload, which go to memory past the L2s; the ramps and tails it removes are smaller than that
    int last_scan_phases = 0;    // (radad_knn_last_scan_phases)
    unsigned* stat = nullptr;    // device [3] float bits: max |y'|, max |y' - yh| (y' = y - mu when the plane is centred, else y) and
                                 // max |y| over rows [0, stat_rows)
    int64_t stat_rows = 0;
    // centred plane (decided when the plane is built, from the rows the store holds then): the plane holds f16((y - mu) 2^e)
    float* cmu = nullptr;        // [dim] the common component, or nullptr: plane of the rows themselves
    float mu_norm = 0.f, mu_sq = 0.f;   // |mu| (rounded up: it enters error bounds) and |mu|^2 as summed on the device
    float* rbias = nullptr;      // [hi_cap] per-row bias of a centred plane: |y - mu|^2 (L2) or mu.y (IP / cosine)
    float* kacc = nullptr;       // K-split tile scan (small stores of wide rows): partial accumulators of the two halves of a tile ...
    int* kflag = nullptr;        // ... and the arrival counters (zero between launches)
    int64_t kacc_tiles = 0;
    int* xarrive = nullptr;      // exact pass: arrival counters of the query groups (zero between launches)
    int64_t xarrive_cap = 0;
    double* rs_key = nullptr;    // k_refine_small: [SQ_NQ][KNN_CERT_CAP] float64 keys of the candidates, where the query's workgroups meet
    int* rs_count = nullptr;     // ... and [SQ_NQ] arrival counters (zero between launches)
    int uniform_e = HI_E_PER_ROW; // one power-of-two scale 2^e for every row of the plane (rows of one magnitude), or HI_E_PER_ROW
    // queries the certificate rejected in the most recent search: counted on the device, copied to pinned host memory
    // behind the search (no synchronisation inside search); feeds the adaptive choice below and radad_knn_last_recheck
    // The counters of search i land in slot i % 2 of pinned host memory, written by k_exact_scan together with the search's batch size
    // and, last, its stamp (sequence number + 1).  Every later search looks at BOTH slots without waiting and acts on any report it
    // has not consumed yet (stamp_seen) that belongs to a search issued since the last tuning change (tuned_at) -- however far the
    // host runs ahead of the device (round 4 acted only on the report of exactly search_seq - 2: a host more than two searches
    // ahead, e.g. a queued bench loop, never saw one).  WHEN a report is first seen depends on host / device timing, so
    // the search at which a handle retunes may differ between runs; what a search returns never does (exact by construction).
    int* host_count = nullptr;   // pinned [2][8]: rejected queries, sum of candidates, rejections by reason x 4, [6] the report's stamp, [7] its batch size
    int stamp_seen[2] = {0, 0};  // stamp of the last report consumed from each slot
    // LOOKING before the exact pass.  The exact float64 kernel costs nq x ntotal x dim multiply-adds for a fully rejected batch: 10 s for
    // 10 240 queries against 10 M x 512 (BASELINE config 4 with a store whose planted rows neighbour EVERY query: 73 % of the batch
    // overflowed its 1024 candidates), and a host that queues searches gets no report before it has queued them all.  So the first
    // large tile-scan search after anything that can change the outcome (handle creation, a plane (re)build, a tuning step, the end of
    // an fp32 fallback) WAITS for its re-rank -- one stream synchronisation -- and, when more than a quarter of the batch was rejected,
    // retunes and runs again at once instead of entering the exact kernel.  A search that passes clears the flag: steady state never
    // synchronises.
    bool verify_next = true;
    int hi_fail_streak = 0;      // consecutive returns from the fp32 fallback that were rejected again: the fallback doubles (8, 16, ... 512 searches)
    int64_t verified_retries = 0;
    int64_t reports_consumed = 0;   // (radad_knn_tuning_info)
    bool owned_serial = false;   // the handle is private to another object that orders its calls itself (the IVF index's centroid store): no ev_done
    int* host_count_dev = nullptr;   // the same memory as the device sees it
    hipEvent_t ev_done = nullptr;
    hipEvent_t ev_begun = nullptr;      // end of radad_knn_search_begin's device work: _finish waits for it (it may run on another stream)
    bool have_last = false;
    hipStream_t last_stream = nullptr;   // where the last search was enqueued (the next one on ANOTHER stream orders itself behind it; the statistics calls synchronise on it)
    int64_t count_nq[2] = {0, 0};    // batch size of the search a slot belongs to
    uint64_t search_seq = 0;     // certified searches so far
    int hi_skip = 0;             // searches left on the fp32 kernels after the certified scan rejected too many queries
    // Re-deciding the plane (its centre mu and its scale are decided from the rows the store holds WHEN IT IS BUILT; add_vectors_batch
    // appends 10 000 rows at a time, vector_database.py:134-138): it is dropped and rebuilt -- 0.84 ms per million rows, one
    // synchronisation -- when the store has doubled since the decision, when rows appended since then measure 8x beyond what the
    // decision saw (max |y'| or the rounding residual: the f16 scale has 8x headroom), or when a batch was mostly rejected and rows
    // have been appended since (otherwise the same plane would come back: the fp32 fallback stays the remedy)
    int64_t plane_decided_rows = 0;
    float plane_stat[2] = {0.f, 0.f};     // max |y'|, max |y' - yh| right after the plane was built
    bool replan = false;
    int plane_rebuilds = 0;      // (radad_knn_plane_info: how often the plane was re-decided)
    uint64_t tuned_at = 0;       // search_seq at the last change of the plane / of cap_boost: reports of EARLIER searches are not acted on
    int cap_boost = 1;           // candidate buffers of the tile scan are 1024 entries per query x this (1 or 4): raised once when a batch
                                 // was mostly rejected (a store of several clusters has a large |y - mu|, hence a large eps: more rows
                                 // pass the floor than 1024) -- before the handle resorts to the fp32 kernels
    size_t esize() const { return f16 ? 2 : 4; }
    size_t row_bytes() const { return (size_t)dim * esize(); }
    // search workspace (grown on demand, reused)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    int last_qtiles = 0, last_splits = 0, last_threads = KNN_THREADS;
    int last_kind = RADAD_SCAN_F32_TILE;   // which scan kernel the last search ran (radad_knn_last_scan_kind)
    int last_scan_launches = 1;            // ... in how many launches
    EventRing prof;
    SearchCtx pending;           // radad_knn_search_begin without its _finish yet
    std::mutex mu;
};

static void knn_drop_plane(radad_knn_t h);

static int knn_realloc(radad_knn_t h, int64_t cap) {
    char* nrows = nullptr;
    float* nnorm = nullptr;
    if (hipMalloc(&nrows, (size_t)cap * h->row_bytes()) != hipSuccess) {
        radad_set_error("hipMalloc of %lld x %d rows failed", (long long)cap, h->dim);
        return RADAD_ENOMEM;
    }
    if (h->metric == RADAD_METRIC_L2 && hipMalloc(&nnorm, (size_t)cap * sizeof(float)) != hipSuccess) {
        (void)hipFree(nrows);
        radad_set_error("hipMalloc of %lld norms failed", (long long)cap);
        return RADAD_ENOMEM;
    }
    (void)hipDeviceSynchronize();       // searches on other streams may still read the old buffers
    if (h->ntotal > 0) {
        RADAD_HIP_CHECK(hipMemcpy(nrows, h->rows, (size_t)h->ntotal * h->row_bytes(), hipMemcpyDeviceToDevice));
        if (nnorm) RADAD_HIP_CHECK(hipMemcpy(nnorm, h->ynorm, (size_t)h->ntotal * sizeof(float), hipMemcpyDeviceToDevice));
    }
    if (h->rows) (void)hipFree(h->rows);
    if (h->ynorm) (void)hipFree(h->ynorm);
    h->rows = nrows;
    h->ynorm = nnorm;
    h->capacity = cap;
    knn_drop_plane(h);                           // rebuilt lazily for the new capacity
    return RADAD_OK;
}

// geometric growth so repeated add() calls (vector_database.py:134-138 adds 10 000 rows at a time) stay O(n)
static int knn_grow(radad_knn_t h, int64_t need) {
    if (need <= h->capacity) return RADAD_OK;
    int64_t cap = std::max<int64_t>(need, h->capacity + h->capacity / 2);
    return knn_realloc(h, std::max<int64_t>(cap, 1024));
}

static int knn_workspace(radad_knn_t h, size_t bytes) {
    if (bytes <= h->ws_bytes) return RADAD_OK;
    if (h->ws) (void)hipFree(h->ws);
    h->ws = nullptr;
    h->ws_bytes = 0;
    if (hipMalloc(&h->ws, bytes) != hipSuccess) {
        radad_set_error("hipMalloc of %zu workspace bytes failed", bytes);
        return RADAD_ENOMEM;
    }
    h->ws_bytes = bytes;
    return RADAD_OK;
}

// column means of the first `m` rows, deterministic: KM_SLICES partial sums per column in fixed order, then one block adds them
// up, and measures |mu|^2 against the rows' mean |y|^2 (out2[0], out2[1]) for the host's decision
constexpr int KM_SLICES = 64;
__global__ __launch_bounds__(256) void k_col_partial(const float* __restrict__ rows, int64_t m, int64_t row_stride, int dim, float* __restrict__ part /*[KM_SLICES][dim + 1]*/) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int64_t per = (m + KM_SLICES - 1) / KM_SLICES;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(r0 + per, m);
    float s = 0.f, ss = 0.f;
    if (c < dim)
        for (int64_t r = r0; r < r1; ++r) { const float v = rows[r * row_stride * dim + c]; s += v; ss = fmaf(v, v, ss); }      // (sample row r = store row r x stride)
    if (c < dim) part[(int64_t)blockIdx.y * (dim + 1) + c] = s;
    // sum of squares of this block's columns and rows -> one value per (slice, column block), added up by k_col_final
    __shared__ float red[256];
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[(int64_t)KM_SLICES * (dim + 1) + blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void k_col_final(const float* __restrict__ part, int64_t m, int dim, int n_cblocks, float* __restrict__ mu,
                                                   float* __restrict__ out2) {
    __shared__ float red[256];
    float mm = 0.f;
    for (int c = threadIdx.x; c < dim; c += 256) {
        float s = 0.f;
        for (int sl = 0; sl < KM_SLICES; ++sl) s += part[(int64_t)sl * (dim + 1) + c];
        const float v = s / (float)m;
        mu[c] = v;
        mm = fmaf(v, v, mm);
    }
    red[threadIdx.x] = mm;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float mu2 = red[0];
    __syncthreads();
    float ss = 0.f;
    for (int i = threadIdx.x; i < KM_SLICES * n_cblocks; i += 256) ss += part[(int64_t)KM_SLICES * (dim + 1) + i];
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { out2[0] = mu2; out2[1] = red[0] / (float)m; }
}

// max |x - mu| over the elements of the first m rows (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ rows, const float* __restrict__ mu, int64_t m, int64_t row_stride, int dim,
                                                unsigned* __restrict__ out /*[2]: max, and min over rows of the row maximum*/) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    float mx = 0.f;
    for (int i = lane; i < dim; i += 64) mx = fmaxf(mx, fabsf(rows[row * row_stride * dim + i] - (mu ? mu[i] : 0.f)));
    mx = wave_max(mx);
    if (lane == 0) {        // (atomics only when the row moves an extreme it can see: same-address atomics serialise)
        const unsigned u = __float_as_uint(mx);
        if (u > *reinterpret_cast<volatile unsigned*>(&out[0])) atomicMax(&out[0], u);
        if (u < *reinterpret_cast<volatile unsigned*>(&out[1])) atomicMin(&out[1], u);
    }
}

static void knn_drop_plane(radad_knn_t h) {
    if (h->hi) (void)hipFree(h->hi);
    if (h->rscale) (void)hipFree(h->rscale);
    if (h->cmu) (void)hipFree(h->cmu);
    if (h->rbias) (void)hipFree(h->rbias);
    h->hi = nullptr; h->rscale = nullptr; h->cmu = nullptr; h->rbias = nullptr; h->mu_norm = 0.f; h->mu_sq = 0.f;
    h->hi_rows = 0; h->hi_cap = 0; h->uniform_e = HI_E_PER_ROW;
}

// Decide whether the plane is CENTRED and compute mu (one-off, when the plane is built; synchronises).  Centred when the common
// component carries a noticeable part of the rows' energy (|mu|^2 >= 2 % of the mean |y|^2): pooled encoder embeddings do (all
// positive activations around a shared mean), random directions do not -- and an un-centred cosine plane keeps its one scale for
// the whole store, i.e. the scan variant without any per-score arithmetic.
constexpr int64_t KM_MAX_ROWS = 65536;
constexpr float KM_CENTRE_FRACTION = 0.02f;
static bool knn_choose_centre(radad_knn_t h, hipStream_t st) {
    const int64_t m = std::min<int64_t>(h->ntotal, KM_MAX_ROWS);
    if (m < 64 || h->f16) return false;
    const int centre_env = h->opt_centre;
    if (centre_env == 0) return false;
    const int ncb = (h->dim + 255) / 256;
    float *part = nullptr, *mu = nullptr, *out2 = nullptr;
    const size_t pbytes = ((size_t)KM_SLICES * (h->dim + 1) + (size_t)KM_SLICES * ncb) * sizeof(float);
    bool ok = hipMalloc(&part, pbytes) == hipSuccess && hipMalloc(&mu, (size_t)h->dim * sizeof(float)) == hipSuccess &&
              hipMalloc(&out2, 2 * sizeof(float)) == hipSuccess;
    float host2[2] = {0.f, 1.f};
    if (ok) {
        // (the sample is spread over the WHOLE store: its first rows alone say nothing about rows appended later)
        hipLaunchKernelGGL(k_col_partial, dim3(ncb, KM_SLICES), dim3(256), 0, st, (const float*)h->rows, m, std::max<int64_t>(1, h->ntotal / m), h->dim, part);
        hipLaunchKernelGGL(k_col_final, dim3(1), dim3(256), 0, st, (const float*)part, m, h->dim, ncb, mu, out2);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(host2, out2, sizeof(host2), hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    }
    if (part) (void)hipFree(part);
    if (out2) (void)hipFree(out2);
    const bool centre = ok && (centre_env == 1 || host2[0] >= KM_CENTRE_FRACTION * host2[1]) && host2[0] > 0.f && host2[0] < INFINITY;
    if (!centre) { if (mu) (void)hipFree(mu); (void)hipGetLastError(); return false; }
    h->cmu = mu;
    h->mu_norm = sqrtf(host2[0]) * (1.0f + 0x1p-10f);
    h->mu_sq = host2[0];
    return true;
}

// store statistics (max |y|, max |y - yh|) and, when wanted, the f16 hi plane, brought up to date with the rows
// (appends only touch the new rows).  Returns false when the plane was wanted but is not available.
static bool knn_ensure_hi(radad_knn_t h, hipStream_t st, bool want_plane) {
    if (!h->stat) {
        if (hipMalloc(&h->stat, 4 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return false; }
        (void)hipMemsetAsync(h->stat, 0, 4 * sizeof(unsigned), st);
        h->stat_rows = 0;
    }
    const bool plane = want_plane && !h->hi_off && !h->f16 && h->dim % 64 == 0;
    if (plane && h->hi && h->hi_cap == h->capacity && h->ntotal > h->plane_decided_rows &&
        (h->replan || h->ntotal >= 2 * h->plane_decided_rows)) {
        (void)hipDeviceSynchronize();
        knn_drop_plane(h);
        ++h->plane_rebuilds;
        h->tuned_at = h->search_seq;
    }
    if (plane) h->replan = false;
    const bool fresh = plane && (h->hi_cap != h->capacity || !h->hi);
    if (fresh) {
        (void)hipDeviceSynchronize();
        knn_drop_plane(h);
        h->verify_next = true;                           // a new plane: the next large search looks at its certificate before the exact pass
        const bool centred = knn_choose_centre(h, st);
        // One scale for the whole plane when the rows are of one magnitude (the largest row maximum within 2^6 of the smallest over
        // the rows the store holds now: clip embeddings are; rows scaled over 2^22, as a test does, are not): the scan then needs
        // no per-score arithmetic.  2^e puts the largest element in [2^12, 2^13): a later row up to 8x larger still fits f16, beyond
        // that it saturates and its (measured) residual rejects queries.  Un-centred cosine rows have |x| <= 1: e = 14 as before.
        if (h->metric != RADAD_METRIC_COSINE || centred) {
            unsigned* mm = nullptr;
            unsigned host[2] = {0u, 0x7f800000u};
            const int64_t m = std::min<int64_t>(h->ntotal, KM_MAX_ROWS);
            if (m > 0 && hipMalloc(&mm, 2 * sizeof(unsigned)) == hipSuccess) {
                (void)hipMemcpyAsync(mm, host, sizeof(host), hipMemcpyHostToDevice, st);
                hipLaunchKernelGGL(k_absmax, dim3((unsigned)ceil_div64(m, 4)), dim3(256), 0, st, (const float*)h->rows, (const float*)h->cmu, m,
                                   std::max<int64_t>(1, h->ntotal / m), h->dim, mm);
                if (hipMemcpyAsync(host, mm, sizeof(host), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
                    float mx, mn;
                    memcpy(&mx, &host[0], 4); memcpy(&mn, &host[1], 4);
                    if (mx > 0.f && mx < INFINITY && mn > 0.f && mx <= mn * 64.f) {
                        int fe;
                        (void)frexpf(mx, &fe);
                        h->uniform_e = std::max(-100, std::min(100, 13 - fe));
                    }
                }
                (void)hipFree(mm);
            }
            (void)hipGetLastError();
        }
        const bool per_row = (h->metric != RADAD_METRIC_COSINE || centred) && h->uniform_e == HI_E_PER_ROW;
        if (hipMalloc(&h->hi, (size_t)h->capacity * h->dim * 2) != hipSuccess ||
            (per_row && hipMalloc(&h->rscale, (size_t)h->capacity * sizeof(float)) != hipSuccess) ||
            (centred && hipMalloc(&h->rbias, (size_t)h->capacity * sizeof(float)) != hipSuccess)) {
            (void)hipGetLastError();
            knn_drop_plane(h);
            h->hi_off = 1;              // no room for the plane: stay on the fp32 kernels
        } else {
            h->hi_cap = h->capacity;
        }
        // the statistics describe the operands of the plane (y - mu when centred): start them over with it
        (void)hipMemsetAsync(h->stat, 0, 4 * sizeof(unsigned), st);
        h->stat_rows = 0;
    }
    const bool have_plane = plane && h->hi;
    const int64_t from = have_plane ? std::min(h->hi_rows, h->stat_rows) : h->stat_rows;
    if (from < h->ntotal) {
        HiRowsParams hp;
        hp.in = h->rows + (size_t)from * h->row_bytes(); hp.in_f16 = h->f16;
        hp.hi = have_plane ? h->hi + (size_t)from * h->dim : nullptr;
        hp.scale_out = (have_plane && h->rscale) ? h->rscale + from : nullptr;
        hp.stat_max = h->stat; hp.eps_out = nullptr; hp.ystat = nullptr;
        hp.norm_out = nullptr; hp.zero_flags = nullptr; hp.zero_counters = nullptr;
        hp.n = h->ntotal - from; hp.dim = h->dim;
        // un-centred cosine rows have |x| <= 1: one scale for the whole store (no per-score arithmetic in the scan);
        // an fp16 store is its own plane: statistics only, un-scaled
        hp.fixed_e = h->f16 ? 0 : ((h->metric == RADAD_METRIC_COSINE && !h->cmu) ? 14 : h->uniform_e);
        hp.l2 = h->metric == RADAD_METRIC_L2 ? 1 : 0; hp.exact_ops = h->f16 ? 1 : 0;
        // (a plane that exists keeps its centring for the statistics-only calls too: the statistics are those of ITS operands)
        hp.mu = h->cmu; hp.mu_norm = h->mu_norm; hp.mu_sq = h->mu_sq; hp.biased = 0;
        hp.bias_out = (have_plane && h->rbias) ? h->rbias + from : nullptr; hp.qconst_out = nullptr;
        hipLaunchKernelGGL(k_hi_rows, dim3((unsigned)ceil_div64(hp.n, 4)), dim3(256), 0, st, hp);
        if (hipGetLastError() != hipSuccess) return false;
        h->stat_rows = h->ntotal;
        if (have_plane) h->hi_rows = h->ntotal;
        if (have_plane) {
            // what the plane's operands measure now (only searches that follow an append come through here: one small read-back)
            float host[2] = {0.f, 0.f};
            if (hipMemcpyAsync(host, h->stat, sizeof(host), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
                if (fresh) { h->plane_stat[0] = host[0]; h->plane_stat[1] = host[1]; h->plane_decided_rows = h->ntotal; }
                else if ((h->plane_stat[0] > 0.f && host[0] > 8.f * h->plane_stat[0]) || (h->plane_stat[1] > 0.f && host[1] > 8.f * h->plane_stat[1])) {
                    h->replan = true;                      // rows beyond what the scale was chosen for: decide again, now
                    return knn_ensure_hi(h, st, want_plane);
                }
            } else (void)hipGetLastError();
        }
    }
    return !want_plane || have_plane || h->f16;
}

// wide kernel: 256-query tiles, one workgroup per CU; two rounds of workgroups keep the tail short.  (The list-based scan of
// rounds 1-2 also needed >= 64 chunks so that no 16-entry list was used up; the emit-mode scan has no lists: a sharded batch of
// 32 query tiles now runs 16 chunks of 31 tiles instead of 64 chunks of 8, i.e. a quarter of the per-workgroup start-up cost.)
static void knn_geometry_wide(int64_t n, int64_t nq, int* n_qtiles, int* n_splits, int64_t* chunk_rows) {
    const int qt = (int)ceil_div64(nq, KW_N);
    const int64_t tiles = ceil_div64(n, KW_M);
    int64_t want = ceil_div64(512, qt);
    want = std::min<int64_t>(want, tiles);
    want = std::max<int64_t>(8, ceil_div64(want, 8) * 8);
    want = std::min<int64_t>(want, 1024);
    int64_t tiles_per = ceil_div64(tiles, want);
    // A launch takes (rounds of 256 workgroups) x (tiles per chunk) tile times, and both factors round up.  Round 4 always took
    // ceil(512 / query tiles) chunks ("two rounds keep the tail short"): right for 4 query tiles x 1 M rows (2 x 31 = 1 x 62), wrong
    // where the rounding bites -- BASELINE config 2's 100 k rows = 391 tiles x 4 query tiles: 128 chunks of 4 tiles = 392 workgroups =
    // 2 rounds x 4 = 8 tile times, 64 chunks of 7 = 224 workgroups = ONE round of 7; configs 4 / 5 at full size, 40 query tiles: 16
    // chunks = 640 workgroups = 2.5 -> 3 rounds x 2442 tiles, 32 chunks = 1280 = exactly 5 rounds x 1221 (-17 %).  So: the cheapest by
    // that count among one round's worth of chunks and `want`, `want` + 8, ... 4 x `want`, when it saves 3 % or more.
    if (qt <= 256) {
        auto cost = [&](int64_t per) { return ceil_div64((int64_t)qt * ceil_div64(tiles, per), 256) * per; };
        int64_t best = want, best_per = tiles_per, best_cost = cost(tiles_per);
        auto consider = [&](int64_t splits) {
            splits = std::min<int64_t>(std::max<int64_t>(8, splits / 8 * 8), std::min<int64_t>(1024, std::max<int64_t>(8, ceil_div64(tiles, 8) * 8)));
            const int64_t per = ceil_div64(tiles, splits);
            const int64_t c = cost(per);
            if (c * 100 < best_cost * 97) { best = splits; best_per = per; best_cost = c; }      // (a gain under 3 % is not worth leaving two rounds)
        };
        consider(256 / qt);
        for (int64_t sp = want + 8; sp <= 4 * want; sp += 8) consider(sp);
        want = best; tiles_per = best_per;
    }
    *n_qtiles = qt;
    *n_splits = (int)want;
    *chunk_rows = tiles_per * KW_M;
}

extern "C" {

int radad_knn_create(int dim, int metric, int device, int64_t id_base, radad_knn_t* out) {
    return radad_knn_create_ex(dim, metric, RADAD_STORE_F32, device, id_base, out);
}

int radad_knn_create_ex(int dim, int metric, int store_dtype, int device, int64_t id_base, radad_knn_t* out) {
    RADAD_REQUIRE(out != nullptr, "radad_knn_create: out is NULL");
    RADAD_REQUIRE(store_dtype == RADAD_STORE_F32 || store_dtype == RADAD_STORE_F16, "radad_knn_create: unsupported store dtype %d", store_dtype);
    RADAD_REQUIRE(dim > 0 && (dim % 4) == 0, "radad_knn_create: dim must be a positive multiple of 4 (got %d)", dim);
    RADAD_REQUIRE(metric >= 0 && metric <= 2, "radad_knn_create: unsupported metric %d", metric);
    int ndev = 0;
    RADAD_HIP_CHECK(hipGetDeviceCount(&ndev));
    RADAD_REQUIRE(device >= 0 && device < ndev, "radad_knn_create: device %d not in [0,%d)", device, ndev);
    radad_knn_s* h = new (std::nothrow) radad_knn_s();
    if (!h) { radad_set_error("out of host memory"); return RADAD_ENOMEM; }
    h->dim = dim; h->metric = metric; h->device = device; h->id_base = id_base; h->f16 = store_dtype == RADAD_STORE_F16;
    // RADAD_KNN_HI=0 keeps every search on the fp32 kernels (bench.py --scan f32); read once, at creation
    // defaults of the handle's options (radad_knn_set_option changes them); an environment variable of the old name overrides the
    // default for handles created from now on, announced once on stderr -- for experiments only
    { const char* e = radad_env_override("RADAD_KNN_HI", "0 keeps every search of stores created from now on off the certified f16 scan (fp32 kernels: ~8x slower scans)"); h->hi_off = (e && atoi(e) == 0) ? 1 : 0; }
    { const char* e = radad_env_override("RADAD_KNN_CENTRE", "0 never centres the f16 plane, 1 always does (default: decided per store from |mean|^2 / mean |y|^2)"); h->opt_centre = e ? atoi(e) : -1; }
    { const char* e = radad_env_override("RADAD_KNN_SMALLQ_HI", "0 makes small batches stream the fp32 rows instead of the f16 plane (twice the bytes)"); h->opt_smallq_hi = e ? atoi(e) : 1; }
    { const char* e = radad_env_override("RADAD_WIDE_MIN_Q", "smallest batch that takes the 256-query tile scan (default 17)"); h->opt_wide_min_q = e ? atoi(e) : SQ_NQ + 1; }
    {
        DeviceGuard g(device);
        if (hipHostMalloc(reinterpret_cast<void**>(&h->host_count), 24 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_begun, hipEventDisableTiming) != hipSuccess) {
            radad_set_error("radad_knn_create: pinned counter / events could not be created");
            radad_knn_destroy(h);
            return RADAD_EHIP;
        }
        memset(h->host_count, 0, 24 * sizeof(int));
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&h->host_count_dev), h->host_count, 0) != hipSuccess) {
            radad_set_error("radad_knn_create: pinned counter is not device-visible");
            radad_knn_destroy(h);
            return RADAD_EHIP;
        }
    }
    *out = h;
    return RADAD_OK;
}

int radad_knn_destroy(radad_knn_t h) {
    if (!h) return RADAD_OK;
    {
        DeviceGuard g(h->device);
        if (h->rows) (void)hipFree(h->rows);
        if (h->ynorm) (void)hipFree(h->ynorm);
        if (h->ws) (void)hipFree(h->ws);
        if (h->hi) (void)hipFree(h->hi);
        if (h->rscale) (void)hipFree(h->rscale);
        if (h->cmu) (void)hipFree(h->cmu);
        if (h->rbias) (void)hipFree(h->rbias);
        if (h->stat) (void)hipFree(h->stat);
        if (h->kacc) (void)hipFree(h->kacc);
        if (h->kflag) (void)hipFree(h->kflag);
        if (h->xarrive) (void)hipFree(h->xarrive);
        if (h->rs_key) (void)hipFree(h->rs_key);
        if (h->rs_count) (void)hipFree(h->rs_count);
        if (h->host_count) (void)hipHostFree(h->host_count);
        if (h->ev_done) (void)hipEventDestroy(h->ev_done);
        if (h->ev_begun) (void)hipEventDestroy(h->ev_begun);
        h->prof.destroy();
    }
    delete h;
    return RADAD_OK;
}

int radad_knn_dim(radad_knn_t h, int* dim) { RADAD_REQUIRE(h && dim, "NULL argument"); *dim = h->dim; return RADAD_OK; }
int radad_knn_metric(radad_knn_t h, int* metric) { RADAD_REQUIRE(h && metric, "NULL argument"); *metric = h->metric; return RADAD_OK; }
int radad_knn_ntotal(radad_knn_t h, int64_t* n) { RADAD_REQUIRE(h && n, "NULL argument"); *n = h->ntotal; return RADAD_OK; }

int radad_knn_set_option(radad_knn_t h, int option, int value) {
    RADAD_REQUIRE(h, "NULL handle");
    std::lock_guard<std::mutex> lk(h->mu);
    switch (option) {
        case RADAD_KNN_OPT_HI_PLANE:
            RADAD_REQUIRE(value == 0 || value == 1, "radad_knn_set_option: HI_PLANE takes 0 or 1");
            RADAD_REQUIRE(h->hi == nullptr || value == 1, "radad_knn_set_option: the f16 plane of this store is already built; switch it off before the first large-batch search");
            h->hi_off = value ? 0 : 1;
            return RADAD_OK;
        case RADAD_KNN_OPT_CENTRE:
            RADAD_REQUIRE(value >= -1 && value <= 1, "radad_knn_set_option: CENTRE takes -1 (decide per store), 0 or 1");
            RADAD_REQUIRE(h->hi == nullptr, "radad_knn_set_option: the f16 plane of this store is already built; choose its centring before the first large-batch search");
            h->opt_centre = value;
            return RADAD_OK;
        case RADAD_KNN_OPT_SMALLQ_HI:
            RADAD_REQUIRE(value == 0 || value == 1, "radad_knn_set_option: SMALLQ_HI takes 0 or 1");
            h->opt_smallq_hi = value;
            return RADAD_OK;
        case RADAD_KNN_OPT_WIDE_MIN_Q:
            RADAD_REQUIRE(value >= 1, "radad_knn_set_option: WIDE_MIN_Q must be >= 1");
            h->opt_wide_min_q = value;
            return RADAD_OK;
        case RADAD_KNN_OPT_DENSE:
            RADAD_REQUIRE(value == 0 || value == 1, "radad_knn_set_option: DENSE takes 0 or 1");
            h->opt_dense = value;
            return RADAD_OK;
        case RADAD_KNN_OPT_LIVE_FLOOR:
            RADAD_REQUIRE(value == 0 || value == 1, "radad_knn_set_option: LIVE_FLOOR takes 0 or 1");
            h->opt_live_floor = value;
            return RADAD_OK;
        default:
            radad_set_error("radad_knn_set_option: unknown option %d", option);
            return RADAD_EINVAL;
    }
}

int radad_knn_reserve(radad_knn_t h, int64_t capacity) {
    RADAD_REQUIRE(h, "NULL handle");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    RADAD_REQUIRE(!h->pending.valid, "radad_knn_reserve: a begun search (radad_knn_search_begin) must be finished or aborted before the store changes");
    if (capacity <= h->capacity) return RADAD_OK;
    RADAD_HIP_CHECK(hipDeviceSynchronize());
    return knn_realloc(h, capacity);
}

int radad_knn_add(radad_knn_t h, const float* rows_dev, int64_t n, void* stream) {
    RADAD_REQUIRE(h, "NULL handle");
    RADAD_REQUIRE(n >= 0, "radad_knn_add: n < 0");
    if (n == 0) return RADAD_OK;
    RADAD_REQUIRE(rows_dev, "radad_knn_add: rows is NULL");
    RADAD_REQUIRE(h->ntotal + n < (int64_t)IDX_SENTINEL, "radad_knn_add: more than 2^31-2 rows per store (shard it)");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    // (the second half of a begun search reads rows / ntotal live while its candidate ids date from the first half)
    RADAD_REQUIRE(!h->pending.valid, "radad_knn_add: a begun search (radad_knn_search_begin) must be finished or aborted before the store changes");
    hipStream_t st = (hipStream_t)stream;
    if (h->ntotal + n > h->capacity) {
        RADAD_HIP_CHECK(hipStreamSynchronize(st));   // the old buffers may still be in use on `st`
        int rc = knn_grow(h, h->ntotal + n);
        if (rc) return rc;
    }
    const int mode = h->metric == RADAD_METRIC_COSINE ? 2 : (h->metric == RADAD_METRIC_L2 ? 1 : 0);
    char* dst = h->rows + (size_t)h->ntotal * h->row_bytes();
    float* yn = h->ynorm ? h->ynorm + h->ntotal : nullptr;
    const unsigned grid = (unsigned)ceil_div64(n, 4);
    if (h->f16) hipLaunchKernelGGL(k_rows_prepare<_Float16>, dim3(grid), dim3(256), 0, st, rows_dev, (_Float16*)dst, yn, n, h->dim, mode);
    else hipLaunchKernelGGL(k_rows_prepare<float>, dim3(grid), dim3(256), 0, st, rows_dev, (float*)dst, yn, n, h->dim, mode);
    RADAD_HIP_CHECK(hipGetLastError());
    h->ntotal += n;
    return RADAD_OK;
}

int radad_knn_add_host(radad_knn_t h, const float* rows_host, int64_t n) {
    RADAD_REQUIRE(h, "NULL handle");
    if (n == 0) return RADAD_OK;
    RADAD_REQUIRE(rows_host && n > 0, "radad_knn_add_host: bad rows/n");
    float* tmp = nullptr;
    {
        DeviceGuard g(h->device);
        const size_t bytes = (size_t)n * h->dim * sizeof(float);
        if (hipMalloc(&tmp, bytes) != hipSuccess) { radad_set_error("hipMalloc of staging buffer failed"); return RADAD_ENOMEM; }
        hipError_t e = hipMemcpy(tmp, rows_host, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(tmp); radad_set_error("H2D copy failed: %s", hipGetErrorString(e)); return RADAD_EHIP; }
    }
    int rc = radad_knn_add(h, tmp, n, nullptr);
    {
        DeviceGuard g(h->device);
        hipError_t e = hipStreamSynchronize(nullptr);
        (void)hipFree(tmp);
        if (rc == RADAD_OK && e != hipSuccess) { radad_set_error("add failed: %s", hipGetErrorString(e)); rc = RADAD_EHIP; }
    }
    return rc;
}

// choose the launch geometry: enough equal splits to put >= 2 workgroups on each of the 256 CUs
static void knn_geometry(int64_t n, int64_t nq, int* n_qtiles, int* n_splits, int64_t* chunk_rows) {
    const int qt = (int)ceil_div64(nq, KT_N);
    const int64_t tiles = ceil_div64(n, KT_M);
    int64_t want = ceil_div64(512, qt);            // splits so that qt*splits ~ 512 workgroups
    want = std::min<int64_t>(want, tiles);
    want = std::max<int64_t>(8, ceil_div64(want, 8) * 8);
    want = std::min<int64_t>(want, 1024);
    const int64_t tiles_per = ceil_div64(tiles, want);
    *n_qtiles = qt;
    *n_splits = (int)want;
    *chunk_rows = tiles_per * KT_M;
}

int radad_knn_search(radad_knn_t h, const float* q_dev, int64_t nq, int k, float* out_dist_dev, int64_t* out_idx_dev,
                     void* stream) {
    return radad_knn_search_f64(h, q_dev, nq, k, out_dist_dev, out_idx_dev, nullptr, stream);
}

static int knn_search_core(radad_knn_t h, const void* q_dev, int q_dtype, int64_t nq, int k, int margin, float* out_dist_dev,
                           int64_t* out_idx_dev, double* out_key_dev, void* stream);

int radad_knn_search_f64(radad_knn_t h, const float* q_dev, int64_t nq, int k, float* out_dist_dev, int64_t* out_idx_dev,
                         double* out_key_dev, void* stream) {
    return knn_search_core(h, q_dev, RADAD_Q_F32, nq, k, KNN_MARGIN, out_dist_dev, out_idx_dev, out_key_dev, stream);
}

int radad_knn_search_ex(radad_knn_t h, const void* q_dev, int q_dtype, int64_t nq, int k, float* out_dist_dev,
                        int64_t* out_idx_dev, double* out_key_dev, void* stream) {
    RADAD_REQUIRE(q_dtype == RADAD_Q_F32 || q_dtype == RADAD_Q_BF16, "radad_knn_search_ex: unsupported query dtype %d", q_dtype);
    return knn_search_core(h, q_dev, q_dtype, nq, k, KNN_MARGIN, out_dist_dev, out_idx_dev, out_key_dev, stream);
}

}  // extern "C"

__global__ __launch_bounds__(256) void k_fill_f32(float* __restrict__ out, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v;
}

// bf16 -> fp32 (exact): queries handed over as bfloat16 (BASELINE config 5) are decoded once, then take the fp32 path
__global__ __launch_bounds__(256) void k_bf16_to_f32(const unsigned short* __restrict__ in, float* __restrict__ out, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    const u16x4 b = *reinterpret_cast<const u16x4*>(in + i);
    *reinterpret_cast<f32x4*>(out + i) = f32x4{__uint_as_float((unsigned)b[0] << 16), __uint_as_float((unsigned)b[1] << 16),
                                               __uint_as_float((unsigned)b[2] << 16), __uint_as_float((unsigned)b[3] << 16)};
}

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// K split of the tile scan over two workgroups per tile: when a launch has one tile per workgroup, at most 128 workgroups (half the
// chip) and at least 16 K steps (dim >= 1024) -- the reference's own store, 25 423 x 5376: 100 tiles of 84 K steps.  Returns 1 or 2 and
// makes sure the scratch exists (allocated once per handle; the counters start, and are left, at zero).
static int knn_tile_ksplit(radad_knn_t h, int qtiles_grid, int n_splits, int64_t chunk_rows, int64_t rows, hipStream_t st) {
    const int64_t tiles = ceil_div64(rows, KW_M);
    if (chunk_rows != KW_M || (int64_t)qtiles_grid * n_splits > 128 || h->dim * 2 / 128 < 16) return 1;
    const int64_t need = tiles * qtiles_grid;
    if ((size_t)need * 2 * 32 * KW_THREADS * sizeof(f32x4) > ((size_t)256 << 20)) return 1;      // (a spread sample over a large store: not worth 0.5 MB per tile)
    if (need > h->kacc_tiles) {
        (void)hipStreamSynchronize(st);
        if (h->kacc) (void)hipFree(h->kacc);
        if (h->kflag) (void)hipFree(h->kflag);
        h->kacc = nullptr; h->kflag = nullptr; h->kacc_tiles = 0;
        if (hipMalloc(&h->kacc, (size_t)need * 2 * 32 * KW_THREADS * sizeof(f32x4)) != hipSuccess ||
            hipMalloc(&h->kflag, (size_t)need * sizeof(int)) != hipSuccess || hipMemset(h->kflag, 0, (size_t)need * sizeof(int)) != hipSuccess) {
            (void)hipGetLastError();
            if (h->kacc) (void)hipFree(h->kacc);
            if (h->kflag) (void)hipFree(h->kflag);
            h->kacc = nullptr; h->kflag = nullptr;
            return 1;                                   // no room: the un-split launch is always possible
        }
        h->kacc_tiles = need;
    }
    return 2;
}

// The search proper.  `margin` spare entries per (query, chunk) list on the fp32 tile kernels (KNN_MARGIN for user
// searches; the IVF coarse quantiser asks for fewer so that nprobe = 32 still fits the 32-entry register lists).
//
// Scan (a filter), one of:
//   certified f16 tile scan (knn_hi.inc)   batches > 16 queries (or small batches too wide for the streaming kernel's LDS) on
//                                     stores of >= 16384 rows, k <= 128: hi plane of an fp32 store or the fp16 store itself; emits
//                                     every row at or above the query's admission floor into the query's candidate buffer
//   k_knn_hi_smallq / k_knn_f32_smallq     <= 16 queries: HBM-bound streaming of the f16 plane / the fp32 rows
//   k_knn_f32_reg / k_knn_f32         the fp32 (or fp16-operand) tile kernels, lists of k + margin
// then k_merge_refine: float64 re-rank of every candidate within 2 eps of the k-th, per-query certificate; then the
// exact float64 kernel for the queries the certificate rejected (device-driven, usually zero work).  Nothing in here
// synchronises with the host unless the workspace has to grow.
//
// Two halves around an optional exchange (radad_knn_search_begin / _finish, for row-sharded stores): phase 1 prepares the queries
// and scans, and can report per query a lower bound of the exact k-th best score of THIS store; phase 2 re-ranks -- with the
// maximum of the shards' bounds, only what can still be among the global k best -- and runs the exact kernel.

// a batch was mostly rejected by the certificate: re-decide the plane if rows were appended since it was decided, else widen the
// candidate buffers, else leave the f16 kernels for a while (8 searches, doubling while every return is rejected again)
static void knn_retune_after_mass_rejection(radad_knn_t h) {
    if (h->hi && h->ntotal > h->plane_decided_rows) h->replan = true;
    else if (h->cap_boost == 1) h->cap_boost = 4;
    else { h->hi_skip = 8 << std::min(h->hi_fail_streak, 6); ++h->hi_fail_streak; }
    h->tuned_at = h->search_seq;
    h->verify_next = true;
}

// the most recent search's device work is complete (its statistics are in the pinned host copy)
static hipError_t knn_wait_last_search(radad_knn_t h) {
    if (!h->have_last) return hipSuccess;
    if (hipStreamSynchronize(h->last_stream) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();                            // (the stream is gone: everything on the device, then)
    return hipDeviceSynchronize();
}

static int knn_search_phase1(radad_knn_t h, const void* q_in, int q_dtype, int64_t nq, int k, int margin, float* lb_out, hipStream_t st,
                             SearchCtx* ctx) {
    // one workspace per handle: a search on another stream waits for the previous one (threads are serialised by h->mu,
    // the device work by this event)
    // (searches share the handle's workspace.  On ONE stream they are ordered anyway; a search on another stream than the last one
    // first records an event behind whatever that stream holds and waits for it -- recording one behind EVERY search, as until round 4,
    // held the kernel after each search back ~6 us, 5 % of an online search)
    if (h->have_last && !h->owned_serial && st != h->last_stream) {
        if (hipEventRecord(h->ev_done, h->last_stream) == hipSuccess) RADAD_HIP_CHECK(hipStreamWaitEvent(st, h->ev_done, 0));
        else { (void)hipGetLastError(); RADAD_HIP_CHECK(hipDeviceSynchronize()); }      // (that stream no longer exists)
    }
    // from here on this search's kernels go to `st` and use the handle's workspace: whatever comes next -- also after an error return
    // or radad_knn_search_abort between the halves -- orders itself behind `st` (until round 4 this was only recorded at the END
    // of phase 2: a search on another stream after an aborted one shared the workspace with kernels still running)
    h->last_stream = st;
    h->have_last = true;

    h->prof.next_search();
    const bool cert = k <= KNN_CERT_MAX_K;      // beyond: legacy k + margin candidates, no certificate
    // how the certificates of earlier searches fared: any report that has arrived and was not looked at yet (see the handle); this
    // search takes over slot search_seq & 1
    // (until round 4 an event behind every search was waited for here: an event record between dependent kernels costs the stream ~6 us,
    // a tenth of an online search.  The exact kernel stamps its report; one that has not arrived yet is simply not looked at.)
    for (int sl = 0; cert && sl < 2; ++sl) {
        volatile int* rep = reinterpret_cast<volatile int*>(h->host_count + 8 * sl);
        const int stamp = rep[6];
        if (stamp == 0 || stamp == h->stamp_seen[sl]) continue;
        std::atomic_thread_fence(std::memory_order_acquire);
        const int rejected = rep[0], rep_nq = rep[7];
        std::atomic_thread_fence(std::memory_order_acquire);
        if (rep[6] != stamp) continue;                    // the device is rewriting the slot right now: next search
        h->stamp_seen[sl] = stamp;
        ++h->reports_consumed;
        // the report's own search: the one search at or below search_seq with that stamp's low 30 bits
        const uint64_t cur30 = h->search_seq & 0x3fffffff, rep30 = (uint64_t)(stamp - 1);
        const uint64_t rep_seq = h->search_seq - ((cur30 - rep30) & 0x3fffffff);
        // (a report from before the last change of the plane or of the buffers says nothing about them)
        if (h->hi_skip == 0 && rep_nq >= 64 && (int64_t)rejected * 4 > rep_nq && rep_seq >= h->tuned_at) knn_retune_after_mass_rejection(h);
    }
    const int cslot = (int)(h->search_seq & 1);
#ifdef RADAD_DEBUG_HOOKS
    if (getenv("RADAD_DEBUG_KNN")) h->hi_skip = 0;     // timing ablations (wrong results, every query rejected): stay on the kernel under test
#endif
    const int l2 = h->metric == RADAD_METRIC_L2 ? 1 : 0;

    // ---- choice of the scan -------------------------------------------------------------------------------------
    int n_qtiles, n_splits;
    int64_t chunk_rows;
    knn_geometry(std::max<int64_t>(h->ntotal, 1), nq, &n_qtiles, &n_splits, &chunk_rows);
    const int wide_min_q = h->opt_wide_min_q;
    bool use_hi = false, skipped_hi = false;
    int s_splits = 0;
    const int ksel = k + margin;                 // list length of the fp32 tile kernels
    // small batches (<= 16 queries: the online predict() search, pipeline.py:1038-1054, is ONE query of dim 5376 / 3584) stream the
    // store; the kernels park only the nq queries handed over in LDS, so any dim goes as long as that block fits beside the slots
    // (16 queries of dim 5376 do not: such a batch takes the tile kernels like a large one)
    const size_t sq_slot_bytes = std::max<size_t>(sizeof(float2) * 4 * SQ_NQ * SQ_SLOTS, sizeof(u64) * 3 * SQ_NQ * 32) + sizeof(int) * 4 * SQ_NQ;
    const size_t sq_lds_hi = sizeof(_Float16) * (size_t)nq * (h->dim + 8) + sq_slot_bytes;
    const size_t sq_lds_f32 = sizeof(float) * (size_t)nq * (h->dim + 4) + sq_slot_bytes;
    const bool sq_fits = nq <= SQ_NQ && ksel <= 32 &&
                         ((!h->hi_off && h->dim % 64 == 0 && sq_lds_hi <= SQ_LDS_BUDGET) || (!h->f16 && h->dim % 32 == 0 && sq_lds_f32 <= SQ_LDS_BUDGET));
    // the certified tile scan: any k the certificate covers (its candidate buffers are sized from k); the floor's rank k + margin
    // must exist in the sample (16 entries per sample tile)
    if (cert && (nq >= wide_min_q || !sq_fits) && h->ntotal > 0 && h->dim % 64 == 0 && !h->hi_off) {
        int wq, ws; int64_t wc;
        knn_geometry_wide(h->ntotal, nq, &wq, &ws, &wc);
        // the sample pre-pass: one tile per workgroup, at most KW_SAMPLE_SPLITS tiles and 1/8 of the store (whatever the number of
        // query tiles: every phase of the scan is sized from the sample, a small sample means more phases)
        // ... and about 3 % of it: the pre-pass multiplies every query tile with its rows, so on a shard of a row-sharded store -- 1/G of
        // the rows against G times the queries -- a fixed 16 384-row sample was 12 % of the scan's own work at G = 8 (0.16 of 1.7 ms);
        // large enough, though, for the floor's rank to exist twice over (16 entries per sample tile)
        {
            const int64_t tiles = h->ntotal / KW_M;
            int64_t want = (tiles * 3 / 100 + 4) / 8 * 8;
            const int64_t need = ((int64_t)(2 * ksel + KW_SAMPLE_LIST - 1) / KW_SAMPLE_LIST + 7) / 8 * 8;
            want = std::max<int64_t>(std::max<int64_t>(want, need), 8);
            // ... but never fewer tiles than fill ONE round of the chip: the pre-pass is one tile per workgroup, its time is that one
            // tile's latency whether 32 or 256 workgroups run it, and a larger sample is a tighter first floor (BASELINE config 2's
            // 100 k rows took 8 tiles = 2048 rows: every tile of the scan's first layer then admitted ~2 rows per query -- 512 returning
            // atomics per workgroup and tile; with 48 tiles the whole store is one launch behind the sample's floor alone)
            want = std::max<int64_t>(want, std::min<int64_t>(KW_SAMPLE_SPLITS, (256 / std::max(1, std::min(wq, 256))) / 8 * 8));
            s_splits = (int)std::min<int64_t>(std::min<int64_t>(KW_SAMPLE_SPLITS, want), tiles / 8 / 8 * 8);
        }
        if (s_splits >= 8 && s_splits * KW_SAMPLE_LIST >= 2 * ksel) {
            if (h->hi_skip > 0) { --h->hi_skip; skipped_hi = true; if (h->hi_skip == 0) h->verify_next = true; }
            else if (knn_ensure_hi(h, st, true)) { use_hi = true; n_qtiles = wq; }
        }
    }
    if (!use_hi && cert && !knn_ensure_hi(h, st, false)) { radad_set_error("store statistics could not be computed"); return RADAD_EHIP; }
    const bool smallq_geom = !use_hi && nq <= SQ_NQ && ksel <= 32 && h->ntotal > 0;
    // the small batch over the f16 plane (certified like the tile scan): stores the plane is kept for, or fp16 stores
    const int smallq_hi_on = h->opt_smallq_hi;
    bool smallq_hi = false;
    if (smallq_geom && cert && smallq_hi_on && h->dim % 64 == 0 && h->ntotal >= 16384 && !h->hi_off && !skipped_hi &&
        sq_lds_hi <= SQ_LDS_BUDGET) {
        // (every search that would take a certified f16 kernel counts the skip down: a handle that only sees small batches after a
        // mass rejection used to stay on the fp32 kernel for ever)
        if (h->hi_skip > 0) --h->hi_skip;
        else smallq_hi = knn_ensure_hi(h, st, true);
    }
    // a small fp32 store (the IVF index's centroids; a database of a few thousand files): every score + select on the staged copy
    const int dense_plen = (int)((std::max<int64_t>(h->ntotal, 1) + 3) / 4 * 4);
    const bool dense = cert && !use_hi && !smallq_hi && !h->f16 && h->opt_dense && h->ntotal >= 1 &&
                       h->ntotal <= (nq <= SQ_NQ ? RF_STAGE_MAX_SMALLQ : RF_STAGE_MAX) && h->dim % 16 == 0 && nq * (int64_t)dense_plen <= ((int64_t)1 << 24);
    const bool smallq = smallq_geom && !dense && !smallq_hi && !h->f16 && h->dim % 32 == 0 && sq_lds_f32 <= SQ_LDS_BUDGET;
    int sq_rows_per_wave = 0;
    // a small store of wide rows (the reference's own: 25 423 x 5376) has too few 16-row steps to occupy the chip with one wave per
    // row slice: the K-split form puts four waves on every step
    const bool sq_ksplit = smallq_hi && h->dim >= 1024 && ceil_div64(h->ntotal, 16) < 4096;
    // (its lists are 16 entries for k <= 16 -- the reference's k = 15 included -- instead of k + 6 <= 32: half the list registers,
    // twice the waves per SIMD to hide the HBM latency behind.  A workgroup whose 16-entry list is used up by rows within the
    // threshold rejects the query; the exact kernel over so small a store costs ~0.3 ms)
    // (the same 16-entry lists on every f16 small-batch scan since round 4: k_knn_hi_smallq<32> streams the 1 M x 512 store in 0.40 ms,
    // <16> in 0.20 -- 155 VGPRs against the 32-entry lists' panel of half the loads in flight)
    const int ksel_sq = (smallq_hi && k <= 16) ? 16 : ksel;
    if (sq_ksplit) {
        int64_t rpg = 16;
        while (ceil_div64(h->ntotal, rpg) * ksel_sq > RF_STAGE_MAX_SMALLQ) rpg += 16;
        sq_rows_per_wave = (int)rpg;                  // (rows per WORKGROUP in this form)
        n_splits = (int)ceil_div64(h->ntotal, rpg);
        n_qtiles = 1;
    } else if (smallq || smallq_hi) {
        // a wave streams >= 128 KB (128 rows at dim 512) so that its lists' hand-over stays small beside the stream, but no more
        // rows than leave 8 waves for every CU; the lists of a query (one per workgroup) should fit the re-rank's staged form
        // one full round of resident waves: 1024 SIMDs x the waves per SIMD the kernel's registers allow (k_knn_hi_smallq<16>: 155
        // VGPRs = 3; the 32-entry and fp32 variants: 2).  (Measured: no difference against 2048 waves on the 1 M x 512 store -- 0.2075
        // vs 0.208 ms, 4.93 TB/s either way: the stream is not limited by the number of waves in flight.)
        const int64_t waves_wanted = 1024 * ((smallq_hi && ksel_sq <= 16) ? 3 : 2);
        const size_t rb = smallq_hi ? (size_t)h->dim * 2 : (size_t)h->dim * 4;
        const int64_t rows_min = std::max<int64_t>(16, std::min<int64_t>(128, ceil_div64(ceil_div64(128 * 1024, (int64_t)rb), 16) * 16));
        int64_t rpw = std::max<int64_t>(ceil_div64(ceil_div64(h->ntotal, waves_wanted), 16) * 16, rows_min);
        while (rpw < 128 && ceil_div64(ceil_div64(h->ntotal, rpw), 4) * ksel_sq > RF_STAGE_MAX_SMALLQ) rpw += 16;
        sq_rows_per_wave = (int)rpw;
        n_splits = (int)ceil_div64(ceil_div64(h->ntotal, rpw), 4);                // workgroups of 4 waves = lists per query
        n_qtiles = 1;
    }
    const bool f16_tile = !use_hi && !smallq_hi && h->f16 && ksel <= 32 && h->dim % 64 == 0;
    // the tile scan's phases: a floor taken from m rows admits ~(rank / m) of what it is applied to, so every launch covers at most
    // 8 x the rows its floor was taken from -- the first 8 x the sample with the sample's floor, the next 8 x that with the floor the
    // candidates so far give (k_kth_floor), and so on: 2 launches up to 1.2 M rows, 3 up to 9.5 M.  Each admits ~8 (k + margin) rows
    // per query; the candidate buffer holds 32 (k + margin) (>= 1024): more (stores of near-duplicates) rejects the query.
    const int64_t hi_phase0 = (int64_t)8 * s_splits * KW_M;
    const int emit_cap = use_hi ? std::min(RF_STAGE_MAX, std::max(1024 * h->cap_boost, 32 * ksel)) : 0;
    h->last_qtiles = n_qtiles;
    if (dense) { n_qtiles = (int)ceil_div64(nq, 16); n_splits = 1; }
    h->last_threads = use_hi ? KW_THREADS : ((smallq || smallq_hi || dense) ? SQ_THREADS : KNN_THREADS);
    h->last_kind = use_hi ? RADAD_SCAN_HI_TILE : dense ? RADAD_SCAN_F32_DENSE : smallq_hi ? RADAD_SCAN_HI_SMALLQ : smallq ? RADAD_SCAN_F32_SMALLQ
                          : f16_tile ? RADAD_SCAN_F16_TILE : RADAD_SCAN_F32_TILE;
    const int plen = use_hi ? emit_cap : dense ? dense_plen : ksel_sq;   // entries of a partial list / of the candidate buffer
    const int n_parts = (use_hi || dense) ? 1 : n_splits;
    // (dense: eps of exact fp32 products is ~1e-6 of |q||y| -- hardly a row beyond the k best is within 2 eps; k + 32 candidates keep
    // the re-rank's workgroup at 34 KB of LDS for 4096 staged scores, four per CU instead of three: the IVF coarse step's 1024
    // workgroups in one round.  More near-ties than that reject the query: exact kernel.)
    // (a handle that has widened its candidate buffers -- cap_boost: a store whose rows crowd within 2 eps of the k-th best -- also
    // re-ranks four times as many: the fp32 funnel in front of the float64 re-score takes them at ~2 KB of row reads each)
    const int cap = cert ? (dense ? k + KNN_CERT_EXTRA : std::max(k + KNN_CERT_EXTRA, KNN_CERT_CAP * (use_hi ? h->cap_boost : 1))) : ksel;
    const int xgroup = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)(64 * 1024) / ((size_t)h->dim * 4)));

    // ---- workspace: qf (decoded bf16) | qn (normalised) | qh (f16 queries) | qscale | qconst | eps | thr_init | cand_cnt |
    //                 flag_count, flag_sel | part_score | part_idx | exact partial keys | ids
    const size_t qrow_f32 = al256((size_t)nq * h->dim * sizeof(float));
    const size_t b_qf = q_dtype == RADAD_Q_BF16 ? qrow_f32 : 0;
    const size_t b_qn = h->metric == RADAD_METRIC_COSINE ? qrow_f32 : 0;
    const size_t b_qh = (use_hi || f16_tile || smallq_hi) ? al256((size_t)nq * h->dim * 2) : 0;
    const size_t b_vec = al256((size_t)nq * sizeof(float));
    // (the sample pre-pass's lists live BEHIND the candidate buffers, not in them: k_floor_from_sample clears a query's buffer to -inf
    // while other queries' sample lists are still being read)
    const size_t cand_elems = (size_t)nq * (size_t)n_parts * plen;
    const size_t part_elems = cand_elems + (use_hi ? (size_t)nq * KW_SAMPLE_SPLITS * KW_SAMPLE_LIST : 0);
    const size_t b_part = al256(part_elems * sizeof(float));
    const size_t b_xk = cert ? al256((size_t)nq * KX_SLICES * k * sizeof(double)) : 0;
    const size_t b_xi = cert ? al256((size_t)nq * KX_SLICES * k * sizeof(int)) : 0;
    size_t off = 0;
    const size_t o_qf = off; off += b_qf;
    const size_t o_qn = off; off += b_qn;
    const size_t o_qh = off; off += b_qh;
    const size_t o_qscale = off; off += b_vec;
    const size_t o_qconst = off; off += b_vec;
    const size_t o_eps = off; off += b_vec;
    const size_t o_thr = off; off += b_vec;
    const size_t o_ak = off; off += b_vec;             // a_k of the candidates (two-half searches: k_kth_floor -> k_merge_refine)
    const size_t o_cnt = off; off += b_vec;            // cand_cnt [nq] int (zeroed by k_hi_rows with the counters)
    const size_t o_fcount = off; off += 256 + KNN_PROG_MAX * sizeof(int);   // flag_count + statistics | the tile scan's progress counters (one per query tile)
    const size_t o_fsel = off; off += b_vec;
    const size_t o_ps = off; off += b_part;
    const size_t o_pi = off; off += b_part;
    const size_t o_xk = off; off += b_xk;
    const size_t o_xi = off; off += b_xi;
    if (off > h->ws_bytes) {
        RADAD_HIP_CHECK(hipDeviceSynchronize());
        int rc = knn_workspace(h, off);
        if (rc) return rc;
    }
    char* ws = (char*)h->ws;
    _Float16* qh = (_Float16*)(ws + o_qh);
    float* qscale = (float*)(ws + o_qscale);
    float* qconst = (float*)(ws + o_qconst);
    float* eps = (float*)(ws + o_eps);
    float* thr_init = (float*)(ws + o_thr);
    float* qconst_ak = (float*)(ws + o_ak);
    int* cand_cnt = (int*)(ws + o_cnt);
    int* flag_count = (int*)(ws + o_fcount);
    float* ps = (float*)(ws + o_ps);
    int* pi = (int*)(ws + o_pi);

    // ---- queries: decode, normalise, round (+ scale) to f16, error bound ------------------------------------------
    const float* q_use = (const float*)q_in;
    const unsigned rgrid = (unsigned)ceil_div64(nq, 4);
    if (q_dtype == RADAD_Q_BF16) {
        float* qf = (float*)(ws + o_qf);
        const int64_t ne = nq * h->dim;
        hipLaunchKernelGGL(k_bf16_to_f32, dim3((unsigned)ceil_div64(ne, 1024)), dim3(256), 0, st, (const unsigned short*)q_in, qf, ne);
        q_use = qf;
    }
    const bool hi_q = use_hi || smallq_hi;                 // f16 queries with a per-query scale
    const float* mu = (hi_q && !h->f16) ? h->cmu : nullptr; // centred plane: the queries are centred the same way
    const bool biased = hi_q && (l2 || mu);                // the scale + bias variant of the f16 kernels (RSC 2)
    const bool prep = cert || use_hi || f16_tile;          // k_hi_rows runs: it also normalises and clears the flags
    if (h->metric == RADAD_METRIC_COSINE) {
        float* qn = (float*)(ws + o_qn);
        if (!prep) hipLaunchKernelGGL(k_rows_prepare<float>, dim3(rgrid), dim3(256), 0, st, q_use, qn, (float*)nullptr, nq, h->dim, 2);
        // (with prep the same normalisation happens inside k_hi_rows, which reads the raw queries)
    }
    // (|q|^2 is not needed: ranking uses 2 q.y - |y|^2 and the reported distance is re-scored exactly)
    if (prep) {
        HiRowsParams hp;
        hp.in = q_use; hp.in_f16 = 0;
        hp.hi = (hi_q || f16_tile) ? qh : nullptr;
        hp.scale_out = hi_q ? qscale : nullptr;
        hp.stat_max = nullptr; hp.eps_out = cert ? eps : nullptr; hp.ystat = h->stat;
        hp.n = nq; hp.dim = h->dim;
        hp.fixed_e = hi_q ? HI_E_PER_ROW : 0;            // the fp16 tile kernel multiplies un-scaled fp16 queries
        hp.l2 = l2; hp.exact_ops = (hi_q || f16_tile) ? 0 : 1;
        hp.norm_out = h->metric == RADAD_METRIC_COSINE ? (float*)(ws + o_qn) : nullptr;
        hp.zero_flags = cert ? cand_cnt : nullptr; hp.zero_counters = cert ? flag_count : nullptr;
        hp.mu = mu; hp.mu_norm = mu ? h->mu_norm : 0.f; hp.mu_sq = mu ? h->mu_sq : 0.f; hp.biased = (biased || (f16_tile && l2)) ? 1 : 0;
        hp.bias_out = nullptr; hp.qconst_out = biased ? qconst : nullptr;
        launch_hi_rows(hp, st);
    }
    if (h->metric == RADAD_METRIC_COSINE) q_use = (const float*)(ws + o_qn);
    RADAD_HIP_CHECK(hipGetLastError());

    // ---- scan ----------------------------------------------------------------------------------------------------
    KnnParams p;
    p.db = h->rows; p.db_f16 = h->f16; p.ynorm = h->ynorm; p.q = f16_tile ? (const void*)qh : (const void*)q_use; p.n = h->ntotal; p.nq = (int)nq; p.dim = h->dim; p.k = ksel;
    p.l2 = l2;
    p.n_qtiles = n_qtiles; p.n_splits = n_splits; p.chunk_rows = chunk_rows; p.part_score = ps; p.part_idx = pi;
    p.debug = 0;
    const dim3 grid((unsigned)(n_qtiles * n_splits));
    if (use_hi) {
        KnnHiParams wp;
        wp.db = h->f16 ? (const void*)h->rows : (const void*)h->hi;
        wp.rscale = h->f16 ? nullptr : h->rscale;
        wp.uscale = h->f16 ? 1.0f : (h->uniform_e != HI_E_PER_ROW ? ldexpf(1.0f, -h->uniform_e) : (h->metric == RADAD_METRIC_COSINE ? 0x1p-14f : 1.0f));
        // bias of the scale + bias variant: a centred plane has its own (|y - mu|^2 or mu.y); un-centred L2 uses |y|^2 as stored
        wp.rbias = mu ? h->rbias : h->ynorm; wp.bias_sign = l2 ? -1.f : 1.f; wp.mult = l2 ? 2.f : 1.f; wp.qconst = biased ? qconst : nullptr;
        wp.q = qh; wp.qscale = qscale;
        wp.n = h->ntotal; wp.nq = (int)nq; wp.row_bytes = h->dim * 2; wp.l2 = l2; wp.id_off = 0;
        wp.n_qtiles = n_qtiles; wp.part_score = ps; wp.part_idx = pi;
        wp.cand_cap = emit_cap; wp.cand_cnt = cand_cnt;
        wp.thr_init = nullptr;
        wp.ksplit = 1; wp.kacc = nullptr; wp.kflag = nullptr; wp.loose_floor = 0; wp.chunk_stride = 0;
        wp.floor_live = nullptr; wp.eps = nullptr; wp.k_sel = 0; wp.prog = nullptr; wp.prog_thr[0] = wp.prog_thr[1] = wp.prog_thr[2] = wp.prog_thr[3] = 0;
        wp.debug = 0; wp.stamps = nullptr;
#ifdef RADAD_DEBUG_HOOKS        // timing experiments only (make exp); never in the shipped library
        { const char* dbg = getenv("RADAD_DEBUG_KNN"); wp.debug = dbg ? atoi(dbg) : 0; }
        {   // RADAD_KNN_STAMPS=<file>: in-kernel s_memtime stamps of the 8 waves of workgroup 0, dumped after the scan (synchronises)
            static unsigned long long* stamp_buf = nullptr;
            if (getenv("RADAD_KNN_STAMPS")) {
                if (!stamp_buf) (void)hipMalloc(&stamp_buf, 8 * 4096 * 8);
                (void)hipMemsetAsync(stamp_buf, 0, 8 * 4096 * 8, st);
                wp.stamps = stamp_buf;
            }
        }
#endif
        const int rsc = biased ? (wp.rscale ? 2 : 3) : (wp.rscale ? 1 : 0);
        const size_t lds = knn_hi_lds_bytes();
        const void* fns[4] = {reinterpret_cast<const void*>(k_knn_hi<0>), reinterpret_cast<const void*>(k_knn_hi<1>),
                              reinterpret_cast<const void*>(k_knn_hi<2>), reinterpret_cast<const void*>(k_knn_hi<3>)};
        const void* sfns[4] = {reinterpret_cast<const void*>(k_knn_hi_sample<0>), reinterpret_cast<const void*>(k_knn_hi_sample<1>),
                               reinterpret_cast<const void*>(k_knn_hi_sample<2>), reinterpret_cast<const void*>(k_knn_hi_sample<3>)};
        RADAD_HIP_CHECK(hipFuncSetAttribute(fns[rsc], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        RADAD_HIP_CHECK(hipFuncSetAttribute(sfns[rsc], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // ONE launch over the whole store with the floors raised inside it (knn_hi.inc: floor_live), when the candidate buffers are the
        // 1024 entries a wave holds in registers and a workgroup's chunk is long enough to refresh at least once before the sample's
        // floor alone would fill them: after t tiles of every resident chunk stream ~t x 256 x (streams per query tile) rows have been
        // scanned, and the phases of the launch-per-phase form (8 x the sample, 8 x that, ...) become refresh points t, 8 t, 64 t.
        // Round 4 ran one launch per phase (2 up to 1.2 M rows, 3 up to 9.5 M, k_kth_floor between them): every launch has its ramp
        // and its tail, and a 100 k-row store (BASELINE config 2) ran 64 + 327 tiles as 1 + 3 tile times on 512 workgroups.
        int live_nr = 0, live_at[4] = {0, 0, 0, 0};
        const bool one_go = (int64_t)ksel * h->ntotal <= (int64_t)(emit_cap / 3) * s_splits * KW_M;
        if (!one_go && emit_cap <= 1024 && h->opt_live_floor == 1 && h->dim >= 128 && n_qtiles <= KNN_PROG_MAX) {
            // refresh when a query tile has started 8 x the sample's tiles, 8 x that, ... (the phases of the launch-per-phase form), as
            // long as a fifth of the store is still to come
            const int64_t tiles = ceil_div64(h->ntotal, KW_M);
            for (int64_t t = hi_phase0 / KW_M; t + KW_REFRESH_WGS <= tiles - tiles / 5 && live_nr < 4; t *= 8) live_at[live_nr++] = (int)t;
        }
        // sample pre-pass: the (k + margin)-th best score over the first rows of the store is a score at least k rows reach, so the
        // scan admits from it (minus 2 eps) instead of -inf (the score of a (row, query) pair does not depend on the tiling).
        // Its lists live behind the candidate buffers.
        {
            KnnHiParams sp = wp;
            // The sample's tiles are SPREAD over the store (every (tiles / s_splits)-th one).  The first s_splits tiles, as until round 4,
            // are a sample of the store only while its rows are in random order: on a store whose rows arrive cluster by cluster
            // (add_vectors_batch appends 10 000 rows at a time, vector_database.py:134-138) a query of the LAST cluster got a floor
            // from rows of the first -- so loose that every row of its own cluster passed it, the candidate buffer overflowed and the
            // certificate rejected all 256 queries of a batch (tests/test_gpu_reference_shapes.py::test_a_store_that_drifts...).
            sp.n = h->ntotal; sp.n_splits = s_splits; sp.chunk_rows = KW_M;
            sp.part_score = ps + cand_elems; sp.part_idx = pi + cand_elems;
            sp.chunk_stride = std::max<int64_t>(1, (h->ntotal / KW_M) / s_splits) * KW_M;
            const int sq_grid = sp.n_qtiles <= 8 ? sp.n_qtiles : (sp.n_qtiles + 7) / 8 * 8;
            // (the K split's scratch is indexed by row0 / KW_M: the sample's tiles lie anywhere in the store)
            sp.ksplit = knn_tile_ksplit(h, sq_grid, sp.n_splits, KW_M, h->ntotal, st);
            sp.kacc = h->kacc; sp.kflag = h->kflag;
            const dim3 sg((unsigned)(sq_grid * sp.n_splits * sp.ksplit)), sb(KW_THREADS);
            if (rsc == 0) hipLaunchKernelGGL(k_knn_hi_sample<0>, sg, sb, lds, st, sp);
            else if (rsc == 1) hipLaunchKernelGGL(k_knn_hi_sample<1>, sg, sb, lds, st, sp);
            else if (rsc == 2) hipLaunchKernelGGL(k_knn_hi_sample<2>, sg, sb, lds, st, sp);
            else hipLaunchKernelGGL(k_knn_hi_sample<3>, sg, sb, lds, st, sp);
            const int n_ent = sp.n_splits * KW_SAMPLE_LIST;
            hipLaunchKernelGGL(k_floor_from_sample, dim3((unsigned)nq), dim3(RF_THREADS), (size_t)n_ent * 4 + 1040, st, (const float*)sp.part_score,
                               (const int*)sp.part_idx, n_ent, k + margin, (const float*)eps, thr_init, live_nr > 0 ? ps : (float*)nullptr, emit_cap,
                               live_nr > 0 ? flag_count + KNN_PROG_OFF : (int*)nullptr, n_qtiles);
            wp.thr_init = thr_init;
        }
        const dim3 b(KW_THREADS);
        auto launch_range = [&](int64_t r0, int64_t r1) {      // rows [r0, r1) of the store, one full round (or two) of workgroups
            KnnHiParams rp = wp;
            const size_t rb = (size_t)wp.row_bytes;
            rp.db = (const char*)wp.db + (size_t)r0 * rb;
            rp.rscale = wp.rscale ? wp.rscale + r0 : nullptr;
            rp.rbias = wp.rbias ? wp.rbias + r0 : nullptr;
            rp.n = r1 - r0; rp.id_off = r0;
            rp.loose_floor = (r0 == 0 && r1 == h->ntotal && s_splits < KW_SAMPLE_SPLITS && live_nr == 0) ? 1 : 0;      // one launch behind a small sample
            // (the counting form for the FIRST phase of a large store -- its floor is the sample's, ~64 candidates per tile -- was measured
            // too: 0.883-0.887 against 0.874-0.877 ms per search, three alternations on one box, profiles/r5_ab_counting_emit_phase0.txt: no)
            if (live_nr > 0) {
                rp.floor_live = thr_init; rp.eps = eps; rp.k_sel = k;
                for (int i = 0; i < 4; ++i) rp.prog_thr[i] = live_at[i];
                rp.prog = flag_count + KNN_PROG_OFF;
            }
            int gq, gs; int64_t gc;
            knn_geometry_wide(rp.n, nq, &gq, &gs, &gc);
            rp.n_splits = gs; rp.chunk_rows = gc; rp.chunk_stride = gc;
            h->last_splits = gs;
            const int gq_grid = gq <= 8 ? gq : (gq + 7) / 8 * 8;                      // (more than 8 query tiles: whole groups of 8, see the kernel)
            rp.ksplit = knn_tile_ksplit(h, gq_grid, gs, gc, rp.n, st);
            rp.kacc = h->kacc; rp.kflag = h->kflag;
            const dim3 g2((unsigned)(gq_grid * gs * rp.ksplit));
            h->prof.begin(st);      // the event pair brackets a scan launch only (the kernel the roofline is quoted on; the phases of one
                                    // search are two entries)
            if (rsc == 0) hipLaunchKernelGGL(k_knn_hi<0>, g2, b, lds, st, rp);
            else if (rsc == 1) hipLaunchKernelGGL(k_knn_hi<1>, g2, b, lds, st, rp);
            else if (rsc == 2) hipLaunchKernelGGL(k_knn_hi<2>, g2, b, lds, st, rp);
            else hipLaunchKernelGGL(k_knn_hi<3>, g2, b, lds, st, rp);
            h->prof.end(st);
        };
        h->last_scan_launches = 0;
        h->last_scan_phases = 0;
        if (live_nr > 0) {
            launch_range(0, h->ntotal);
            h->last_scan_launches = 1;
            h->last_scan_phases = 1 + live_nr;
        }
        for (int64_t r0 = 0, span = hi_phase0; live_nr == 0 && r0 < h->ntotal; span *= 8) {
            // (a last phase of less than a quarter of its predecessor is not worth a launch of its own: it joins it)
            int64_t r1 = std::min<int64_t>(h->ntotal, r0 + span);
            if (h->ntotal - r1 < span / 4) r1 = h->ntotal;
            // a store the sample's floor alone filters to a third of the buffer (~(k + margin) N / sample rows) is scanned in one go
            if (r0 == 0 && (int64_t)ksel * h->ntotal <= (int64_t)(emit_cap / 3) * s_splits * KW_M) r1 = h->ntotal;
            // A launch deals whole tiles to its row splits, ceil(tiles / splits) each: what decides its time is that quotient, and a
            // remainder of a few tiles costs a whole extra tile per workgroup (64 + 260 tiles over 128 splits = 1 + 3 tile times, the last
            // round of the second launch nearly empty; 68 + 256 tiles = 1 + 2).  When the LAST launch follows this one, up to a quarter
            // more tiles move into this one if that lowers the sum of the two quotients.  (BASELINE config 2 -- 100 k rows = 64 + 327
            // tiles -- gains nothing from it: 1 + 3 either way; its scan stays at 0.27 of the MFMA peak, 1 564 tile tasks over 256 CUs.)
            if (r1 < h->ntotal && (h->ntotal - r1 <= span * 8 || h->ntotal - r1 - span * 8 < span * 2)) {
                auto tile_time = [&](int64_t rows) {
                    int gq, gs; int64_t gc;
                    knn_geometry_wide(rows, nq, &gq, &gs, &gc);
                    return gc / KW_M;
                };
                const int64_t t_this = ceil_div64(r1 - r0, KW_M), t_rest = ceil_div64(h->ntotal - r1, KW_M);
                int64_t best = tile_time(r1 - r0) + tile_time(h->ntotal - r1), best_s = 0;
                for (int64_t sft = 1; sft <= std::min<int64_t>(t_this / 4, t_rest - 1); ++sft) {
                    const int64_t c = tile_time(r1 - r0 + sft * KW_M) + tile_time(h->ntotal - r1 - sft * KW_M);
                    if (c < best) { best = c; best_s = sft; }
                }
                r1 += best_s * KW_M;
            }
            if (r0 > 0) {
                KthParams kp;
                kp.score = ps; kp.cnt = cand_cnt; kp.cap = emit_cap; kp.k = k; kp.eps = eps; kp.floor_io = thr_init; kp.lb_out = nullptr; kp.ak_out = nullptr;
                hipLaunchKernelGGL(k_kth_floor, dim3((unsigned)nq), dim3(RF_THREADS), (size_t)emit_cap * 4 + 1040, st, kp);
            }
            launch_range(r0, r1);
            ++h->last_scan_launches;
            ++h->last_scan_phases;
            r0 = r1;
        }
#ifdef RADAD_DEBUG_HOOKS
        if (wp.stamps) {
            std::vector<unsigned long long> hs(8 * 4096);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(hs.data(), wp.stamps, hs.size() * 8, hipMemcpyDeviceToHost);
            if (FILE* f = fopen(getenv("RADAD_KNN_STAMPS"), "wb")) { fwrite(hs.data(), 8, hs.size(), f); fclose(f); }
        }
#endif
    } else if (dense) {
        DenseParams dp;
        dp.db = (const float*)h->rows; dp.ynorm = h->ynorm; dp.q = q_use; dp.n = h->ntotal; dp.nq = (int)nq; dp.dim = h->dim; dp.l2 = l2;
        dp.plen = dense_plen; dp.score = ps; dp.idx = pi; dp.cnt = cand_cnt;
        h->prof.begin(st);
        if (nq <= SQ_NQ) hipLaunchKernelGGL((k_knn_dense<1, 1>), dim3((unsigned)ceil_div64(dense_plen, 64), (unsigned)ceil_div64(nq, 16)), dim3(256), 0, st, dp);
        else hipLaunchKernelGGL((k_knn_dense<2, 4>), dim3((unsigned)ceil_div64(dense_plen, 128), (unsigned)ceil_div64(nq, 64)), dim3(256), 0, st, dp);
        h->prof.end(st);
    } else if (smallq_hi) {
        SmallQHiParams sp;
        sp.db = h->f16 ? (const _Float16*)h->rows : h->hi;
        sp.rscale = h->f16 ? nullptr : h->rscale;
        sp.uscale = h->f16 ? 1.0f : (h->uniform_e != HI_E_PER_ROW ? ldexpf(1.0f, -h->uniform_e) : (h->metric == RADAD_METRIC_COSINE ? 0x1p-14f : 1.0f));
        sp.rbias = biased ? (mu ? h->rbias : h->ynorm) : nullptr; sp.bias_sign = l2 ? -1.f : 1.f; sp.mult = l2 ? 2.f : 1.f;
        sp.qconst = biased ? qconst : nullptr;
        sp.q = qh; sp.qscale = qscale; sp.n = h->ntotal; sp.nq = (int)nq; sp.dim = h->dim; sp.k = ksel_sq;
        sp.rows_per_wave = sq_rows_per_wave; sp.n_parts = n_splits; sp.part_score = ps; sp.part_idx = pi;
        const size_t lds = sq_lds_hi;
        const dim3 sgrid((unsigned)n_splits);
        if (sq_ksplit) {
            const void* fn = ksel_sq <= 16 ? reinterpret_cast<const void*>(k_knn_hi_smallq_ksplit<16>) : reinterpret_cast<const void*>(k_knn_hi_smallq_ksplit<32>);
            RADAD_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            h->prof.begin(st);
            if (ksel_sq <= 16) hipLaunchKernelGGL(k_knn_hi_smallq_ksplit<16>, sgrid, dim3(SQ_THREADS), lds, st, sp);
            else hipLaunchKernelGGL(k_knn_hi_smallq_ksplit<32>, sgrid, dim3(SQ_THREADS), lds, st, sp);
            h->prof.end(st);
        } else if (ksel_sq <= 16) {
            RADAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_hi_smallq<16>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            h->prof.begin(st);
            hipLaunchKernelGGL(k_knn_hi_smallq<16>, sgrid, dim3(SQ_THREADS), lds, st, sp);
            h->prof.end(st);
        } else {
            RADAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_hi_smallq<32>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            h->prof.begin(st);
            hipLaunchKernelGGL(k_knn_hi_smallq<32>, sgrid, dim3(SQ_THREADS), lds, st, sp);
            h->prof.end(st);
        }
    } else if (smallq) {
        SmallQParams sp;
        sp.db = (const float*)h->rows; sp.ynorm = h->ynorm; sp.q = q_use; sp.n = h->ntotal; sp.nq = (int)nq; sp.dim = h->dim; sp.k = ksel;
        sp.l2 = p.l2; sp.rows_per_wave = sq_rows_per_wave; sp.n_parts = n_splits; sp.part_score = ps; sp.part_idx = pi;
        const size_t lds = sq_lds_f32;
        const dim3 sgrid((unsigned)n_splits);
        if (ksel <= 16) {
            RADAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_f32_smallq<16>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            h->prof.begin(st);
            hipLaunchKernelGGL(k_knn_f32_smallq<16>, sgrid, dim3(SQ_THREADS), lds, st, sp);
            h->prof.end(st);
        } else {
            RADAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_f32_smallq<32>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            h->prof.begin(st);
            hipLaunchKernelGGL(k_knn_f32_smallq<32>, sgrid, dim3(SQ_THREADS), lds, st, sp);
            h->prof.end(st);
        }
    } else if (f16_tile || (!h->f16 && ksel <= 32 && h->dim % KT_K == 0)) {
        // tile kernel with LDS-DMA staging and register-resident lists: <list entries, fp16 operands>
        const void* fn = f16_tile ? (ksel <= 16 ? reinterpret_cast<const void*>(k_knn_f32_reg<16, true>)
                                                : reinterpret_cast<const void*>(k_knn_f32_reg<32, true>))
                                  : (ksel <= 16 ? reinterpret_cast<const void*>(k_knn_f32_reg<16, false>)
                                                : reinterpret_cast<const void*>(k_knn_f32_reg<32, false>));
        RADAD_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)knn_reg_lds_bytes()));
        h->prof.begin(st);
        if (f16_tile && ksel <= 16) hipLaunchKernelGGL((k_knn_f32_reg<16, true>), grid, dim3(KNN_THREADS), knn_reg_lds_bytes(), st, p);
        else if (f16_tile) hipLaunchKernelGGL((k_knn_f32_reg<32, true>), grid, dim3(KNN_THREADS), knn_reg_lds_bytes(), st, p);
        else if (ksel <= 16) hipLaunchKernelGGL((k_knn_f32_reg<16, false>), grid, dim3(KNN_THREADS), knn_reg_lds_bytes(), st, p);
        else hipLaunchKernelGGL((k_knn_f32_reg<32, false>), grid, dim3(KNN_THREADS), knn_reg_lds_bytes(), st, p);
        h->prof.end(st);
    } else {
        // generic kernel (any k, any dim % 4 == 0): register staging, lists in the partial arrays; an fp16 store is
        // decoded to fp32 while staging and multiplied with the fp32 queries
        const void* fn = h->f16 ? reinterpret_cast<const void*>(k_knn_f32<true>) : reinterpret_cast<const void*>(k_knn_f32<false>);
        RADAD_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)knn_lds_bytes()));
        h->prof.begin(st);
        if (h->f16) hipLaunchKernelGGL(k_knn_f32<true>, grid, dim3(KNN_THREADS), knn_lds_bytes(), st, p);
        else hipLaunchKernelGGL(k_knn_f32<false>, grid, dim3(KNN_THREADS), knn_lds_bytes(), st, p);
        h->prof.end(st);
    }
    if (!use_hi) { h->last_splits = n_splits; h->last_scan_launches = 1; }
    RADAD_HIP_CHECK(hipGetLastError());

    // ---- sharded search: lower bounds of the exact scores of this store's k best rows, per query -------------------------------
    // (scan score - eps of the k rows with the best scan scores).  The scan score of the f16 tile kernel estimates q.y (inner
    // product, cosine) or -|q - y|^2 (L2) whatever the plane's centring, so the bounds of different shards compare; the other
    // kernels report none (-inf).
    if (lb_out) {
        if (use_hi && cert) {
            KthParams kp;
            kp.score = ps; kp.cnt = cand_cnt; kp.cap = emit_cap; kp.k = k; kp.eps = eps; kp.floor_io = nullptr; kp.lb_out = lb_out;
            kp.ak_out = qconst_ak;
            hipLaunchKernelGGL(k_kth_floor, dim3((unsigned)nq), dim3(RF_THREADS), (size_t)emit_cap * 4 + 1040, st, kp);
        } else {
            hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)ceil_div64(nq * k, 256)), dim3(256), 0, st, lb_out, nq * k, -INFINITY);
        }
        RADAD_HIP_CHECK(hipGetLastError());
    }

    ctx->valid = true;
    ctx->nq = nq; ctx->k = k; ctx->l2 = l2; ctx->cslot = cslot; ctx->n_parts = n_parts; ctx->plen = plen; ctx->cap = cap; ctx->xgroup = xgroup;
    ctx->cert = cert; ctx->emit = use_hi || dense; ctx->use_floor = use_hi; ctx->hi_tile = use_hi; ctx->q_use = q_use;
    ctx->small_lists = !use_hi && !dense && (smallq || smallq_hi);
    ctx->canonical = use_hi || smallq_hi || !l2;
    ctx->have_ak = lb_out != nullptr && use_hi && cert; ctx->o_ak = o_ak;
    h->last_o_cnt = o_cnt; h->last_o_thr = o_thr; h->last_emit_nq = use_hi ? nq : 0;
    ctx->o_eps = o_eps; ctx->o_thr = o_thr; ctx->o_cnt = o_cnt; ctx->o_fcount = o_fcount; ctx->o_fsel = o_fsel; ctx->o_ps = o_ps; ctx->o_pi = o_pi;
    ctx->o_xk = o_xk; ctx->o_xi = o_xi;
    return RADAD_OK;
}

constexpr int RADAD_RETRY_INTERNAL = 1;      // knn_search_phase2 -> knn_search_core: retuned after a mass rejection, run the search again
static int knn_search_phase2(radad_knn_t h, const SearchCtx& c, const float* global_lb, bool may_retry, float* out_dist_dev, int64_t* out_idx_dev,
                             double* out_key_dev, hipStream_t st) {
    char* ws = (char*)h->ws;
    int* flag_count = (int*)(ws + c.o_fcount);
    int* flag_sel = (int*)(ws + c.o_fsel);
    const int64_t nq = c.nq;
    const int k = c.k;
    // ---- float64 re-rank + certificate ----------------------------------------------------------------------------
    RefineParams m;
    m.score = (const float*)(ws + c.o_ps); m.idx = (const int*)(ws + c.o_pi); m.n_parts = c.n_parts; m.k = k; m.dim = h->dim; m.l2 = c.l2; m.nq = nq;
    m.part_len = c.plen; m.cap = c.cap;
    m.eps = c.cert ? (const float*)(ws + c.o_eps) : nullptr; m.thr_init = c.use_floor ? (const float*)(ws + c.o_thr) : nullptr; m.qflag = nullptr;
    m.part_cnt = c.emit ? (const int*)(ws + c.o_cnt) : nullptr; m.global_lb = (c.cert && c.canonical) ? global_lb : nullptr;
    m.ak_in = c.have_ak ? (const float*)(ws + c.o_ak) : nullptr;
    m.flag_count = flag_count; m.flag_sel = flag_sel;
    m.db = h->rows; m.db_f16 = h->f16; m.q = c.q_use; m.id_map = nullptr; m.id_base = h->id_base; m.out_dist = out_dist_dev; m.out_idx = out_idx_dev;
    m.out_key = out_key_dev;
    m.debug = 0;
#ifdef RADAD_DEBUG_HOOKS
    { const char* dbg = getenv("RADAD_DEBUG_KNN"); m.debug = dbg ? atoi(dbg) : 0; }
#endif
    m.stats = c.cert ? flag_count + 1 : nullptr;
    RADAD_REQUIRE(c.n_parts <= RF_THREADS * RF_MAXL, "radad_knn_search: %d partial lists per query exceed the re-rank kernel's %d", c.n_parts,
                  RF_THREADS * RF_MAXL);
    if (c.cert && c.small_lists && nq <= SQ_NQ && m.n_parts <= RS_MAX_PARTS && m.cap <= KNN_CERT_CAP + KNN_CERT_MAX_K) {
        // a small batch: RS_SPLIT workgroups per query (k_refine_small)
        if (!h->rs_key) {                                // (once per handle: allocation and clearing synchronise)
            const size_t kb = (size_t)SQ_NQ * (KNN_CERT_CAP + KNN_CERT_MAX_K) * sizeof(double);
            if (hipMalloc((void**)&h->rs_key, kb) != hipSuccess || hipMalloc((void**)&h->rs_count, SQ_NQ * sizeof(int)) != hipSuccess ||
                hipMemset(h->rs_count, 0, SQ_NQ * sizeof(int)) != hipSuccess) {
                (void)hipGetLastError();
                if (h->rs_key) (void)hipFree(h->rs_key);
                if (h->rs_count) (void)hipFree(h->rs_count);
                h->rs_key = nullptr; h->rs_count = nullptr;
                radad_set_error("hipMalloc of the small-batch re-rank's buffers failed");
                return RADAD_ENOMEM;
            }
        }
        const size_t rlds = refine_small_lds_bytes(c.cap);
        hipLaunchKernelGGL(k_refine_small, dim3(RS_SPLIT, (unsigned)nq), dim3(RS_THREADS), rlds, st, m, h->rs_key, h->rs_count);
    } else {   // certified mode with lists that fit the LDS: selection on a staged copy (one round of loads instead of a pointer chase)
        const size_t entries = (size_t)m.n_parts * m.part_len;
        RADAD_REQUIRE(!c.emit || (m.eps && entries <= (size_t)(nq <= SQ_NQ ? RF_STAGE_MAX_SMALLQ : RF_STAGE_MAX)),
                      "radad_knn_search: candidate buffer larger than the re-rank stages");
        if (m.eps && entries <= (size_t)(nq <= SQ_NQ ? RF_STAGE_MAX_SMALLQ : RF_STAGE_MAX)) {
            const size_t rlds = refine_lds_bytes(c.cap) + entries * 8 + 1024;
            if (rlds > 48 * 1024)
                RADAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge_refine<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
            hipLaunchKernelGGL(k_merge_refine<true>, dim3((unsigned)nq), dim3(RF_THREADS), rlds, st, m);
        }
        else
            hipLaunchKernelGGL(k_merge_refine<false>, dim3((unsigned)nq), dim3(RF_THREADS), refine_lds_bytes(c.cap), st, m);
    }
    RADAD_HIP_CHECK(hipGetLastError());

    // ---- look before the exact pass (see radad_knn_s::verify_next) ---------------------------------------------------------------
    if (c.cert && c.hi_tile && h->verify_next && nq >= 64 && (double)nq * (double)h->ntotal * (double)h->dim >= 4e11) {
        int* rej = h->host_count + 16;
        RADAD_HIP_CHECK(hipMemcpyAsync(rej, flag_count, sizeof(int), hipMemcpyDeviceToHost, st));
        RADAD_HIP_CHECK(hipStreamSynchronize(st));
        if ((int64_t)*rej * 4 > nq && may_retry) {
            knn_retune_after_mass_rejection(h);
            ++h->verified_retries;
            return RADAD_RETRY_INTERNAL;
        }
        if ((int64_t)*rej * 4 <= nq) { h->verify_next = false; h->hi_fail_streak = 0; }
    }

    // ---- the queries the certificate rejected: exact float64 search, sized and driven by the device-side count ------
    if (c.cert) {
        ExactParams x;
        x.db = h->rows; x.db_f16 = h->f16; x.q = c.q_use; x.sel = flag_sel; x.count = flag_count;
        x.n = h->ntotal; x.dim = h->dim; x.k = k; x.l2 = c.l2; x.group = c.xgroup;
        x.slice_rows = ceil_div64(std::max<int64_t>(h->ntotal, 1), KX_SLICES);
        x.pkey = (double*)(ws + c.o_xk); x.pidx = (int*)(ws + c.o_xi); x.id_base = h->id_base;
        x.out_dist = out_dist_dev; x.out_idx = out_idx_dev; x.out_key = out_key_dev;
        x.host_stats = h->host_count_dev + 8 * c.cslot; x.stamp = (int)(h->search_seq & 0x3fffffff) + 1; x.nq_report = (int)std::min<int64_t>(nq, 0x7fffffff);
        const size_t xlds = (size_t)c.xgroup * h->dim * 4 + (size_t)KX_WAVES * c.xgroup * k * 12 + 16;
        RADAD_REQUIRE(xlds <= 160 * 1024, "radad_knn_search: dim %d x k %d too large for the exact kernel", h->dim, k);
        RADAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_exact_scan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)xlds));
        if (nq > h->xarrive_cap) {                       // arrival counters of the query groups (zero between launches)
            RADAD_HIP_CHECK(hipStreamSynchronize(st));
            if (h->xarrive) (void)hipFree(h->xarrive);
            h->xarrive = nullptr; h->xarrive_cap = 0;
            const int64_t cap = nq + nq / 2 + 64;
            if (hipMalloc((void**)&h->xarrive, (size_t)cap * sizeof(int)) != hipSuccess || hipMemset(h->xarrive, 0, (size_t)cap * sizeof(int)) != hipSuccess) {
                (void)hipGetLastError();
                if (h->xarrive) (void)hipFree(h->xarrive);
                h->xarrive = nullptr;
                radad_set_error("hipMalloc of the exact pass's arrival counters failed");
                return RADAD_ENOMEM;
            }
            h->xarrive_cap = cap;
        }
        x.arrive = h->xarrive;
        hipLaunchKernelGGL(k_exact_scan, dim3(KX_SLICES, KX_GROUPS_Y), dim3(KX_THREADS), xlds, st, x);
        RADAD_HIP_CHECK(hipGetLastError());
        h->count_nq[c.cslot] = nq;                                     // (k_exact_scan writes the counters and its stamp to the pinned host copy)
        ++h->search_seq;
    }
    h->last_stream = st;
    h->have_last = true;
    return RADAD_OK;
}

static int knn_search_core(radad_knn_t h, const void* q_in, int q_dtype, int64_t nq, int k, int margin, float* out_dist_dev,
                           int64_t* out_idx_dev, double* out_key_dev, void* stream) {
    RADAD_REQUIRE(h, "NULL handle");
    RADAD_REQUIRE(k >= 1 && k <= RADAD_KNN_MAX_K, "radad_knn_search: k=%d outside [1,%d]", k, RADAD_KNN_MAX_K);
    RADAD_REQUIRE(nq >= 0 && nq < (1ll << 31) - KT_N, "radad_knn_search: bad nq");
    if (nq == 0) return RADAD_OK;
    RADAD_REQUIRE(q_in && out_dist_dev && out_idx_dev, "radad_knn_search: NULL buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    RADAD_REQUIRE(!h->pending.valid, "radad_knn_search: a radad_knn_search_begin on this handle has not been finished");
    for (int attempt = 0;; ++attempt) {       // (at most: plane re-decided, buffers widened, fp32 kernels)
        SearchCtx c;
        int rc = knn_search_phase1(h, q_in, q_dtype, nq, k, margin, nullptr, (hipStream_t)stream, &c);
        if (rc) return rc;
        rc = knn_search_phase2(h, c, nullptr, attempt < 3, out_dist_dev, out_idx_dev, out_key_dev, (hipStream_t)stream);
        if (rc != RADAD_RETRY_INTERNAL) return rc;
    }
}

extern "C" {

int radad_knn_search_begin(radad_knn_t h, const void* q_dev, int q_dtype, int64_t nq, int k, float* kth_lower_bound_dev, void* stream) {
    RADAD_REQUIRE(h, "NULL handle");
    RADAD_REQUIRE(q_dtype == RADAD_Q_F32 || q_dtype == RADAD_Q_BF16, "radad_knn_search_begin: unsupported query dtype %d", q_dtype);
    RADAD_REQUIRE(k >= 1 && k <= RADAD_KNN_MAX_K, "radad_knn_search_begin: k=%d outside [1,%d]", k, RADAD_KNN_MAX_K);
    RADAD_REQUIRE(nq > 0 && nq < (1ll << 31) - KT_N, "radad_knn_search_begin: bad nq");
    RADAD_REQUIRE(q_dev && kth_lower_bound_dev, "radad_knn_search_begin: NULL buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    RADAD_REQUIRE(!h->pending.valid, "radad_knn_search_begin: the previous begin on this handle has not been finished");
    SearchCtx c;
    int rc = knn_search_phase1(h, q_dev, q_dtype, nq, k, KNN_MARGIN, kth_lower_bound_dev, (hipStream_t)stream, &c);
    if (rc) return rc;
    RADAD_HIP_CHECK(hipEventRecord(h->ev_begun, (hipStream_t)stream));
    h->pending = c;
    return RADAD_OK;
}

int radad_knn_search_abort(radad_knn_t h) {
    RADAD_REQUIRE(h, "NULL handle");
    std::lock_guard<std::mutex> lk(h->mu);
    h->pending.valid = false;                 // (the workspace of the abandoned first half is simply reused by the next search)
    return RADAD_OK;
}

int radad_knn_search_finish(radad_knn_t h, const float* global_lower_bound_dev, float* out_dist_dev, int64_t* out_idx_dev,
                            double* out_key_dev, void* stream) {
    RADAD_REQUIRE(h, "NULL handle");
    RADAD_REQUIRE(out_dist_dev && out_idx_dev, "radad_knn_search_finish: NULL buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    RADAD_REQUIRE(h->pending.valid, "radad_knn_search_finish: no search was begun on this handle");
    const SearchCtx c = h->pending;
    h->pending.valid = false;
    RADAD_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, h->ev_begun, 0));       // the scan of _begin (no-op on the same stream)
    return knn_search_phase2(h, c, global_lower_bound_dev, false, out_dist_dev, out_idx_dev, out_key_dev, (hipStream_t)stream);
}

}  // extern "C"

extern "C" {

int radad_knn_search_host(radad_knn_t h, const float* q_host, int64_t nq, int k, float* out_dist_host,
                          int64_t* out_idx_host) {
    RADAD_REQUIRE(h, "NULL handle");
    if (nq == 0) return RADAD_OK;
    RADAD_REQUIRE(q_host && out_dist_host && out_idx_host && nq > 0 && k >= 1, "radad_knn_search_host: bad argument");
    DeviceGuard g(h->device);
    float *dq = nullptr, *dd = nullptr;
    int64_t* di = nullptr;
    const size_t qb = (size_t)nq * h->dim * sizeof(float);
    int rc = RADAD_OK;
    if (hipMalloc(&dq, qb) != hipSuccess || hipMalloc(&dd, (size_t)nq * k * sizeof(float)) != hipSuccess ||
        hipMalloc(&di, (size_t)nq * k * sizeof(int64_t)) != hipSuccess) {
        radad_set_error("hipMalloc of search staging buffers failed");
        rc = RADAD_ENOMEM;
    }
    if (rc == RADAD_OK && hipMemcpy(dq, q_host, qb, hipMemcpyHostToDevice) != hipSuccess) { radad_set_error("H2D copy failed"); rc = RADAD_EHIP; }
    if (rc == RADAD_OK) rc = radad_knn_search(h, dq, nq, k, dd, di, nullptr);
    if (rc == RADAD_OK && (hipMemcpy(out_dist_host, dd, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
                           hipMemcpy(out_idx_host, di, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess)) {
        radad_set_error("D2H copy failed: %s", hipGetErrorString(hipGetLastError()));
        rc = RADAD_EHIP;
    }
    if (dq) (void)hipFree(dq);
    if (dd) (void)hipFree(dd);
    if (di) (void)hipFree(di);
    return rc;
}

int radad_knn_reconstruct(radad_knn_t h, const int64_t* idx_dev, int64_t n, float* out_dev, void* stream) {
    RADAD_REQUIRE(h, "NULL handle");
    if (n == 0) return RADAD_OK;
    RADAD_REQUIRE(idx_dev && out_dev && n > 0, "radad_knn_reconstruct: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    if (h->f16)
        hipLaunchKernelGGL(k_gather_rows<_Float16>, dim3((unsigned)ceil_div64(n, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const _Float16*)h->rows, idx_dev, n, h->ntotal, h->id_base, h->dim, out_dev);
    else
        hipLaunchKernelGGL(k_gather_rows<float>, dim3((unsigned)ceil_div64(n, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)h->rows, idx_dev, n, h->ntotal, h->id_base, h->dim, out_dev);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

int radad_knn_reconstruct_host(radad_knn_t h, const int64_t* idx_host, int64_t n, float* out_host) {
    RADAD_REQUIRE(h, "NULL handle");
    if (n == 0) return RADAD_OK;
    RADAD_REQUIRE(idx_host && out_host && n > 0, "radad_knn_reconstruct_host: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    // small n (pipeline.py:503 fetches K rows per query): copy row by row straight from the store
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = idx_host[i] - h->id_base;
        float* dst = out_host + i * h->dim;
        if (idx_host[i] < 0 || r < 0 || r >= h->ntotal) { memset(dst, 0, sizeof(float) * h->dim); continue; }
        if (h->f16) {
            std::vector<_Float16> tmp((size_t)h->dim);
            RADAD_HIP_CHECK(hipMemcpy(tmp.data(), h->rows + (size_t)r * h->row_bytes(), h->row_bytes(), hipMemcpyDeviceToHost));
            for (int c = 0; c < h->dim; ++c) dst[c] = (float)tmp[c];
        } else {
            RADAD_HIP_CHECK(hipMemcpy(dst, h->rows + (size_t)r * h->row_bytes(), h->row_bytes(), hipMemcpyDeviceToHost));
        }
    }
    return RADAD_OK;
}

int radad_knn_profile(radad_knn_t h, int enable) {
    RADAD_REQUIRE(h, "NULL handle");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    int rc = h->prof.enable(enable != 0, enable);
    if (rc) radad_set_error("hipEventCreate failed");
    return rc;
}

int radad_knn_profile_read(radad_knn_t h, float* ms_out, int cap, int* n_out) {
    RADAD_REQUIRE(h && ms_out && cap >= 0, "radad_knn_profile_read: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    int rc = h->prof.read(ms_out, cap, n_out);
    if (rc) radad_set_error("reading profile events failed");
    return rc;
}

int radad_knn_last_recheck(radad_knn_t h, int* n_queries) {
    RADAD_REQUIRE(h && n_queries, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    const int ls = (int)((h->search_seq + 1) & 1);       // slot of the most recent certified search
    RADAD_HIP_CHECK(knn_wait_last_search(h));
    *n_queries = (h->host_count && h->search_seq) ? h->host_count[8 * ls] : 0;
    return RADAD_OK;
}

int radad_knn_last_certificate(radad_knn_t h, int64_t* n_queries, int* stats6) {
    RADAD_REQUIRE(h && stats6, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    const int ls = (int)((h->search_seq + 1) & 1);       // slot of the most recent certified search
    RADAD_HIP_CHECK(knn_wait_last_search(h));
    for (int i = 0; i < 6; ++i) stats6[i] = (h->host_count && h->search_seq) ? h->host_count[8 * ls + i] : 0;
    if (n_queries) *n_queries = h->search_seq ? h->count_nq[ls] : 0;
    return RADAD_OK;
}

int radad_knn_last_launch(radad_knn_t h, int* n_query_tiles, int* n_db_splits, int* block_threads) {
    RADAD_REQUIRE(h, "NULL handle");
    if (n_query_tiles) *n_query_tiles = h->last_qtiles;
    if (n_db_splits) *n_db_splits = h->last_splits;
    if (block_threads) *block_threads = h->last_threads;
    return RADAD_OK;
}

int radad_knn_plane_info(radad_knn_t h, int* built, int* centred, int* one_scale) {
    RADAD_REQUIRE(h, "NULL handle");
    std::lock_guard<std::mutex> lk(h->mu);
    if (built) *built = (h->hi != nullptr && h->hi_rows > 0) ? 1 : 0;
    if (centred) *centred = h->cmu != nullptr ? 1 : 0;
    if (one_scale) *one_scale = (h->hi != nullptr && h->rscale == nullptr) ? 1 : 0;
    return RADAD_OK;
}

int radad_knn_plane_rebuilds(radad_knn_t h, int* n_out) {
    RADAD_REQUIRE(h && n_out, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    *n_out = h->plane_rebuilds;
    return RADAD_OK;
}

int radad_knn_last_emitted(radad_knn_t h, int* counts_host, float* floors_host, int64_t nq) {
    RADAD_REQUIRE(h && counts_host, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    RADAD_REQUIRE(h->last_emit_nq > 0 && nq == h->last_emit_nq && h->ws, "radad_knn_last_emitted: the last search of this handle was not a tile scan of %lld queries", (long long)nq);
    RADAD_HIP_CHECK(knn_wait_last_search(h));
    RADAD_HIP_CHECK(hipMemcpy(counts_host, (const char*)h->ws + h->last_o_cnt, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost));
    if (floors_host) RADAD_HIP_CHECK(hipMemcpy(floors_host, (const char*)h->ws + h->last_o_thr, (size_t)nq * sizeof(float), hipMemcpyDeviceToHost));
    return RADAD_OK;
}

int radad_knn_scan_geometry(int64_t n_rows, int64_t n_queries, int* query_tiles, int* chunks, int64_t* rows_per_chunk) {
    RADAD_REQUIRE(n_rows >= 1 && n_queries >= 1 && query_tiles && chunks && rows_per_chunk, "radad_knn_scan_geometry: bad argument");
    knn_geometry_wide(n_rows, n_queries, query_tiles, chunks, rows_per_chunk);
    return RADAD_OK;
}

int radad_knn_last_scan_phases(radad_knn_t h, int* n_phases) {
    RADAD_REQUIRE(h && n_phases, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    *n_phases = h->last_scan_phases;
    return RADAD_OK;
}

int radad_knn_tuning_info(radad_knn_t h, int* cap_boost, int* fp32_searches_left, int64_t* reports_consumed) {
    RADAD_REQUIRE(h && cap_boost && fp32_searches_left && reports_consumed, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    *cap_boost = h->cap_boost;
    *fp32_searches_left = h->hi_skip;
    *reports_consumed = h->reports_consumed;
    return RADAD_OK;
}

int radad_knn_last_scan_launches(radad_knn_t h, int* n_launches) {
    RADAD_REQUIRE(h && n_launches, "NULL argument");
    *n_launches = h->last_scan_launches;
    return RADAD_OK;
}

int radad_knn_last_scan_kind(radad_knn_t h, int* kind) {
    RADAD_REQUIRE(h && kind, "NULL argument");
    *kind = h->last_kind;
    return RADAD_OK;
}

// ---- snapshot: "RADADKNN" | u32 version (2) | i32 dim | i32 metric | i32 store dtype | i64 ntotal | rows as stored
// (already normalised for cosine; fp32 or fp16).  Version 1 files (no dtype field, fp32 rows) are still read.
}  // extern "C"

namespace {

constexpr uint32_t SNAP_VERSION = 2;
constexpr size_t SNAP_STAGE_BYTES = 32ull << 20;   // two pinned staging buffers of this size per transfer

struct SnapHeader {
    uint32_t ver = 0;
    int32_t dim = 0, metric = 0, dtype = RADAD_STORE_F32;
    int64_t ntotal = 0;
    size_t payload_off = 0;   // byte offset of row 0
};

// "RADADKNN" | u32 ver | i32 dim | i32 metric | [v2: i32 dtype] | i64 ntotal | rows
bool snap_parse(const unsigned char* p, size_t len, SnapHeader* out) {
    if (len < 28 || memcmp(p, "RADADKNN", 8) != 0) return false;
    size_t o = 8;
    memcpy(&out->ver, p + o, 4); o += 4;
    if (out->ver != 1 && out->ver != 2) return false;
    memcpy(&out->dim, p + o, 4); o += 4;
    memcpy(&out->metric, p + o, 4); o += 4;
    out->dtype = RADAD_STORE_F32;
    if (out->ver >= 2) { memcpy(&out->dtype, p + o, 4); o += 4; }
    if (len < o + 8) return false;
    memcpy(&out->ntotal, p + o, 8); o += 8;
    out->payload_off = o;
    return out->dim > 0 && out->ntotal >= 0 && (out->dtype == RADAD_STORE_F32 || out->dtype == RADAD_STORE_F16);
}

// read-only mapping of a snapshot file; pages come in on demand, so a shard only touches its own byte range
struct SnapMap {
    int fd = -1;
    const unsigned char* base = nullptr;
    size_t len = 0;
    ~SnapMap() {
        if (base) munmap(const_cast<unsigned char*>(base), len);
        if (fd >= 0) close(fd);
    }
    int open_ro(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) { radad_set_error("cannot open %s", path); return RADAD_EIO; }
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size <= 0) { radad_set_error("%s: cannot stat / empty file", path); return RADAD_EIO; }
        len = (size_t)st.st_size;
        void* m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { radad_set_error("%s: mmap failed", path); return RADAD_EIO; }
        base = static_cast<const unsigned char*>(m);
        return RADAD_OK;
    }
};

// two pinned buffers + a private stream: the CPU fills (or drains) one buffer while the DMA engine moves the other
struct PinnedPipe {
    char* buf[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    ~PinnedPipe() {
        for (int i = 0; i < 2; ++i) {
            if (done[i]) (void)hipEventDestroy(done[i]);
            if (buf[i]) (void)hipHostFree(buf[i]);
        }
        if (st) (void)hipStreamDestroy(st);
    }
    int init() {
        RADAD_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            RADAD_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&buf[i]), SNAP_STAGE_BYTES, hipHostMallocDefault));
            RADAD_HIP_CHECK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        }
        return RADAD_OK;
    }
};

int snap_load_range(radad_knn_t h, const char* path, int64_t row0, int64_t n_rows) {
    SnapMap map;
    int rc = map.open_ro(path);
    if (rc) return rc;
    SnapHeader hd;
    const int32_t want = h->f16 ? RADAD_STORE_F16 : RADAD_STORE_F32;
    if (!snap_parse(map.base, map.len, &hd) || hd.dim != h->dim || hd.metric != h->metric || hd.dtype != want) {
        radad_set_error("%s: bad header or mismatch (file dim %d metric %d dtype %d, handle dim %d metric %d dtype %d)", path, hd.dim,
                        hd.metric, hd.dtype, h->dim, h->metric, want);
        return RADAD_EIO;
    }
    const size_t rb = h->row_bytes();
    if (map.len < hd.payload_off + (size_t)hd.ntotal * rb) { radad_set_error("%s: truncated", path); return RADAD_EIO; }
    if (n_rows < 0) n_rows = hd.ntotal - row0;
    RADAD_REQUIRE(row0 >= 0 && n_rows >= 0 && row0 + n_rows <= hd.ntotal, "radad_knn_load_range: rows outside the snapshot");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    RADAD_REQUIRE(!h->pending.valid, "radad_knn_load: a begun search (radad_knn_search_begin) must be finished or aborted before the store changes");
    RADAD_HIP_CHECK(hipDeviceSynchronize());
    h->ntotal = 0;
    h->hi_rows = 0; h->stat_rows = 0;      // the hi plane and the statistics describe the old contents
    if (h->stat) RADAD_HIP_CHECK(hipMemset(h->stat, 0, 4 * sizeof(unsigned)));
    if ((rc = knn_grow(h, n_rows))) return rc;
    const unsigned char* src = map.base + hd.payload_off + (size_t)row0 * rb;
    const size_t total = (size_t)n_rows * rb;
    if (total) {
        (void)madvise(const_cast<unsigned char*>(map.base), map.len, MADV_SEQUENTIAL);
        PinnedPipe pipe;
        if ((rc = pipe.init())) return rc;
        int slot = 0;
        for (size_t off = 0; off < total; off += SNAP_STAGE_BYTES, slot ^= 1) {
            const size_t m = std::min(SNAP_STAGE_BYTES, total - off);
            RADAD_HIP_CHECK(hipEventSynchronize(pipe.done[slot]));      // the copy that last used this buffer has landed
            memcpy(pipe.buf[slot], src + off, m);                       // page-in + stage; overlaps the other buffer's DMA
            RADAD_HIP_CHECK(hipMemcpyAsync(h->rows + off, pipe.buf[slot], m, hipMemcpyHostToDevice, pipe.st));
            RADAD_HIP_CHECK(hipEventRecord(pipe.done[slot], pipe.st));
        }
        RADAD_HIP_CHECK(hipStreamSynchronize(pipe.st));
    }
    if (h->metric == RADAD_METRIC_L2 && n_rows > 0) {      // |y|^2 is not part of the snapshot
        const unsigned grid = (unsigned)ceil_div64(n_rows, 4);
        if (h->f16) hipLaunchKernelGGL(k_row_sqnorm<_Float16>, dim3(grid), dim3(256), 0, nullptr, (const _Float16*)h->rows, n_rows, h->dim, h->ynorm);
        else hipLaunchKernelGGL(k_row_sqnorm<float>, dim3(grid), dim3(256), 0, nullptr, (const float*)h->rows, n_rows, h->dim, h->ynorm);
        RADAD_HIP_CHECK(hipGetLastError());
        RADAD_HIP_CHECK(hipDeviceSynchronize());
    }
    h->ntotal = n_rows;
    return RADAD_OK;
}

}  // namespace

extern "C" {

int radad_knn_save(radad_knn_t h, const char* path) {
    RADAD_REQUIRE(h && path, "radad_knn_save: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard g(h->device);
    RADAD_HIP_CHECK(hipDeviceSynchronize());
    FILE* f = fopen(path, "wb");
    if (!f) { radad_set_error("cannot open %s for writing", path); return RADAD_EIO; }
    const char magic[8] = {'R', 'A', 'D', 'A', 'D', 'K', 'N', 'N'};
    const uint32_t ver = SNAP_VERSION;
    const int32_t dim = h->dim, metric = h->metric, dtype = h->f16 ? RADAD_STORE_F16 : RADAD_STORE_F32;
    const int64_t nt = h->ntotal;
    bool ok = fwrite(magic, 1, 8, f) == 8 && fwrite(&ver, 4, 1, f) == 1 && fwrite(&dim, 4, 1, f) == 1 &&
              fwrite(&metric, 4, 1, f) == 1 && fwrite(&dtype, 4, 1, f) == 1 && fwrite(&nt, 8, 1, f) == 1;
    const size_t total = (size_t)nt * h->row_bytes();
    int rc = RADAD_OK;
    if (ok && total) {
        PinnedPipe pipe;
        if ((rc = pipe.init())) { fclose(f); return rc; }
        // D2H of chunk i+1 runs while chunk i is written to the file
        auto fetch = [&](size_t off, int slot) {
            const size_t m = std::min(SNAP_STAGE_BYTES, total - off);
            if (hipMemcpyAsync(pipe.buf[slot], h->rows + off, m, hipMemcpyDeviceToHost, pipe.st) != hipSuccess) return false;
            return hipEventRecord(pipe.done[slot], pipe.st) == hipSuccess;
        };
        int slot = 0;
        ok = fetch(0, 0);
        for (size_t off = 0; ok && off < total; off += SNAP_STAGE_BYTES, slot ^= 1) {
            const size_t m = std::min(SNAP_STAGE_BYTES, total - off);
            if (off + SNAP_STAGE_BYTES < total) ok = fetch(off + SNAP_STAGE_BYTES, slot ^ 1);
            ok = ok && hipEventSynchronize(pipe.done[slot]) == hipSuccess;
            ok = ok && fwrite(pipe.buf[slot], 1, m, f) == m;
        }
        (void)hipStreamSynchronize(pipe.st);
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { radad_set_error("write to %s failed", path); return RADAD_EIO; }
    return RADAD_OK;
}

int radad_knn_load(radad_knn_t h, const char* path) {
    RADAD_REQUIRE(h && path, "radad_knn_load: NULL argument");
    return snap_load_range(h, path, 0, -1);
}

int radad_knn_load_range(radad_knn_t h, const char* path, int64_t row0, int64_t n_rows) {
    RADAD_REQUIRE(h && path, "radad_knn_load_range: NULL argument");
    return snap_load_range(h, path, row0, n_rows);
}

int radad_knn_snapshot_info(const char* path, int* dim_out, int* metric_out, int* store_dtype_out, int64_t* ntotal_out) {
    RADAD_REQUIRE(path, "radad_knn_snapshot_info: NULL path");
    FILE* f = fopen(path, "rb");
    if (!f) { radad_set_error("cannot open %s", path); return RADAD_EIO; }
    unsigned char head[32];
    const size_t got = fread(head, 1, sizeof(head), f);
    fclose(f);
    SnapHeader hd;
    if (!snap_parse(head, got, &hd)) { radad_set_error("%s: not a RADADKNN snapshot", path); return RADAD_EIO; }
    if (dim_out) *dim_out = hd.dim;
    if (metric_out) *metric_out = hd.metric;
    if (store_dtype_out) *store_dtype_out = hd.dtype;
    if (ntotal_out) *ntotal_out = hd.ntotal;
    return RADAD_OK;
}

int radad_topk_merge(int metric, const float* in_dist_dev, const int64_t* in_idx_dev, int n_parts, int64_t nq, int k,
                     float* out_dist_dev, int64_t* out_idx_dev, int device, void* stream) {
    return merge_lists<float>(metric, in_dist_dev, in_idx_dev, n_parts, nq, k, out_dist_dev, out_idx_dev, nullptr, device, stream);
}

int radad_topk_merge_f64(int metric, const double* in_key_dev, const int64_t* in_idx_dev, int n_parts, int64_t nq, int k,
                         float* out_dist_dev, int64_t* out_idx_dev, double* out_key_dev, int device, void* stream) {
    return merge_lists<double>(metric, in_key_dev, in_idx_dev, n_parts, nq, k, out_dist_dev, out_idx_dev, out_key_dev, device,
                               stream);
}

int radad_filter_topk(const float* in_dist_dev, const int64_t* in_idx_dev, int64_t nq, int k_in, int k_keep,
                      const int64_t* row_tags_dev, int64_t ntotal, int64_t id_base, const int64_t* excl_sorted_dev, int64_t n_excl,
                      float* out_dist_dev, int64_t* out_idx_dev, int device, void* stream) {
    RADAD_REQUIRE(nq >= 0 && k_in >= 0 && k_keep >= 1 && n_excl >= 0 && ntotal >= 0, "radad_filter_topk: bad shape");
    if (nq == 0) return RADAD_OK;
    RADAD_REQUIRE(out_dist_dev && out_idx_dev && (k_in == 0 || (in_dist_dev && in_idx_dev)), "radad_filter_topk: NULL buffer");
    RADAD_REQUIRE(n_excl == 0 || (excl_sorted_dev && row_tags_dev), "radad_filter_topk: exclusion set without tags");
    DeviceGuard g(device);
    hipLaunchKernelGGL(k_filter_topk, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, (hipStream_t)stream, in_dist_dev, in_idx_dev,
                       nq, k_in, k_keep, row_tags_dev, ntotal, id_base, excl_sorted_dev, n_excl, out_dist_dev, out_idx_dev);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

int radad_kth_largest(const float* in_dev, int64_t n, int groups, int per_group, int k, float* out_dev, int device, void* stream) {
    RADAD_REQUIRE(n >= 0 && groups >= 1 && per_group >= 1 && (int64_t)groups * per_group <= 64 * KTH_PER_LANE && k >= 1 && k <= groups * per_group,
                  "radad_kth_largest: bad shape (groups x per_group <= %d, 1 <= k <= groups x per_group)", 64 * KTH_PER_LANE);
    if (n == 0) return RADAD_OK;
    RADAD_REQUIRE(in_dev && out_dev, "radad_kth_largest: NULL buffer");
    DeviceGuard g(device);
    hipLaunchKernelGGL(k_kth_largest, dim3((unsigned)ceil_div64(n, 4)), dim3(256), 0, (hipStream_t)stream, in_dev, n, groups, per_group, k, out_dev);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

int radad_rownorm(const float* in_dev, float* out_dev, int64_t n, int dim, int device, void* stream) {
    RADAD_REQUIRE(dim > 0 && dim % 4 == 0 && n >= 0, "radad_rownorm: bad shape");
    if (n == 0) return RADAD_OK;
    RADAD_REQUIRE(in_dev && out_dev, "radad_rownorm: NULL buffer");
    DeviceGuard g(device);
    hipLaunchKernelGGL(k_rows_prepare<float>, dim3((unsigned)ceil_div64(n, 4)), dim3(256), 0, (hipStream_t)stream, in_dev, out_dev,
                       (float*)nullptr, n, dim, 2);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

}  // extern "C"

#include "ivf.inc"
