// M1-M3 -- 256-bit Hamming brute-force matcher.
// Replaces Matcher::DescriptorDistance (src/Algorithm/Matcher.cpp:30-43) and
// cv::BFMatcher(NORM_HAMMING, crossCheck).match() (test/test_orb_match.cpp:86-93).
// Bit-exact with oracle/hamming.c (integer arithmetic; first minimum wins ties).
//
// k_hamming_nn: lane = one or two rows of set A with their 256-bit descriptors in 8 / 16 VGPRs; set B streams
// through LDS in 256-row tiles (two 16-byte coalesced loads per lane to stage, then every lane
// reads the SAME tile row -> LDS broadcast, conflict-free); per pair 8 x (v_xor + v_bcnt with
// accumulate) and a running minimum (a packed (distance, row) key, or (min, 2nd min, argmin) when the second-best distance
// is wanted).  The kernel is VALU-bound by two
// orders of magnitude (72 KB per 1000x1000 pair vs 1.6e7 lane-ops), so the layout
// goal is only that the 64 KB of descriptors are read from HBM exactly once per workgroup column.
// Cross-check (OpenCV batchDistance semantics) = the same kernel run train->query with a fused
// epilogue: atomicMin(key[nearest query], dist<<32 | train row) -- min distance, then lowest
// train index, exactly the order-dependent "strictly smaller replaces" rule of OpenCV.
#include "ygz_internal.h"
#include <string.h>
#include <vector>
#include <stdlib.h>

#define HM_TILE 256
#ifndef HM_ROWS
#define HM_ROWS 2
#endif
#define HM_BCNT(acc, x) asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x))

struct HamArgs {
    const uint32_t *desc;        // descriptor store: set s at desc + s*set_stride (u32 units)
    size_t set_stride;
    const int32_t *set_count;    // [sets] rows per set (device)
    const int32_t *pair_a, *pair_b;   // [pairs] set ids: every row of a searches b
    size_t out_stride;           // rows reserved per pair in the outputs
    int32_t *out_idx, *out_dist, *out_dist2;     // [pairs][out_stride]; dist2 may be null
    unsigned long long *scatter_key;             // cross-check target [pairs][out_stride] or null
};

// R = rows of A per lane (8 R VGPRs); a workgroup covers 256 R rows of A.  Set B streams through LDS in 256-row tiles (one row
// staged per lane), every lane reads the SAME tile row (broadcast) and uses it for its R rows.  Per pair of rows: 8 v_xor,
// 8 chained v_bcnt_u32_b32 (the popcount adds to an accumulator operand; left to itself the compiler builds 8 independent
// popcounts and a 3-instruction add tree) and one v_min_u32 on the key (distance << 16 | row) -- first minimum wins.
// Measured alternatives at 256 pairs x 968 x 968 rows (MI355X, 203 us per launch for this form, 58 % VALU-busy by the SQ
// counters): B rows through the scalar data cache into SGPRs (no LDS, no barriers) 208 us; R = 1: 215 us, R = 4: 274 us;
// B split over 2-4 times as many, shorter wavefronts: 250-266 us; the compiler's own popcount tree instead of the chain: 240 us.
template <bool SECOND, int R>
__global__ __launch_bounds__(256) void k_hamming_nn(HamArgs A)
{
    __shared__ __attribute__((aligned(16))) uint32_t tb[HM_TILE][8];
    // grid = (pairs, row chunks): the pair index runs fastest, so the chunks that exist (sets hold ~1000 of `cells` rows) are
    // dispatched first and spread evenly over the CUs, and the chunks of one pair (same set B) stay on one XCD (pair mod 8)
    const int p = blockIdx.x, chunk = blockIdx.y;
    const int sa = A.pair_a[p], sb = A.pair_b[p];
    const int nA = A.set_count[sa], nB = A.set_count[sb];
    if ((int)(chunk * 256 * R) >= nA) return;                       // block-uniform
    const int row0 = chunk * 256 * R + threadIdx.x;                 // rows row0 + 256 r
    const uint32_t *da = A.desc + (size_t)sa * A.set_stride;
    const uint32_t *db = A.desc + (size_t)sb * A.set_stride;
    uint4 a0[R], a1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        a0[r] = make_uint4(0, 0, 0, 0); a1[r] = a0[r];
        if (row0 + 256 * r < nA) {
            a0[r] = reinterpret_cast<const uint4 *>(da + 8 * (size_t)(row0 + 256 * r))[0];
            a1[r] = reinterpret_cast<const uint4 *>(da + 8 * (size_t)(row0 + 256 * r))[1];
        }
    }
    int best[R], second[R], bi[R];
    uint32_t bk[R];                                                  // rows < 65536 (checked by the caller)
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = 0x7FFFFFFF; second[r] = 0x7FFFFFFF; bi[r] = -1; bk[r] = 0xFFFFFFFFu; }
#define HM_ROWS_OF(b0x, b0y, b0z, b0w, b1x, b1y, b1z, b1w, j)                                                       \
    _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                             \
        int d = __popc(a0[r].x ^ (b0x));                                                                        \
        HM_BCNT(d, a0[r].y ^ (b0y)); HM_BCNT(d, a0[r].z ^ (b0z)); HM_BCNT(d, a0[r].w ^ (b0w));                  \
        HM_BCNT(d, a1[r].x ^ (b1x)); HM_BCNT(d, a1[r].y ^ (b1y)); HM_BCNT(d, a1[r].z ^ (b1z)); HM_BCNT(d, a1[r].w ^ (b1w)); \
        if (SECOND) { const bool lt = d < best[r]; second[r] = lt ? best[r] : min(second[r], d); bi[r] = lt ? (j) : bi[r]; best[r] = lt ? d : best[r]; } \
        else bk[r] = min(bk[r], ((uint32_t)d << 16) + (uint32_t)(j)); }
    for (int j0 = 0; j0 < nB; j0 += HM_TILE) {
        __syncthreads();
        const int jr = j0 + threadIdx.x;
        if (jr < nB) {
            const uint4 *src = reinterpret_cast<const uint4 *>(db + 8 * (size_t)jr);
            reinterpret_cast<uint4 *>(tb[threadIdx.x])[0] = src[0];
            reinterpret_cast<uint4 *>(tb[threadIdx.x])[1] = src[1];
        }
        __syncthreads();
        const int cnt = min(HM_TILE, nB - j0);
        int jj = 0;
        for (; jj + 4 <= cnt; jj += 4) {
            uint4 b0[4], b1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { b0[u] = reinterpret_cast<const uint4 *>(tb[jj + u])[0]; b1[u] = reinterpret_cast<const uint4 *>(tb[jj + u])[1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { HM_ROWS_OF(b0[u].x, b0[u].y, b0[u].z, b0[u].w, b1[u].x, b1[u].y, b1[u].z, b1[u].w, j0 + jj + u) }
        }
        for (; jj < cnt; ++jj) {
            const uint4 b0 = reinterpret_cast<const uint4 *>(tb[jj])[0], b1 = reinterpret_cast<const uint4 *>(tb[jj])[1];
            HM_ROWS_OF(b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, j0 + jj)
        }
    }
#undef HM_ROWS_OF
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + 256 * r;
        if (row >= nA) continue;
        if (!SECOND && bk[r] != 0xFFFFFFFFu) { best[r] = (int)(bk[r] >> 16); bi[r] = (int)(bk[r] & 0xFFFFu); }
        const size_t o = (size_t)p * A.out_stride + row;
        A.out_idx[o] = bi[r]; A.out_dist[o] = best[r];
        if (SECOND && A.out_dist2) A.out_dist2[o] = second[r];
        if (A.scatter_key && bi[r] >= 0)
            atomicMin(&A.scatter_key[(size_t)p * A.out_stride + bi[r]], ((unsigned long long)(uint32_t)best[r] << 32) | (uint32_t)row);
    }
}

// ---- the nearest-neighbour search on the matrix cores ---------------------------------------------------------------------------------------
// With a' = 1 - 2a in {+1, -1}:  popcount(a ^ b) = |a| + a'.b  (a'.b = |b| - 2 a.b): the 1000 x 1000 distance matrix of a frame pair is a GEMM
// with K = 256 (2.6e8 multiply-adds) plus a term that is constant along a row, so the search runs on an MFMA and the VALU only has to (a) turn
// bits into operand elements and (b) keep the running minimum.  The accumulator chain of a tile STARTS from the tile index t and ends as
// 64 (dist - |a|) + t: ordered by distance, then by tile, so the running minimum is ONE v_min per accumulator and the first minimum wins as in
// the reference's scan.  The minima are per (row, column mod 32); every 64 tiles (and at the end) they are folded into 32-bit keys
// (dist - |a| + 512) << 16 | column, reduced over the 32 lanes once, and |a| is added.  Columns past the end of B exist only in the last tile:
// they get a bias no real column has.  A wavefront owns 32 RG rows of set A -- expanded once -- and walks set B in tiles of 32 rows, every
// expanded B operand feeding RG independent accumulator chains.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2)
// + 4 (lane >> 5).  (Rounds 1-2 ran this on v_mfma_i32_32x32x32_i8, 102 us per 512 pairs, also with the expanded B tiles shared through LDS,
// 120 us; both forms were removed in round 5 -- DESIGN.md Appendix B keeps their measurements.)
typedef int hm_v4i __attribute__((ext_vector_type(4)));
typedef int hm_v16i __attribute__((ext_vector_type(16)));

// ---- the same search on the FP4 matrix path of gfx950 ----------------------------------------------------------------------------------------
// v_mfma_scale_f32_32x32x64_f8f6f4 with E2M1 operands takes K = 64 per instruction at the issue cost of the 32x32x32 int8 MFMA
// (tools/ubench/mfma_f4_probe: 33 against 34 cycles, 7.6 against 3.75 P-op/s): the 256-bit rows need FOUR K-steps instead of eight, and
// bit -> nibble is half the expansion of bit -> byte.  Everything is exact: the operands are 0, +-0.5, +-1, +-2 (FP4 codes), every product
// is 0 or +-1, the block scale 2^6 of the A operand makes it 0 or +-64, and the FP32 accumulator chain starts from (float)tile index, so
// an accumulator ends as the INTEGER 64 (dist - |a|) + t (|value| < 2^15) and the running minimum is one
// v_min_f32.  Encoding: a set bit at nibble position q of a B dword is the FP4 code 0.5 (q = 0), 1 (q = 1) or 2 (q = 2): w & 0x1111.., w &
// 0x2222.., w & 0x4444.. are operand registers as they stand; position 3 would be the sign bit, so it is read as (w >> 1) & 0x4444.. (= 2).
// The A side holds a' = 1 - 2a as +-2, +-1, +-0.5, +-0.5 for the four position classes, so every product is +-1.  Lane (l31, half) holds the
// 128 bits [128 half, 128 half + 128) of row / column l31: K-step s = dword s of those, 32 nibbles in 4 registers -- the same map on both
// sides, which is all a dot product needs.  Results are bit-identical to the int8 and VALU forms (tests/test_gpu_switches.py).
typedef float hm_v16f __attribute__((ext_vector_type(16)));
typedef int hm_v8i __attribute__((ext_vector_type(8)));

__device__ __forceinline__ hm_v8i hm_f4_a(uint32_t w)
{
    hm_v8i r = { (int)(0x44444444u | ((w & 0x11111111u) << 3)), (int)(0x22222222u | ((w & 0x22222222u) << 2)),
                 (int)(0x11111111u | ((w & 0x44444444u) << 1)), (int)(0x11111111u | (w & 0x88888888u)), 0, 0, 0, 0 };
    return r;
}
__device__ __forceinline__ hm_v8i hm_f4_b(uint32_t w)
{
    hm_v8i r = { (int)(w & 0x11111111u), (int)(w & 0x22222222u), (int)(w & 0x44444444u), (int)((w >> 1) & 0x44444444u), 0, 0, 0, 0 };
    return r;
}
#define HM_F4_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((a_), (b_), (c_), 4, 4, 0, 133 /* A block scale 2^6 */, 0, 127)

template <int RG>
__global__ __launch_bounds__(64) void k_hamming_f4(HamArgs A)
{
    const int p = blockIdx.x, lane = threadIdx.x;
    const int sa = A.pair_a[p], sb = A.pair_b[p];
    const int nA = A.set_count[sa], nB = A.set_count[sb];
    const int r0 = blockIdx.y * (32 * RG);
    if (r0 >= nA) return;
    const uint32_t *da = A.desc + (size_t)sa * A.set_stride, *db = A.desc + (size_t)sb * A.set_stride;
    const int half = lane >> 5, l31 = lane & 31;
#define HM_LDB(dst_, j_) { dst_ = make_uint4(0, 0, 0, 0); if ((j_) < nB) dst_ = reinterpret_cast<const uint4 *>(db + 8 * (size_t)(j_))[half]; }
    uint4 a_raw[RG], b, bn1, bn2;
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        a_raw[g] = make_uint4(0, 0, 0, 0);
        const int row = r0 + 32 * g + l31;
        if (row < nA) a_raw[g] = reinterpret_cast<const uint4 *>(da + 8 * (size_t)row)[half];
    }
    HM_LDB(b, l31) HM_LDB(bn1, l31 + 32) HM_LDB(bn2, l31 + 64)
    hm_v8i Aop[RG][4];
    int pa[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        const uint4 a = a_raw[g];
        const uint32_t w[4] = { a.x, a.y, a.z, a.w };
        int pc = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) { Aop[g][s] = hm_f4_a(w[s]); pc += __popc(w[s]); }
        pa[g] = pc + __shfl_xor(pc, 32);
    }
    __shared__ uint32_t fin[RG * 16][64];
    // the accumulators are kept POSITIVE (the chain starts from 2^15 + tile index, |64 (dist - |a|)| <= 2^14): positive floats order like their
    // bit patterns, so the running minimum is an integer v_min on the raw registers (fminf would canonicalise both operands first: 3 VALU
    // instructions per accumulator instead of 1)
    int run[RG][16];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) run[g][r] = 0x7f000000;
    const int n_tiles = (nB + 31) >> 5;
    hm_v16f cinit;                                                   // splat(2^15 + tile index inside the block of 64)
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = 32768.f;
    bool folded = false;
    for (int t = 0; t < n_tiles; ++t) {
        const uint32_t w[4] = { b.x, b.y, b.z, b.w };
        b = bn1; bn1 = bn2;
        HM_LDB(bn2, 32 * t + l31 + 96)
        hm_v16f acc[RG];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const hm_v8i Bop = hm_f4_b(w[s]);
#pragma unroll
            for (int g = 0; g < RG; ++g) acc[g] = HM_F4_MFMA(Aop[g][s], Bop, s == 0 ? cinit : acc[g]);
        }
        if (t == n_tiles - 1 && 32 * t + l31 >= nB) {                // the ragged end of B (its rows are zero: they would win)
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = __int_as_float(0x7f000000);
        }
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) run[g][r] = min(run[g][r], __float_as_int(acc[g][r]));
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[r] += 1.f;
        if ((t & 63) == 63 || t == n_tiles - 1) {                    // fold the block of 64 tiles into the 32-bit keys
            const int tb = t & ~63;
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rb = run[g][r];
                    const int v = (int)__int_as_float(rb) - 32768;   // an integer-valued float: 64 (dist - |a|) + tile
                    uint32_t key = ((uint32_t)((v >> 6) + 512) << 16) | (uint32_t)(32 * (tb + (v & 63)) + l31);
                    key = rb != 0x7f000000 ? key : 0xffffffffu;
                    if (folded) key = min(key, fin[16 * g + r][lane]);
                    fin[16 * g + r][lane] = key;
                    run[g][r] = 0x7f000000;
                }
            folded = true;
#pragma unroll
            for (int r = 0; r < 16; ++r) cinit[r] = 32768.f;
        }
    }
#undef HM_LDB
    if (!folded) {
#pragma unroll
        for (int k = 0; k < RG * 16; ++k) fin[k][lane] = 0xffffffffu;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int pass = 0; pass < (32 * RG + 63) / 64; ++pass) {
        const int q = 64 * pass + lane, g = q >> 5, rowl = q & 31;
        const bool live = q < 32 * RG;
        const int reg = (rowl & 3) + 4 * (rowl >> 3), hq = (rowl >> 2) & 1;
        uint32_t key = 0xffffffffu;
        if (live) {
            const uint4 *src = reinterpret_cast<const uint4 *>(&fin[16 * g + reg][32 * hq]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint4 v = src[(j + lane) & 7];
                key = min(key, min(min(v.x, v.y), min(v.z, v.w)));
            }
        }
        int pa_row = 0;
#pragma unroll
        for (int gg = 0; gg < RG; ++gg) { const int v = __shfl(pa[gg], rowl); pa_row = (g == gg) ? v : pa_row; }
        const int row = r0 + q;
        if (live && row < nA) {
            int idx = -1, dist = 0x7FFFFFFF;
            if (nB > 0) { idx = (int)(key & 0xffffu); dist = (int)(key >> 16) - 512 + pa_row; }
            const size_t o = (size_t)p * A.out_stride + row;
            A.out_idx[o] = idx; A.out_dist[o] = dist;
            if (A.scatter_key && idx >= 0)
                atomicMin(&A.scatter_key[(size_t)p * A.out_stride + idx], ((unsigned long long)(uint32_t)dist << 32) | (uint32_t)row);
        }
    }
}

// cross_check 1: decode the scatter keys; cross_check 2: mutual test qi -> tq
__global__ __launch_bounds__(256) void k_match_finalize(const int32_t *__restrict__ set_count, const int32_t *__restrict__ pair_q,
                                                        size_t out_stride, int mode, const unsigned long long *__restrict__ key,
                                                        const int32_t *__restrict__ tq, int32_t *__restrict__ idx,
                                                        int32_t *__restrict__ dist)
{
    const int p = blockIdx.y;
    const int nq = set_count[pair_q[p]];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    const size_t o = (size_t)p * out_stride + i;
    if (mode == 1) {
        const unsigned long long k = key[o];
        if (k == ~0ull) { idx[o] = -1; dist[o] = 0x7FFFFFFF; }
        else { idx[o] = (int32_t)(uint32_t)(k & 0xFFFFFFFFull); dist[o] = (int32_t)(k >> 32); }
    } else {
        const int j = idx[o];
        if (j < 0 || tq[(size_t)p * out_stride + j] != i) { idx[o] = -1; dist[o] = 0x7FFFFFFF; }
    }
}

int ygz_run_match(ygz_hip_ctx *ctx, const uint32_t *desc, size_t set_stride, const int32_t *set_count,
                  const int32_t *pair_q, const int32_t *pair_t, int n_pairs, int max_rows, int cross_check, bool want_second)
{
    const size_t Cn = (size_t)ctx->cells;
    ctx->pf_valid = false;                            // the M3 flags of the previous result are stale
    HamArgs A;
    A.desc = desc; A.set_stride = set_stride; A.set_count = set_count; A.out_stride = Cn;
    const dim3 grid(n_pairs, ygz_div_up(max_rows, 256)), block(256), grid_f(ygz_div_up(max_rows, 256), n_pairs);
    const dim3 grid_s(n_pairs, ygz_div_up(max_rows, 256 * HM_ROWS));
    const bool wide = max_rows > 0xFFFF;              // the <false> kernel packs (distance, row) into one 32-bit key
    static const bool valu_only = [] { const char *e = getenv("YGZ_HAMMING_VALU"); return e && e[0] == '1'; }();   // A/B switch: the VALU form (also what second-best / > 65535-row searches use)
    const dim3 grid_f4(n_pairs, ygz_div_up(max_rows, 32 * 2)), block_m(64);
    if (cross_check == 0 || cross_check == 2) {       // query -> train
        A.pair_a = pair_q; A.pair_b = pair_t;
        A.out_idx = ctx->m_idx; A.out_dist = ctx->m_dist; A.out_dist2 = want_second ? ctx->m_dist2 : nullptr;
        A.scatter_key = nullptr;
        if (want_second || wide) YGZ_LAUNCH(ctx, KID_HAMMING_NN, (k_hamming_nn<true, 1>), grid, block, A);
        else if (valu_only) YGZ_LAUNCH(ctx, KID_HAMMING_NN, (k_hamming_nn<false, HM_ROWS>), grid_s, block, A);
        else YGZ_LAUNCH(ctx, KID_HAMMING_NN, k_hamming_f4<2>, grid_f4, block_m, A);
    }
    if (cross_check == 1 || cross_check == 2) {       // train -> query
        if (cross_check == 1)
            YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->m_key, 0xFF, ((size_t)n_pairs * Cn > (size_t)max_rows ? (size_t)n_pairs * Cn : (size_t)max_rows) * 8, ctx->stream));   // (one pair may use the rows of all pairs: ygz_hip_hamming_match)
        A.pair_a = pair_t; A.pair_b = pair_q;
        A.out_idx = ctx->m_tq; A.out_dist = ctx->m_td; A.out_dist2 = nullptr;
        A.scatter_key = (cross_check == 1) ? ctx->m_key : nullptr;
        if (wide) YGZ_LAUNCH(ctx, KID_HAMMING_NN, (k_hamming_nn<true, 1>), grid, block, A);
        else if (valu_only) YGZ_LAUNCH(ctx, KID_HAMMING_NN, (k_hamming_nn<false, HM_ROWS>), grid_s, block, A);
        else YGZ_LAUNCH(ctx, KID_HAMMING_NN, k_hamming_f4<2>, grid_f4, block_m, A);
        YGZ_LAUNCH(ctx, KID_MATCH_FINALIZE, k_match_finalize, grid_f, block, set_count, pair_q, Cn, cross_check,
                           ctx->m_key, ctx->m_tq, ctx->m_idx, ctx->m_dist);
    }
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

extern "C" {

int ygz_hip_match_slots(ygz_hip_ctx *ctx, const int32_t *query_slot, const int32_t *train_slot, int n_pairs, int cross_check)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !query_slot || !train_slot || n_pairs < 1 || cross_check < 0 || cross_check > 2) return YGZ_E_INVALID;
    if (n_pairs > ctx->prm.max_frames) return YGZ_E_CAPACITY;
    for (int i = 0; i < n_pairs; ++i)
        if (query_slot[i] < 0 || query_slot[i] >= ctx->prm.max_frames || train_slot[i] < 0 || train_slot[i] >= ctx->prm.max_frames)
            return YGZ_E_INVALID;
    // the caller's arrays may be temporaries: the copy engine reads a page-locked copy (no wait for the stream -- a pipeline
    // enqueues this call behind an upload and an extraction that are still running)
    int32_t *st = (int32_t *)ygz_stage(ctx, (size_t)n_pairs * 8);
    if (!st) return YGZ_E_HIP;
    memcpy(st, query_slot, (size_t)n_pairs * 4); memcpy(st + n_pairs, train_slot, (size_t)n_pairs * 4);
    YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->pair_q, st, (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->pair_t, st + n_pairs, (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
    ctx->n_pairs = n_pairs;
    return ygz_run_match(ctx, ctx->kp_desc, (size_t)ctx->cells * 8, ctx->n_kp, ctx->pair_q, ctx->pair_t, n_pairs, ctx->cells,
                     cross_check, false);
}

// resident variant for pipelines: pair tables already uploaded by a previous ygz_hip_match_slots call
int ygz_hip_match_slots_again(ygz_hip_ctx *ctx, int cross_check)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || ctx->n_pairs < 1 || cross_check < 0 || cross_check > 2) return YGZ_E_INVALID;
    YgzAuxScope aux(ctx, 2);
    return ygz_run_match(ctx, ctx->kp_desc, (size_t)ctx->cells * 8, ctx->n_kp, ctx->pair_q, ctx->pair_t, ctx->n_pairs, ctx->cells,
                     cross_check, false);
}

int ygz_hip_get_matches(ygz_hip_ctx *ctx, int pair, int32_t *train_idx, int32_t *dist, int capacity, int *nq_out)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || pair < 0 || pair >= ctx->n_pairs || !nq_out) return YGZ_E_INVALID;
    int32_t qslot = 0, nq = 0;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&qslot, ctx->pair_q + pair, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&nq, ctx->n_kp + qslot, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *nq_out = nq;
    if (nq > capacity) return YGZ_E_CAPACITY;
    const size_t o = (size_t)pair * ctx->cells;
    if (nq > 0) {
        if (train_idx) YGZ_HIPCHK(ctx, hipMemcpyAsync(train_idx, ctx->m_idx + o, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (dist) YGZ_HIPCHK(ctx, hipMemcpyAsync(dist, ctx->m_dist + o, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return YGZ_OK;
}

int ygz_hip_hamming_match(ygz_hip_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int cross_check,
                          int32_t *train_idx, int32_t *dist, int32_t *dist2)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || nq < 0 || nt < 0 || (nq > 0 && !q) || (nt > 0 && !t) || cross_check < 0 || cross_check > 2) return YGZ_E_INVALID;
    if (dist2 && cross_check != 0) return YGZ_E_INVALID;
    // result rows live in the per-pair buffers ([max_frames][cells]); this one pair may use all of them
    if ((size_t)nq > (size_t)ctx->cells * ctx->prm.max_frames || (size_t)nt > (size_t)ctx->cells * ctx->prm.max_frames) return YGZ_E_CAPACITY;
    if (nq == 0) return YGZ_OK;
    const size_t stride_u32 = (size_t)(nq > nt ? nq : nt) * 8 + 8;
    uint8_t *buf = nullptr;
    int rc = ygz_scratch(ctx, SCR_MATCH_Q, 64 + stride_u32 * 4 * 2, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    int32_t hdr[4] = { nq, nt, 0, 1 };          // counts[2], pair_q, pair_t
    YGZ_HIPCHK(ctx, hipMemcpyAsync(buf, hdr, sizeof(hdr), hipMemcpyHostToDevice, ctx->stream));
    uint32_t *d = reinterpret_cast<uint32_t *>(buf + 64);
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d, q, (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
    if (nt > 0) YGZ_HIPCHK(ctx, hipMemcpyAsync(d + stride_u32, t, (size_t)nt * 32, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int32_t *cnt = reinterpret_cast<const int32_t *>(buf);
    rc = ygz_run_match(ctx, d, stride_u32, cnt, cnt + 2, cnt + 3, 1, nq > nt ? nq : nt, cross_check, dist2 != nullptr);
    if (rc != YGZ_OK) return rc;
    if (train_idx) YGZ_HIPCHK(ctx, hipMemcpyAsync(train_idx, ctx->m_idx, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (dist) YGZ_HIPCHK(ctx, hipMemcpyAsync(dist, ctx->m_dist, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (dist2) YGZ_HIPCHK(ctx, hipMemcpyAsync(dist2, ctx->m_dist2, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_pairs = 0;      // per-pair buffers were reused
    return YGZ_OK;
}

// M1-M3 over many host descriptor sets in one call: every set crosses PCIe once, all pairs run in one launch per direction, the
// good-match rule (test/test_orb_match.cpp:95-104) on the device, one copy back.  Rows are [n_pairs][cells] like the resident buffers.
int ygz_hip_match_sets(ygz_hip_ctx *ctx, int n_sets, const uint8_t *const *desc, const int32_t *count, int n_pairs, const int32_t *pair_q,
                       const int32_t *pair_t, int cross_check, int32_t *train_idx, int32_t *dist, uint8_t *good, int32_t *n_good, double *min_dis,
                       double min_floor, double min_ceil, double factor)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || n_sets < 1 || n_pairs < 1 || !desc || !count || !pair_q || !pair_t || cross_check < 0 || cross_check > 2) return YGZ_E_INVALID;
    if (good && (!(min_floor <= min_ceil) || !(factor > 0))) return YGZ_E_INVALID;
    if (n_pairs > ctx->prm.max_frames) return YGZ_E_CAPACITY;               // the per-pair result buffers
    int max_rows = 0;
    for (int s = 0; s < n_sets; ++s) {
        if (count[s] < 0 || (count[s] > 0 && !desc[s])) return YGZ_E_INVALID;
        if (count[s] > ctx->cells) return YGZ_E_CAPACITY;
        max_rows = count[s] > max_rows ? count[s] : max_rows;
    }
    for (int p = 0; p < n_pairs; ++p) if (pair_q[p] < 0 || pair_q[p] >= n_sets || pair_t[p] < 0 || pair_t[p] >= n_sets) return YGZ_E_INVALID;
    const size_t Cn = (size_t)ctx->cells;
    if (max_rows == 0) {
        for (int p = 0; p < n_pairs; ++p) { if (n_good) n_good[p] = 0; if (min_dis) min_dis[p] = min_ceil; }
        return YGZ_OK;
    }
    const size_t stride_u32 = (size_t)max_rows * 8 + 8;
    const size_t hdr_ints = (size_t)n_sets + 2 * (size_t)n_pairs, hdr_bytes = (hdr_ints * 4 + 63) & ~(size_t)63;
    uint8_t *buf = nullptr;
    int rc = ygz_scratch(ctx, SCR_MATCH_Q, hdr_bytes + stride_u32 * 4 * (size_t)n_sets, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    std::vector<int32_t> hdr(hdr_ints);
    for (int s = 0; s < n_sets; ++s) hdr[s] = count[s];
    for (int p = 0; p < n_pairs; ++p) { hdr[(size_t)n_sets + p] = pair_q[p]; hdr[(size_t)n_sets + n_pairs + p] = pair_t[p]; }
    YGZ_HIPCHK(ctx, hipMemcpyAsync(buf, hdr.data(), hdr_ints * 4, hipMemcpyHostToDevice, ctx->stream));
    uint32_t *d = reinterpret_cast<uint32_t *>(buf + hdr_bytes);
    for (int s = 0; s < n_sets; ++s)
        if (count[s] > 0) YGZ_HIPCHK(ctx, hipMemcpyAsync(d + (size_t)s * stride_u32, desc[s], (size_t)count[s] * 32, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));                     // hdr is about to go out of scope for the async copy engine
    const int32_t *cnt = reinterpret_cast<const int32_t *>(buf), *pq = cnt + n_sets, *pt = pq + n_pairs;
    rc = ygz_run_match(ctx, d, stride_u32, cnt, pq, pt, n_pairs, max_rows, cross_check, false);
    if (rc != YGZ_OK) return rc;
    if (good || n_good || min_dis) {
        rc = ygz_launch_match_postfilter(ctx, cnt, pq, n_pairs, min_floor, min_ceil, factor);
        if (rc != YGZ_OK) return rc;
    }
    const size_t rows = (size_t)n_pairs * Cn;
    if (train_idx) YGZ_HIPCHK(ctx, hipMemcpyAsync(train_idx, ctx->m_idx, rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (dist) YGZ_HIPCHK(ctx, hipMemcpyAsync(dist, ctx->m_dist, rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (good) YGZ_HIPCHK(ctx, hipMemcpyAsync(good, ctx->m_good, rows, hipMemcpyDeviceToHost, ctx->stream));
    if (n_good) YGZ_HIPCHK(ctx, hipMemcpyAsync(n_good, ctx->m_good_n, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (min_dis) YGZ_HIPCHK(ctx, hipMemcpyAsync(min_dis, ctx->m_min_dis, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_pairs = 0;      // per-pair buffers were reused
    return YGZ_OK;
}

}  // extern "C"
