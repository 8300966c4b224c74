// L4 -- pyramidal Lucas-Kanade tracker.
// Replaces cv::calcOpticalFlowPyrLK as called by Tracker::TrackKLT (src/Algorithm/Tracker.cpp:92-98:
// 21x21 window, maxLevel 4, 30 iterations / eps 1e-3, OPTFLOW_USE_INITIAL_FLOW).  Arithmetic follows
// the frozen specification in oracle/klt.c (OpenCV generic path: int16 Scharr derivatives, 14-bit
// fixed-point bilinear weights, int16 patch <<5, float normal equations).
//
//  k_klt_pad  the tracker's working images: every pyramid level of the slots in the pair table copied into a
//             buffer with a 24-pixel BORDER_REFLECT_101 frame (what buildOpticalFlowPyramid does with copyMakeBorder),
//             so that no window access in k_klt ever needs border logic.
//  k_scharr   one lane per pixel: 3/10/3 Scharr of the previous image per level (reflect-101 inside
//             the image exactly as calcSharrDeriv), 2 x int16 per pixel, written into a zero-framed buffer
//             (BORDER_CONSTANT 0 around the derivative, as the reference does).
//  k_klt      one wavefront per point, all pyramid levels inside one launch (points are independent,
//             so no per-level launch boundary).  3 lanes per window row x 7 consecutive pixels (21 = 3 x 7):
//             the previous-image patch and its derivatives stay in registers for the whole level; every iteration
//             fetches its 2 x 8 source bytes per lane with 3 aligned dword loads + v_alignbyte (a wave-wide byte gather
//             costs the address unit as much as a dword load), forms the two mismatch sums with full-rate 24-bit
//             mads, and reduces them on the DPP data path in a FIXED order.  The integer terms are identical to the
//             oracle's; each lane adds its 7 terms exactly in integers and the 63 lane sums are combined by a float
//             tree (the oracle adds all 441 terms in float in raster order): same quantities, different rounding,
//             well inside the 1e-5 track tolerance.
//  k_klt3     the 21 x 21 window of the reference: THREE points per wavefront, 21 lanes per point, a lane owns 3 rows x 7
//             pixels; bilinear taps and mismatch sums as v_dot2_i32_i16 on pixel pairs, patch subtraction folded into the dot
//             accumulator, 21-lane segment sums; the wavefront iterates while any of its points is live (see the kernel).
//             k_klt stays for other window sizes and for the debug taps.
#include "ygz_internal.h"
#include <vector>
#include <stdio.h>
#include <stdlib.h>

#define KLT_MAXWIN 21

__device__ __forceinline__ int refl101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * (n - 1) - i;
    return i;
}

#define KLT_BYTE(lo, hi, k) ((int)((((k) < 4) ? ((lo) >> (8 * ((k) & 3))) : ((hi) >> (8 * ((k) & 3)))) & 255u))
typedef uint32_t __attribute__((aligned(1))) klt_u32u;      // 4 adjacent bytes at any address: one (unaligned) global_load_dword

typedef uint32_t klt_u32x4 __attribute__((ext_vector_type(4)));
typedef klt_u32x4 __attribute__((aligned(1))) klt_u32x4u;      // 16 adjacent bytes at any address: one (unaligned) global_load_dwordx4

// padded copy: out[(y+B)*pw + x+B] = in[refl101(y)][refl101(x)], 16 output pixels per lane (4-byte accesses ran these two
// streaming kernels at a third of the HBM rate), once per distinct slot
__global__ __launch_bounds__(256) void k_klt_pad(const uint8_t *__restrict__ img_base, uint8_t *__restrict__ pad_base,
                                                 const int32_t *__restrict__ slots, int w, int h, int n_slots)
{
    int bx_, by_, z_;
    if (!ygz_xcd_remap3(n_slots, bx_, by_, z_)) return;
    const int pw = KLT_PW(w), ph = h + 2 * KLT_B;
    const int chunks = (pw + 15) >> 4, item = bx_ * 256 + (int)threadIdx.x;       // flat (row, 16-pixel chunk) items: no idle lanes at any width
    const int y = item / chunks, x16 = (item - y * chunks) * 16;
    if (y >= ph) return;
    const size_t slot = (size_t)slots[z_];
    const uint8_t *row = img_base + slot * (size_t)w * h + (size_t)refl101(y - KLT_B, h) * w;
    uint8_t *out = pad_base + slot * (size_t)pw * ph + (size_t)y * pw + x16;
    const int sx = x16 - KLT_B;
    if (x16 + 15 < pw && sx >= 0 && sx + 15 < w) { *reinterpret_cast<klt_u32x4u *>(out) = *reinterpret_cast<const klt_u32x4u *>(row + sx); return; }
    for (int q = 0; q < 4 && x16 + 4 * q < pw; ++q) {             // the frame columns and the ragged end of the row: dword by dword
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) v |= (uint32_t)row[refl101(sx + 4 * q + k, w)] << (8 * k);
        *reinterpret_cast<uint32_t *>(out + 4 * q) = v;
    }
}

// calcSharrDeriv: dx = smooth_v(x+1) - smooth_v(x-1), dy = 3/10/3 smooth_h of (row+1 - row-1); output zero-framed.
// Reads the framed copy (its BORDER_REFLECT_101 frame is exactly the border rule of the derivative).  A lane owns 4 pixels x 4 rows:
// six source rows of 6 bytes (two unaligned dword loads each, all twelve in flight together) feed four output rows, and every
// 16-byte store of the wavefront is one contiguous kilobyte (consecutive lanes, consecutive pixels).
__global__ __launch_bounds__(256) void k_scharr(const uint8_t *__restrict__ pad_base, int16_t *__restrict__ deriv_base,
                                                const int32_t *__restrict__ slots, int w, int h, int n_slots)
{
    int bx_, by_, z_;
    if (!ygz_xcd_remap3(n_slots, bx_, by_, z_)) return;
    const int chunks = (w + 3) >> 2, item = bx_ * 256 + (int)threadIdx.x;         // flat (4-row strip, 4-pixel chunk) items
    const int strip = item / chunks, x = (item - strip * chunks) * 4, y0 = 4 * strip;
    if (y0 >= h) return;
    const size_t slot = (size_t)slots[z_];
    const int pw = KLT_PW(w), ph = h + 2 * KLT_B;
    const uint8_t *p = pad_base + slot * (size_t)pw * ph + (size_t)(y0 + KLT_B - 1) * pw + (x + KLT_B - 1);
    uint32_t *deriv = reinterpret_cast<uint32_t *>(deriv_base) + slot * (size_t)pw * ph + (size_t)(y0 + KLT_B) * pw + (x + KLT_B);
    uint32_t lo[6], hi[6];                                        // rows y0-1 .. y0+4 (inside the frame: KLT_B >= 5), columns x-1 .. x+6
#pragma unroll
    for (int r = 0; r < 6; ++r) { lo[r] = *reinterpret_cast<const klt_u32u *>(p + (size_t)r * pw); hi[r] = *reinterpret_cast<const klt_u32u *>(p + (size_t)r * pw + 4); }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (y0 + q >= h) break;
        int t0[6], t1[6];        // trow0 = (s0+s2)*3 + s1*10 ; trow1 = s2 - s0 for columns x-1 .. x+4
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int s0 = KLT_BYTE(lo[q], hi[q], c), s1 = KLT_BYTE(lo[q + 1], hi[q + 1], c), s2 = KLT_BYTE(lo[q + 2], hi[q + 2], c);
            t0[c] = (s0 + s2) * 3 + s1 * 10; t1[c] = s2 - s0;
        }
        uint32_t out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dx = (int16_t)(t0[k + 2] - t0[k]);
            const int dy = (int16_t)((t1[k + 2] + t1[k]) * 3 + t1[k + 1] * 10);
            out[k] = ((uint32_t)(uint16_t)dx) | ((uint32_t)(uint16_t)dy << 16);
        }
        uint32_t *drow = deriv + (size_t)q * pw;
        if (x + 3 < w) *reinterpret_cast<uint4 *>(drow) = make_uint4(out[0], out[1], out[2], out[3]);
        else { for (int k = 0; k < 4; ++k) if (x + k < w) drow[k] = out[k]; }       // the zero frame stays zero
    }
}

struct KltArgs {
    const uint8_t *pad[YGZ_MAX_LEVELS];                            // reflect-framed level images, slot-major
    const int16_t *deriv[YGZ_MAX_LEVELS];                          // zero-framed Scharr images, slot-major
    int w[YGZ_MAX_LEVELS], h[YGZ_MAX_LEVELS];
    int max_level, win, max_count, use_initial_flow, cells, n_pairs;
    double epsilon;            // already squared
    float min_eig_thr;
    const int32_t *pair_q, *pair_t, *trk_n;                        // cur slot, ref slot, points per pair
    const double *trk_px;                                          // [pairs][cells][2] reference pixels
    float *next_pts; uint8_t *status; float *err;                  // [pairs][cells]
    long long *dbg;                                                // optional per-point cycle counters [pairs*cells][4]
};

#define wave_sum_f ygz_wave_sum_f

__device__ __forceinline__ int cv_round_f(float v) { return __float2int_rn(v); }
#define KLT_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

typedef const __attribute__((address_space(1))) uint32_t *klt_gptr;      // global (not flat) loads

// 8 consecutive bytes from an arbitrary byte address: 3 aligned dword loads (one global_load_dwordx3) + v_alignbyte.
// Round 4 measured the alternative, ONE unaligned global_load_dwordx2 (-DYGZ_KLT_UNALIGNED_LOADS): it removes 20 of the 229 VALU instructions of
// a mismatch evaluation in k_klt3 (two v_alignbyte, two v_and and a v_mov per window row) -- and the launch takes exactly as long (2.54 ms
// per 512 pairs either way): the four row gathers of an evaluation keep the CU's one address unit as busy as the 208 remaining instructions
// keep a SIMD, and a misaligned 8-byte access costs it more than an aligned 12-byte one.  The aligned form stays (fewer address-unit
// cycles for the gather-bound kernels that run beside LK in the step).
__device__ __forceinline__ void klt_load8(const uint8_t *p, uint32_t &lo, uint32_t &hi)
{
#ifndef YGZ_KLT_UNALIGNED_LOADS
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    klt_gptr q = (klt_gptr)(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
#else
    const ygz_u32x2 v = *(ygz_gptr2u)reinterpret_cast<uintptr_t>(p);
    lo = v.x; hi = v.y;
#endif
}
// all factors fit 24 bits (pixels 8 bit, derivatives 14 bit, weights 15 bit, differences 14 bit): full-rate v_mad_*24
// instead of the quarter-rate 32-bit multiply
#define KLT_MAD(a, b, c) ((int)__mul24((int)(a), (int)(b)) + (int)(c))
// the 4-tap fixed-point bilinear sums as two v_dot2c_i32_i16 (exact: |tap| < 2^13, weights <= 2^14):
//   pixels: (byte k, byte k+1) of an 8-byte row -> two zero-extended 16-bit halves with one v_perm_b32
//   derivatives: the x (low) or y (high) int16 halves of two adjacent Scharr dwords with one v_perm_b32
typedef short klt_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int klt_dot2(uint32_t a, uint32_t b, int acc)
{ return __builtin_amdgcn_sdot2(__builtin_bit_cast(klt_s2, a), __builtin_bit_cast(klt_s2, b), acc, false); }
#define KLT_PAIR(lo, hi, k) __builtin_amdgcn_perm((hi), (lo), 0x0c000c00u | ((uint32_t)((k) + 1) << 16) | (uint32_t)(k))
#define KLT_BIL9P(l0, h0, l1, h1, k) (klt_dot2(KLT_PAIR(l1, h1, k), wbot, klt_dot2(KLT_PAIR(l0, h0, k), wtop, 256)) >> 9)
// the same minus a patch value, with the subtraction folded into the accumulator: c = 256 - (I << 9), and
// (S + 256 - 512 I) >> 9 == ((S + 256) >> 9) - I for the arithmetic shift
// (the compiler only selects the two-address v_dot2c, which would need a copy of c first: the three-address VOP3P form by hand)
__device__ __forceinline__ int klt_dot2_keep(uint32_t a, uint32_t b, int acc)
{ int d; asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc)); return d; }
__device__ __forceinline__ int klt_dot2_sacc(uint32_t a, uint32_t b, int acc /* wave-uniform */)
{ int d; asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(acc)); return d; }
#define KLT_DIFF9P(l0, h0, l1, h1, k, c) (klt_dot2(KLT_PAIR(l1, h1, k), wbot, klt_dot2_keep(KLT_PAIR(l0, h0, k), wtop, (c))) >> 9)
// the raw sum S + c of KLT_DIFF9P for two pixels -> (diff_0, diff_1) as int16 pair: |S + c| < 2^23, so bytes 1-2 of each are
// (S + c) >> 8 as int16; one packed arithmetic shift finishes the >> 9 of both (2 instructions per pair instead of 3)
#define KLT_SUM9P(l0, h0, l1, h1, k, c) klt_dot2(KLT_PAIR(l1, h1, k), wbot, klt_dot2_keep(KLT_PAIR(l0, h0, k), wtop, (c)))
__device__ __forceinline__ uint32_t klt_diff_pair(int x0, int x1)
{
    const klt_s2 h = __builtin_bit_cast(klt_s2, __builtin_amdgcn_perm((uint32_t)x1, (uint32_t)x0, 0x06050201u));
    return __builtin_bit_cast(uint32_t, (klt_s2)(h >> (klt_s2){ 1, 1 }));
}
#define KLT_DXP(a, b) __builtin_amdgcn_perm((b), (a), 0x05040100u)
#define KLT_DYP(a, b) __builtin_amdgcn_perm((b), (a), 0x07060302u)
#define KLT_WEIGHTS(a, b)                                                                        \
    iw00 = cv_round_f(__fmul_rn(__fmul_rn(__fsub_rn(1.f, a), __fsub_rn(1.f, b)), 16384.f));     \
    iw01 = cv_round_f(__fmul_rn(__fmul_rn(a, __fsub_rn(1.f, b)), 16384.f));                     \
    iw10 = cv_round_f(__fmul_rn(__fmul_rn(__fsub_rn(1.f, a), b), 16384.f));                     \
    iw11 = 16384 - iw00 - iw01 - iw10;                                                          \
    wtop = (uint32_t)iw00 | ((uint32_t)iw01 << 16); wbot = (uint32_t)iw10 | ((uint32_t)iw11 << 16)

// FULL: the 21-wide window (every active lane owns exactly 7 pixels) -- the per-pixel guards fold away
template <bool FULL>
#define KLT_WPB 4          // wavefronts (points) per workgroup: points finish after very different iteration counts, small groups retire early
__global__ __launch_bounds__(64 * KLT_WPB) void k_klt(KltArgs A)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int bx, pair;
    if (!ygz_xcd_remap(A.n_pairs, bx, pair)) return;
    const int pi = bx * KLT_WPB + wv;
    if (pi >= A.trk_n[pair]) return;               // wave-uniform
    const size_t p = (size_t)pair * A.cells + pi;
    const size_t ref_slot = (size_t)A.pair_t[pair], cur_slot = (size_t)A.pair_q[pair];
    const int win = A.win;
    const float half = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const float ppx = (float)A.trk_px[2 * p], ppy = (float)A.trk_px[2 * p + 1];   // cv::Point2f(fea->_pixel) (Tracker.cpp:85)
    float outx = A.use_initial_flow ? A.next_pts[2 * p] : ppx;
    float outy = A.use_initial_flow ? A.next_pts[2 * p + 1] : ppy;
    bool status = true;
    float errv = 0.f;
    long long t_start = A.dbg ? clock64() : 0; int n_it = 0;
    const int row = lane / 3, x0 = 7 * (lane - 3 * row);
    const bool act = row < win && x0 < win;
    const int npx = FULL ? 7 : (act ? min(7, win - x0) : 0);    // pixels of this lane (7 for the 21-wide window)

    for (int level = A.max_level; level >= 0; --level) {
        const int w = A.w[level], h = A.h[level];
        const int pw = KLT_PW(w);
        const size_t psz = (size_t)pw * (h + 2 * KLT_B), org = (size_t)KLT_B * pw + KLT_B;
        const uint8_t *I = A.pad[level] + ref_slot * psz + org, *J = A.pad[level] + cur_slot * psz + org;
        klt_gptr D = (klt_gptr)(reinterpret_cast<const uint32_t *>(A.deriv[level]) + ref_slot * psz + org);
        const float s = __int_as_float((127 - level) << 23);       // (float)(1. / (1 << level)), exact
        float prevx = __fmul_rn(ppx, s), prevy = __fmul_rn(ppy, s);
        float nx, ny;
        if (level == A.max_level) { nx = __fmul_rn(outx, s); ny = __fmul_rn(outy, s); }
        else { nx = __fmul_rn(outx, 2.f); ny = __fmul_rn(outy, 2.f); }
        outx = nx; outy = ny;
        prevx = __fsub_rn(prevx, half); prevy = __fsub_rn(prevy, half);
        const int ipx = (int)floorf(prevx), ipy = (int)floorf(prevy);
        if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
            if (level == 0) { status = false; errv = 0.f; }
            continue;
        }
        float a = __fsub_rn(prevx, (float)ipx), b = __fsub_rn(prevy, (float)ipy);
        int iw00, iw01, iw10, iw11;
        uint32_t wtop, wbot;
        KLT_WEIGHTS(a, b);
        float sA11 = 0.f, sA12 = 0.f, sA22 = 0.f;
        // the lane's 7 patch values (int16 image <<5, int16 derivatives) stay in registers for the whole level
        int iI[7], iDx[7], iDy[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) { iI[k] = 0; iDx[k] = 0; iDy[k] = 0; }
        if (act) {
            const int o = __mul24(ipy + row, pw) + ipx + x0;          // inside the framed buffer for every admissible window
            uint32_t l0, h0, l1, h1;
            klt_load8(I + o, l0, h0); klt_load8(I + o + pw, l1, h1);
            uint32_t d0[8], d1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { d0[k] = D[o + k]; d1[k] = D[o + pw + k]; }
            int q11 = 0, q12 = 0, q22 = 0;                      // exact per-lane sums: Scharr of u8 is < 2^12, 7 products < 2^27
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                if (FULL || k < npx) {
                    const int ival = KLT_BIL9P(l0, h0, l1, h1, k);
                    const int ixval = klt_dot2(KLT_DXP(d1[k], d1[k + 1]), wbot, klt_dot2(KLT_DXP(d0[k], d0[k + 1]), wtop, 8192)) >> 14;
                    const int iyval = klt_dot2(KLT_DYP(d1[k], d1[k + 1]), wbot, klt_dot2(KLT_DYP(d0[k], d0[k + 1]), wtop, 8192)) >> 14;
                    iI[k] = (int)(int16_t)ival; iDx[k] = (int)(int16_t)ixval; iDy[k] = (int)(int16_t)iyval;
                    q11 += __mul24(iDx[k], iDx[k]); q12 += __mul24(iDx[k], iDy[k]); q22 += __mul24(iDy[k], iDy[k]);
                }
            }
            sA11 = (float)q11; sA12 = (float)q12; sA22 = (float)q22;
        }
        const float A11 = __fmul_rn(wave_sum_f(sA11), FLT_SCALE), A12 = __fmul_rn(wave_sum_f(sA12), FLT_SCALE),
                    A22 = __fmul_rn(wave_sum_f(sA22), FLT_SCALE);
        float Dd = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
        const float dif = __fsub_rn(A11, A22);
        const float minEig = __fdiv_rn(__fsub_rn(__fadd_rn(A22, A11),
                                                 ygz_sqrtf_cr(__fadd_rn(__fmul_rn(dif, dif), __fmul_rn(__fmul_rn(4.f, A12), A12)))),
                                       (float)(2 * win * win));
        if (minEig < A.min_eig_thr || Dd < 1.192092896e-07f) {
            if (level == 0) status = false;
            continue;
        }
        Dd = __fdiv_rn(1.f, Dd);
        nx = __fsub_rn(nx, half); ny = __fsub_rn(ny, half);
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < A.max_count; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= w || iny < -win || iny >= h) {
                if (level == 0) status = false;
                break;
            }
            ++n_it;
            a = __fsub_rn(nx, (float)inx); b = __fsub_rn(ny, (float)iny);
            KLT_WEIGHTS(a, b);
            float sb1 = 0.f, sb2 = 0.f;
            if (act) {
                const int o = __mul24(iny + row, pw) + inx + x0;
                uint32_t l0, h0, l1, h1;
                klt_load8(J + o, l0, h0); klt_load8(J + o + pw, l1, h1);
                // the lane's 7 terms are summed exactly in int32 (|diff*I| < 2^26), converted once; pixels beyond the
                // window (k >= npx) carry iDx = iDy = 0, so whatever diff they form is multiplied by 0
                int a1 = 0, a2 = 0;
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const int diff = KLT_BIL9P(l0, h0, l1, h1, k) - iI[k];
                    a1 = KLT_MAD(diff, iDx[k], a1);
                    a2 = KLT_MAD(diff, iDy[k], a2);
                }
                sb1 = (float)a1; sb2 = (float)a2;
            }
            const float b1 = __fmul_rn(wave_sum_f(sb1), FLT_SCALE), b2 = __fmul_rn(wave_sum_f(sb2), FLT_SCALE);
            const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, b2), __fmul_rn(A22, b1)), Dd);
            const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, b1), __fmul_rn(A11, b2)), Dd);
            nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
            outx = __fadd_rn(nx, half); outy = __fadd_rn(ny, half);
            if ((double)dx * (double)dx + (double)dy * (double)dy <= A.epsilon) break;
            if (j > 0 && (double)fabsf(__fadd_rn(dx, pdx)) < 0.01 && (double)fabsf(__fadd_rn(dy, pdy)) < 0.01) {
                outx = __fsub_rn(outx, __fmul_rn(dx, 0.5f)); outy = __fsub_rn(outy, __fmul_rn(dy, 0.5f));
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0) {
            const float qx = __fsub_rn(outx, half), qy = __fsub_rn(outy, half);
            const int inx = (int)floorf(qx), iny = (int)floorf(qy);
            if (inx < -win || inx >= w || iny < -win || iny >= h) { status = false; continue; }
            const float aa = __fsub_rn(qx, (float)inx), bb = __fsub_rn(qy, (float)iny);
            KLT_WEIGHTS(aa, bb);
            float se = 0.f;
            if (act) {
                const int o = __mul24(iny + row, pw) + inx + x0;
                uint32_t l0, h0, l1, h1;
                klt_load8(J + o, l0, h0); klt_load8(J + o + pw, l1, h1);
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    if (FULL || k < npx) {
                        const int diff = KLT_BIL9P(l0, h0, l1, h1, k) - iI[k];
                        se = __fadd_rn(se, fabsf((float)diff));
                    }
                }
            }
            errv = __fdiv_rn(__fmul_rn(wave_sum_f(se), 1.f), (float)(32 * win * win));
        }
    }
    if (lane == 0) {
        A.next_pts[2 * p] = outx; A.next_pts[2 * p + 1] = outy;
        A.status[p] = (uint8_t)status; A.err[p] = errv;
        if (A.dbg) { A.dbg[4 * p] = clock64() - t_start; A.dbg[4 * p + 3] = n_it; }
    }
}

// ---- three points per wavefront (the default 21 x 21 window).  In k_klt<> two thirds of every iteration are per-point
// bookkeeping (weights, window address, bounds, the 2x2 solve, convergence tests) executed by all 64 lanes for ONE point.
// Here a point owns 21 lanes (7 row-triples x 3 column-thirds, 3 rows x 7 pixels per lane), so the same bookkeeping
// instructions serve three points, consecutive window rows share their image row (4 loads and 28 byte-pair permutes for 3
// rows instead of 6 and 42), and the lanes of a point reduce with a fixed-order 21-lane tree.  Points of a wavefront iterate
// until the slowest has converged; finished ones idle (their lanes are masked).  Integer terms are those of k_klt<> and the
// oracle; float sums differ in order only.
__device__ __forceinline__ float klt_seg21_sum(float v, int lane, int q, int seg)
{
    float t;
    t = __shfl(v, lane + 16 < 63 ? lane + 16 : 63); if (q < 5) v = __fadd_rn(v, t);
    t = __shfl(v, lane + 8 < 63 ? lane + 8 : 63);   if (q < 8) v = __fadd_rn(v, t);
    t = __shfl(v, lane + 4 < 63 ? lane + 4 : 63);   if (q < 4) v = __fadd_rn(v, t);
    t = __shfl(v, lane + 2 < 63 ? lane + 2 : 63);   if (q < 2) v = __fadd_rn(v, t);
    t = __shfl(v, lane + 1 < 63 ? lane + 1 : 63);   if (q < 1) v = __fadd_rn(v, t);
    return __shfl(v, seg);
}

// (Round 4 also measured fetching only three of a lane's four window rows and taking the fourth from the lane that owns the next row triple
// through the LDS crossbar -- two ds_bpermute instead of a gather: bit-identical and SLOWER, 2.68 against 2.52 ms per 512 pairs; the dependent
// crossbar round trip in front of the arithmetic costs more than the gather it saves.)
__global__ __launch_bounds__(256) void k_klt3(KltArgs A)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int bx, pair;
    if (!ygz_xcd_remap(A.n_pairs, bx, pair)) return;
    const int n = A.trk_n[pair];
    const int first = (bx * 4 + wv) * 3;
    if (first >= n) return;                                  // wave-uniform
    const int sub = lane / 21, q = lane - 21 * sub, rt = q / 3, x0 = 7 * (q - 3 * rt), row0 = 3 * rt, seg = sub < 3 ? 21 * sub : 63;
    const int pi = first + sub;
    const bool live = sub < 3 && pi < n;                     // lane 63 and the lanes of points past the end idle
    const size_t p = (size_t)pair * A.cells + (live ? pi : first);
    const size_t ref_slot = (size_t)A.pair_t[pair], cur_slot = (size_t)A.pair_q[pair];
    const int win = KLT_MAXWIN;
    const float half = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const float ppx = (float)A.trk_px[2 * p], ppy = (float)A.trk_px[2 * p + 1];   // cv::Point2f(fea->_pixel) (Tracker.cpp:85)
    float outx = A.use_initial_flow ? A.next_pts[2 * p] : ppx;
    float outy = A.use_initial_flow ? A.next_pts[2 * p + 1] : ppy;
    bool status = true;
    float errv = 0.f;

    for (int level = A.max_level; level >= 0; --level) {
        const int w = A.w[level], h = A.h[level];
        const int pw = KLT_PW(w);
        const size_t psz = (size_t)pw * (h + 2 * KLT_B), org = (size_t)KLT_B * pw + KLT_B;
        const uint8_t *I = A.pad[level] + ref_slot * psz + org, *J = A.pad[level] + cur_slot * psz + org;
        klt_gptr D = (klt_gptr)(reinterpret_cast<const uint32_t *>(A.deriv[level]) + ref_slot * psz + org);
        const float s = __int_as_float((127 - level) << 23);       // (float)(1. / (1 << level)), exact
        float prevx = __fmul_rn(ppx, s), prevy = __fmul_rn(ppy, s);
        float nx, ny;
        if (level == A.max_level) { nx = __fmul_rn(outx, s); ny = __fmul_rn(outy, s); }
        else { nx = __fmul_rn(outx, 2.f); ny = __fmul_rn(outy, 2.f); }
        outx = nx; outy = ny;
        prevx = __fsub_rn(prevx, half); prevy = __fsub_rn(prevy, half);
        const int ipx = (int)floorf(prevx), ipy = (int)floorf(prevy);
        const bool in_lv = live && !(ipx < -win || ipx >= w || ipy < -win || ipy >= h);
        if (live && !in_lv && level == 0) { status = false; errv = 0.f; }
        float a = __fsub_rn(prevx, (float)ipx), b = __fsub_rn(prevy, (float)ipy);
        int iw00, iw01, iw10, iw11;
        uint32_t wtop, wbot;
        KLT_WEIGHTS(a, b);
        // the lane's 3 x 7 patch values stay in registers for the whole level: image << 5, and the derivatives of two
        // consecutive pixels per register (dx_k | dx_k+1 << 16, same for dy) -- the layout v_dot2 wants for the mismatch sums
        int cI[21]; uint32_t pDx[11], pDy[11];                      // cI = 256 - (patch value << 9), see KLT_DIFF9P; only read where in_lv
        float sA11 = 0.f, sA12 = 0.f, sA22 = 0.f;
        if (in_lv) {
            const int o = __mul24(ipy + row0, pw) + ipx + x0;
            uint32_t il[4], ih[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) klt_load8(I + o + r * pw, il[r], ih[r]);
            // the (short) casts of the reference are value-preserving (0 <= ival <= 255 << 5, |Scharr| <= 16 * 255): the raw sums are
            // packed two pixels per register with one v_perm, and the A-matrix sums run on the packed pairs (11 v_dot2 each
            // instead of 21 multiply-adds; exact: 21 products < 2^29)
            int prev_x = 0, prev_y = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                uint32_t d0[8], d1[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) { d0[kk] = D[o + r * pw + kk]; d1[kk] = D[o + (r + 1) * pw + kk]; }
#pragma unroll
                for (int kk = 0; kk < 7; ++kk) {
                    const int ival = klt_dot2(KLT_PAIR(il[r + 1], ih[r + 1], kk), wbot, klt_dot2_sacc(KLT_PAIR(il[r], ih[r], kk), wtop, 256)) >> 9;
                    const int ixval = klt_dot2(KLT_DXP(d1[kk], d1[kk + 1]), wbot, klt_dot2_sacc(KLT_DXP(d0[kk], d0[kk + 1]), wtop, 8192)) >> 14;
                    const int iyval = klt_dot2(KLT_DYP(d1[kk], d1[kk + 1]), wbot, klt_dot2_sacc(KLT_DYP(d0[kk], d0[kk + 1]), wtop, 8192)) >> 14;
                    const int li = 7 * r + kk;
                    cI[li] = 256 - (ival << 9);
                    if (li & 1) { pDx[li >> 1] = KLT_DXP((uint32_t)prev_x, (uint32_t)ixval); pDy[li >> 1] = KLT_DXP((uint32_t)prev_y, (uint32_t)iyval); }
                    else if (li == 20) { pDx[10] = (uint32_t)ixval & 0xffffu; pDy[10] = (uint32_t)iyval & 0xffffu; }
                    else { prev_x = ixval; prev_y = iyval; }
                }
            }
            int q11 = 0, q12 = 0, q22 = 0;
#pragma unroll
            for (int kk = 0; kk < 11; ++kk) { q11 = klt_dot2(pDx[kk], pDx[kk], q11); q12 = klt_dot2(pDx[kk], pDy[kk], q12); q22 = klt_dot2(pDy[kk], pDy[kk], q22); }
            sA11 = (float)q11; sA12 = (float)q12; sA22 = (float)q22;
        }
        const float A11 = __fmul_rn(klt_seg21_sum(sA11, lane, q, seg), FLT_SCALE), A12 = __fmul_rn(klt_seg21_sum(sA12, lane, q, seg), FLT_SCALE),
                    A22 = __fmul_rn(klt_seg21_sum(sA22, lane, q, seg), FLT_SCALE);
        float Dd = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
        const float dif = __fsub_rn(A11, A22);
        const float minEig = __fdiv_rn(__fsub_rn(__fadd_rn(A22, A11),
                                                 ygz_sqrtf_cr(__fadd_rn(__fmul_rn(dif, dif), __fmul_rn(__fmul_rn(4.f, A12), A12)))),
                                       (float)(2 * win * win));
        const bool it_ok = in_lv && !(minEig < A.min_eig_thr || Dd < 1.192092896e-07f);
        if (in_lv && !it_ok && level == 0) status = false;
        Dd = __fdiv_rn(1.f, Dd);
        nx = it_ok ? __fsub_rn(nx, half) : nx; ny = it_ok ? __fsub_rn(ny, half) : ny;
        float pdx = 0.f, pdy = 0.f;
        bool done = !it_ok;
        for (int j = 0; j < A.max_count; ++j) {
            if (__ballot(!done) == 0ull) break;
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (!done && (inx < -win || inx >= w || iny < -win || iny >= h)) {
                if (level == 0) status = false;
                done = true;
            }
            const bool act = !done;
            a = __fsub_rn(nx, (float)inx); b = __fsub_rn(ny, (float)iny);
            KLT_WEIGHTS(a, b);
            float sb1 = 0.f, sb2 = 0.f;
            if (act) {
                const int o = __mul24(iny + row0, pw) + inx + x0;
                uint32_t jl[4], jh[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) klt_load8(J + o + r * pw, jl[r], jh[r]);
                int a1 = 0, a2 = 0;                                  // 21 terms, |diff * I| < 2^26: exact in int32
                int df[22];                                          // S + c, the >> 9 happens in klt_diff_pair
                df[21] = 0;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int kk = 0; kk < 7; ++kk) df[7 * r + kk] = KLT_SUM9P(jl[r], jh[r], jl[r + 1], jh[r + 1], kk, cI[7 * r + kk]);
                }
#pragma unroll
                for (int kk = 0; kk < 11; ++kk) {                    // (diff_2k, diff_2k+1) . (dx_2k, dx_2k+1): |diff| < 2^14 fits int16
                    const uint32_t dp = klt_diff_pair(df[2 * kk], df[2 * kk + 1]);
                    a1 = klt_dot2(dp, pDx[kk], a1);
                    a2 = klt_dot2(dp, pDy[kk], a2);
                }
                sb1 = (float)a1; sb2 = (float)a2;
            }
            const float b1 = __fmul_rn(klt_seg21_sum(sb1, lane, q, seg), FLT_SCALE), b2 = __fmul_rn(klt_seg21_sum(sb2, lane, q, seg), FLT_SCALE);
            if (act) {
                const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, b2), __fmul_rn(A22, b1)), Dd);
                const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, b1), __fmul_rn(A11, b2)), Dd);
                nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
                outx = __fadd_rn(nx, half); outy = __fadd_rn(ny, half);
                if ((double)dx * (double)dx + (double)dy * (double)dy <= A.epsilon) done = true;
                else if (j > 0 && (double)fabsf(__fadd_rn(dx, pdx)) < 0.01 && (double)fabsf(__fadd_rn(dy, pdy)) < 0.01) {
                    outx = __fsub_rn(outx, __fmul_rn(dx, 0.5f)); outy = __fsub_rn(outy, __fmul_rn(dy, 0.5f));
                    done = true;
                }
                pdx = dx; pdy = dy;
            }
        }
        if (level == 0) {
            bool want = it_ok && status;
            const float qx = __fsub_rn(outx, half), qy = __fsub_rn(outy, half);
            const int inx = (int)floorf(qx), iny = (int)floorf(qy);
            if (want && (inx < -win || inx >= w || iny < -win || iny >= h)) { status = false; want = false; }
            const float aa = __fsub_rn(qx, (float)inx), bb = __fsub_rn(qy, (float)iny);
            KLT_WEIGHTS(aa, bb);
            float se = 0.f;
            if (want) {
                const int o = __mul24(iny + row0, pw) + inx + x0;
                uint32_t jl[4], jh[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) klt_load8(J + o + r * pw, jl[r], jh[r]);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int kk = 0; kk < 7; ++kk) {
                        const int diff = KLT_DIFF9P(jl[r], jh[r], jl[r + 1], jh[r + 1], kk, cI[7 * r + kk]);
                        se = __fadd_rn(se, fabsf((float)diff));
                    }
                }
            }
            const float tot = klt_seg21_sum(se, lane, q, seg);
            if (want) errv = __fdiv_rn(__fmul_rn(tot, 1.f), (float)(32 * win * win));
        }
    }
    if (live && q == 0) {
        A.next_pts[2 * p] = outx; A.next_pts[2 * p + 1] = outy;
        A.status[p] = (uint8_t)status; A.err[p] = errv;
    }
}

static int launch_klt_impl(ygz_hip_ctx *ctx, int n_pairs, const ygz_klt_params *prm, bool prep_only);
int ygz_launch_klt(ygz_hip_ctx *ctx, int n_pairs, const ygz_klt_params *prm) { return launch_klt_impl(ctx, n_pairs, prm, false); }
// the tracker's working images (reflect-framed copies + Scharr images of the distinct slots of the pair table) ahead of the LK launch:
// they depend only on the pyramids, so a pipeline can build them before it forks its side-stream stages
int ygz_klt_prepare_early(ygz_hip_ctx *ctx)
{
    ygz_klt_params prm;
    ygz_hip_default_klt_params(&prm);
    int rc = launch_klt_impl(ctx, ctx->n_pairs, &prm, true);
    if (rc == YGZ_OK) {
        // the levels that were prepared (default tracker): a later call with other parameters may need more
        int max_level = prm.max_level, sw = ctx->lw[0], sh = ctx->lh[0];
        for (int level = 0; level <= prm.max_level; ++level) { sw = (sw + 1) / 2; sh = (sh + 1) / 2; if (sw <= prm.win || sh <= prm.win) { max_level = level; break; } }
        ctx->klt_prep_valid = true; ctx->klt_prep_levels = max_level + 1;
    }
    return rc;
}
static int launch_klt_impl(ygz_hip_ctx *ctx, int n_pairs, const ygz_klt_params *prm, bool prep_only)
{
    if (prm->win < 3 || prm->win > KLT_MAXWIN || prm->max_level < 0 || prm->max_level >= YGZ_MAX_LEVELS) return YGZ_E_INVALID;
    // buildOpticalFlowPyramid: stop when the next level would not exceed the window
    int max_level = prm->max_level, sw = ctx->lw[0], sh = ctx->lh[0];
    for (int level = 0; level <= prm->max_level; ++level) {
        sw = (sw + 1) / 2; sh = (sh + 1) / 2;
        if (sw <= prm->win || sh <= prm->win) { max_level = level; break; }
    }
    if (max_level + 1 > ctx->n_levels_alloc) {      // non-default tracker: more levels than were built at upload time
        int rc = ygz_ensure_levels(ctx, max_level + 1);
        if (rc != YGZ_OK) return rc;
        for (int s = 0; s < ctx->prm.max_frames; ++s)
            if (ctx->pyr_valid[s] && (rc = ygz_launch_gray_pyramid(ctx, s, 1, 0, max_level + 1)) != YGZ_OK) return rc;
    }
    KltArgs A;
    for (int L = 0; L < YGZ_MAX_LEVELS; ++L) { A.pad[L] = nullptr; A.deriv[L] = nullptr; A.w[L] = A.h[L] = 0; }
    for (int L = 0; L <= max_level; ++L) {
        const int w = ctx->lw[L], h = ctx->lh[L], pw = KLT_PW(w), ph = h + 2 * KLT_B;
        const size_t psz = (size_t)pw * ph;
        if (!ctx->deriv[L]) {
            YGZ_HIPCHK(ctx, hipMalloc((void **)&ctx->deriv[L], (size_t)ctx->prm.max_frames * psz * 4 + 64));
            YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->deriv[L], 0, (size_t)ctx->prm.max_frames * psz * 4 + 64, ctx->stream));   // the zero frame
        }
        if (!ctx->klt_pad[L]) YGZ_HIPCHK(ctx, hipMalloc((void **)&ctx->klt_pad[L], (size_t)ctx->prm.max_frames * psz + 64));
        A.pad[L] = ctx->klt_pad[L]; A.deriv[L] = ctx->deriv[L]; A.w[L] = w; A.h[L] = h;
        if (ctx->klt_prep_valid && L < ctx->klt_prep_levels) continue;   // this level's working images were built ahead (ygz_klt_prepare_early)
        // the framed copy of a level is normally written by the pyramid kernels (image.hip); k_klt_pad only runs for slots whose pyramid
        // was built before the tracker's buffers existed (first call) or without the fused stores
        bool have_pad = (int)ctx->klt_slots_host.size() == ctx->n_klt_slots;
        for (int i = 0; i < ctx->n_klt_slots && have_pad; ++i) if (ctx->pad_levels[ctx->klt_slots_host[i]] <= L) have_pad = false;
        if (!have_pad)
            YGZ_LAUNCH(ctx, KID_KLT_PAD, k_klt_pad, dim3(ygz_div_up(ygz_div_up(pw, 16) * ph, 256), 1, ygz_round_up8(ctx->n_klt_slots)), dim3(256),
                       ctx->lvl[L], ctx->klt_pad[L], ctx->klt_slots, w, h, ctx->n_klt_slots);
        YGZ_LAUNCH(ctx, KID_SCHARR, k_scharr, dim3(ygz_div_up(ygz_div_up(w, 4) * ygz_div_up(h, 4), 256), 1, ygz_round_up8(ctx->n_klt_refs)), dim3(256),
                   ctx->klt_pad[L], ctx->deriv[L], ctx->klt_slots + ctx->n_klt_slots, w, h, ctx->n_klt_refs);
    }
    if ((int)ctx->klt_slots_host.size() == ctx->n_klt_slots)          // every level 0 .. max_level of these slots now has its framed copy
        for (int i = 0; i < ctx->n_klt_slots; ++i) { uint8_t &pl = ctx->pad_levels[ctx->klt_slots_host[i]]; if (pl < max_level + 1) pl = (uint8_t)(max_level + 1); }
    ctx->klt_prep_valid = false;
    if (prep_only) return YGZ_OK;
    if (ctx->klt_prep_pending) { YGZ_HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prep, 0)); ctx->klt_prep_pending = false; }
    A.max_level = max_level; A.win = prm->win; A.cells = ctx->cells; A.n_pairs = n_pairs;
    A.max_count = prm->max_iter < 0 ? 0 : (prm->max_iter > 100 ? 100 : prm->max_iter);
    const double eps = prm->eps < 0 ? 0 : (prm->eps > 10 ? 10 : prm->eps);
    A.epsilon = eps * eps;
    A.min_eig_thr = (float)prm->min_eig_threshold;
    A.use_initial_flow = prm->use_initial_flow;
    A.pair_q = ctx->pair_q; A.pair_t = ctx->pair_t; A.trk_n = ctx->trk_n; A.trk_px = ctx->trk_px;
    A.next_pts = ctx->klt_pts; A.status = ctx->klt_status; A.err = ctx->klt_err;
    A.dbg = nullptr;
    // 21 x 21 (the reference's window, Tracker.h:25): three points per wavefront; any other window size: one point per wavefront
    if (A.win == KLT_MAXWIN) YGZ_LAUNCH(ctx, KID_KLT, k_klt3, dim3(ygz_div_up(ctx->cells, 12), ygz_round_up8(n_pairs)), dim3(256), A);
    else YGZ_LAUNCH(ctx, KID_KLT, k_klt<false>, dim3(ygz_div_up(ctx->cells, KLT_WPB), ygz_round_up8(n_pairs)), dim3(64 * KLT_WPB), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

extern "C" {

void ygz_hip_default_klt_params(ygz_klt_params *p)
{
    p->win = 21; p->max_level = 4; p->max_iter = 30;      // Tracker.h:25-26, Tracker.cpp:97
    p->eps = 0.001; p->min_eig_threshold = 1e-4;          // Tracker.h:27, OpenCV default
    p->use_initial_flow = 1;
}

}  // extern "C"

// Tracker::TrackKLT's survivor rule (src/Algorithm/Tracker.cpp:100-112): a track is kept iff status != 0 and the new position is
// InFrame(pt, border) -- evaluated where the results are, one flag per point
__global__ __launch_bounds__(256) void k_klt_keep(const float *__restrict__ pts, const uint8_t *__restrict__ status, int n, int w, int h,
                                                  int border, uint8_t *__restrict__ keep, int32_t *__restrict__ n_keep)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool k = false;
    if (i < n) {
        const float x = pts[2 * i], y = pts[2 * i + 1];
        k = status[i] != 0 && x >= border && x < w - border && y >= border && y < h - border;      // Frame::InFrame(cv::Point2f, boarder), Frame.h:60-65
        keep[i] = (uint8_t)k;
    }
    const int c = __popcll(__ballot(k));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(n_keep, c);
}

// single pair, host arrays: fills track set 0 and runs the batched kernel with one pair -- in pieces of `cells` points (the track set's size; points
// are independent, so any n is served: cv::calcOpticalFlowPyrLK has no limit either)
static int klt_track_host(ygz_hip_ctx *ctx, int prev_slot, int cur_slot, const float *prev_pts, float *next_pts, int n,
                          const ygz_klt_params *prm, uint8_t *status, float *err, int border, uint8_t *keep, int *n_keep)
{
    if (!ctx || !prm || n < 0 || prev_slot < 0 || prev_slot >= ctx->prm.max_frames || cur_slot < 0 || cur_slot >= ctx->prm.max_frames)
        return YGZ_E_INVALID;
    if (n_keep) *n_keep = 0;
    if (n == 0) return YGZ_OK;
    if (!prev_pts || !next_pts || !status) return YGZ_E_INVALID;
    if (!ctx->pyr_valid[prev_slot] || !ctx->pyr_valid[cur_slot]) return YGZ_E_STATE;
    const double I7[7] = { 0, 0, 0, 1, 0, 0, 0 };
    int rc = ygz_track_set_pairs(ctx, &cur_slot, &prev_slot, I7, I7, 1);
    if (rc != YGZ_OK) return rc;
    std::vector<double> px;
    for (int base = 0; base < n; base += ctx->cells) {
        const int m = n - base < ctx->cells ? n - base : ctx->cells;
        const size_t N = (size_t)m;
        px.resize(N * 2);
        for (int i = 0; i < 2 * m; ++i) px[i] = (double)prev_pts[2 * (size_t)base + i];
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->trk_px, px.data(), N * 16, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->klt_pts, next_pts + 2 * (size_t)base, N * 8, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->trk_n, &m, 4, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if ((rc = ygz_launch_klt(ctx, 1, prm)) != YGZ_OK) return rc;
        int kept = 0;
        if (keep) {
            uint8_t *d_keep = nullptr;
            if ((rc = ygz_scratch(ctx, SCR_KLT_OUT, N + 64, (void **)&d_keep)) != YGZ_OK) return rc;
            int32_t *d_cnt = reinterpret_cast<int32_t *>(d_keep + ((N + 15) & ~(size_t)15));
            YGZ_HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, 4, ctx->stream));
            YGZ_LAUNCH(ctx, KID_TRACK_AUX, k_klt_keep, dim3(ygz_div_up(m, 256)), dim3(256), ctx->klt_pts, ctx->klt_status, m, ctx->lw[0], ctx->lh[0],
                       border, d_keep, d_cnt);
            YGZ_HIPCHK(ctx, hipGetLastError());
            YGZ_HIPCHK(ctx, hipMemcpyAsync(keep + base, d_keep, N, hipMemcpyDeviceToHost, ctx->stream));
            if (n_keep) YGZ_HIPCHK(ctx, hipMemcpyAsync(&kept, d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        YGZ_HIPCHK(ctx, hipMemcpyAsync(next_pts + 2 * (size_t)base, ctx->klt_pts, N * 8, hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(status + base, ctx->klt_status, N, hipMemcpyDeviceToHost, ctx->stream));
        if (err) YGZ_HIPCHK(ctx, hipMemcpyAsync(err + base, ctx->klt_err, N * 4, hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (n_keep) *n_keep += kept;
    }
    return YGZ_OK;
}

extern "C" {

int ygz_hip_klt_track(ygz_hip_ctx *ctx, int prev_slot, int cur_slot, const float *prev_pts, float *next_pts, int n,
                      const ygz_klt_params *prm, uint8_t *status, float *err)
{
    YgzDeviceGuard dg_(ctx);
    return klt_track_host(ctx, prev_slot, cur_slot, prev_pts, next_pts, n, prm, status, err, 0, nullptr, nullptr);
}

int ygz_hip_klt_track_filtered(ygz_hip_ctx *ctx, int prev_slot, int cur_slot, const float *prev_pts, float *next_pts, int n,
                               const ygz_klt_params *prm, int border, uint8_t *status, float *err, uint8_t *keep, int *n_keep)
{
    YgzDeviceGuard dg_(ctx);
    if (!keep && n > 0) return YGZ_E_INVALID;
    return klt_track_host(ctx, prev_slot, cur_slot, prev_pts, next_pts, n, prm, status, err, border, keep, n_keep);
}

}  // extern "C"
