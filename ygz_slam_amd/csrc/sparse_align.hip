// L3 -- SVO-style sparse image alignment, whole Gauss-Newton loop resident on the GPU.
// Replaces SparseImgAlign::run / precomputeReferencePatches / computeResiduals / solve / update
// (src/Algorithm/SparseImageAlign.cpp:21-238) and NLLSSolver::optimizeGaussNewton
// (include/ygz/Algorithm/NLSSolver_impl.hpp:15-89).
//
// One workgroup (256 or 512 lanes) per alignment problem; lane = feature (4x4 patch).  Per iteration:
//   residual pass: lanes warp their features, load the 5x5 current-image window row-wise, form 16 residuals, accumulate
//   J^T res (and the change of H when a feature enters / leaves) in FP64 registers, publish res^2 and an approximate prefix;
//   H / Jres are reduced in a fixed order (DPP wave sums, then the wavefront partials in order);
//   chi2 is the reference's FLOAT sum in feature/pixel order: it decides "error increased -> rollback"
//   (NLSSolver_impl.hpp:53-63), so it is reproduced bit for bit -- but evaluated through per-feature parity maps, a segmented
//   scan and a segment walk instead of n*16 dependent float adds (see sa_chain_term below and DESIGN.md);
//   one lane of wave 1 solves the 6x6 LDLT and forms T * exp(-x) meanwhile; lane 0 takes the accept / stop decision.
// No host round trip per iteration (the reference solves ~30 6x6 systems per frame pair).
#include "ygz_internal.h"
#include <algorithm>
#include <cstring>
#include "se3_dev.h"
#include "ldlt6.h"
#include <stdlib.h>
#include <stdio.h>

// 4 wavefronts (one per SIMD) with up to 256 VGPRs each: the kernel is latency-bound (serial chain / solve phases, dependent
// loads), 8 wavefronts per problem were not faster, and a 512-lane workgroup at 256 VGPRs owns the whole register file of its
// CU -- nothing else could run beside it.  At 256 lanes half of the file stays free and the VALU-bound stages launched on
// the other streams (LK, matcher) fill the idle issue slots: 3.87 -> 3.6 ms per step of the bench.
// (With few problems per launch -- fewer than half the CUs -- nothing competes for the registers and the 512-lane form is
// used: more lanes per problem shorten the residual pass.)

struct SaArgs {
    const uint8_t *lvl[YGZ_MAX_LEVELS];
    int w[YGZ_MAX_LEVELS], h[YGZ_MAX_LEVELS];
    float fx, fy, cx, cy;
    int cells, max_level, min_level, n_iter;
    const int32_t *pair_q, *pair_t, *trk_n;       // cur slot, ref slot, features per pair
    const double *pair_T;                         // [pairs][2][7]; T_ref used here
    const double *trk_px, *trk_depth; const uint8_t *trk_has_mp;     // [pairs][cells]
    uint8_t *work; size_t work_stride;            // per pair: Hf, (unused), dxy, prev_used (96 doubles/cell) | patch_cache | r2 | ctot | pre | fmap | pmap | visible | used
    double *out;                                  // [pairs][16]: pose 7 (in: initial cur->_TCW, out: result), n_meas, iters
    double *dbg;                                  // optional [pairs][8] phase cycle counters (profiling aid) or null
    int lcap;                                     // features whose per-iteration scratch (r2, maps, prefixes) lives in LDS; multiple of 64
    int pcap;                                     // second form: features whose reference patch lives in LDS; multiple of 64
    int n_pairs;                                  // second form: problems of the launch (a workgroup loops over blockIdx.x + k * gridDim.x)
    int prio;                                     // != 0: raise the wavefronts' issue priority (the kernel is latency-bound: one wavefront per SIMD)
    double *lin;                                  // optional [pairs][32]: what computeResiduals(model, linearize = true) leaves -- the float chi2 sum, n_meas, H (21), Jres (6) -- of the last pass
    int rel;                                      // != 0: out[0..6] is T_cur_from_ref itself, in and out (the solver's model, SparseImageAlign.cpp:37,48 left to the caller)
    double *out_host;                             // optional [pairs][16]: page-locked copy of `out`, written by the kernel itself (the single-frame call: no copy back)
};

#define wave_sum_d ygz_wave_sum_d

// ---- the reference's chi2: a FLOAT running sum c = fl(c + res*res) over all features / pixels in order
// (SparseImageAlign.cpp:213), whose value decides when the Gauss-Newton loop stops (NLSSolver_impl.hpp:53) -- so it has to
// be reproduced bit for bit, and as a dependent chain of n*16 float adds it used to be the critical path of the kernel.
// It is evaluated here EXACTLY but not sequentially.  While c stays inside one binade [2^E, 2^(E+1)) it is m * ulp with an
// integer m, and adding x >= 0 rounds to nearest-even on that grid: m += floor(x/ulp) + (frac > 1/2, or frac == 1/2 and the
// result would be odd).  The only sequential state that influences an increment is therefore the PARITY of m (ties).  A run
// of consecutive terms is summarised, for a predicted binade E, by two integers: the total increment for incoming parity 0
// and for incoming parity 1; two such maps compose associatively.  Each lane builds the map of one feature (16 terms,
// integer ALU only) for the binade predicted by an approximate prefix sum; a segmented DPP scan composes the maps of every run
// of features with the same predicted binade; wave 0 then walks the segments: if c really is in the segment's binade and the
// segment does not leave it, ONE integer add replaces 16 x (features of the segment) dependent float adds; if it does, a
// ballot finds the crossing feature (prefix increments are monotone), the features before it are taken at once and only its
// 16 terms are added in hardware floats (the first few features and ~log2(n) later ones per iteration).
// Either way the result is the reference's float, bit for bit.
__device__ __forceinline__ void sa_chain_term(uint32_t xb, int E, int &t0, int &t1, int &bad)
{   // branch-free: every lane of the wave runs the same ~20 integer instructions per term
    const uint32_t bex = xb >> 23, mant = xb & 0x7fffffu;
    const uint32_t mx = bex ? (mant | 0x800000u) : mant;        // x = mx * 2^(ex - 150), ulp of the binade = 2^(E - 150)
    const int shift = E - (int)max(bex, 1u);                     // denormals share the exponent of the smallest normal
    bad |= (shift <= 0) & (mx != 0);                             // x >= 2^E (or inf/nan): the sum leaves the binade
    const int sh = min(max(shift, 1), 31);                       // shift >= 25: x < ulp / 2 -> a = 0, rem = mx < half: no change
    const uint32_t a = mx >> sh, rem = mx & ((1u << sh) - 1u), half = 1u << (sh - 1);
    const int up = rem > half, tie = rem == half;
    t0 += (int)a + (up | (tie & ((t0 + (int)a) & 1)));
    t1 += (int)a + (up | (tie & ((1 + t1 + (int)a) & 1)));
}

// The per-iteration scratch of a feature lives in LDS (the first LC features; the reference patch of the first PC) or in the pair's global work
// arrays (features beyond: 720p and larger grids).  LC and PC are multiples of 64 and a wavefront always handles 64 consecutive features, so the
// choice is wave-uniform: a BRANCH with a pure LDS path (ds_read / ds_write) and a pure global path.  Round 4 selected the POINTER per lane
// ((f < LC ? lds : global)[..]), which makes every access a flat instruction -- a flat load from LDS takes several times a ds_read, and the
// chain walk, which waits for six of them per chunk of 64 features, spent 36 of the 85 thousand cycles of a Gauss-Newton iteration there.
// (explicit address spaces: with generic pointers the compiler folds the two paths back into one flat access of a selected pointer)
typedef float sa_v4f __attribute__((ext_vector_type(4)));
typedef int sa_v4i __attribute__((ext_vector_type(4)));
typedef int sa_v2i __attribute__((ext_vector_type(2)));
#define SA_LDS(T, p) ((__attribute__((address_space(3))) T *)(p))
#define SA_GLB(T, p) ((__attribute__((address_space(1))) T *)(p))
#define SA_F4(v) make_float4((v).x, (v).y, (v).z, (v).w)
#define SA_V4(v) ((sa_v4f){ (v).x, (v).y, (v).z, (v).w })
__device__ __forceinline__ void sa_ld4(bool in_lds, const float4 *lp, size_t ls, const float4 *gp, size_t gs, float4 v[4])
{
    sa_v4f t0, t1, t2, t3;
    if (in_lds) { const __attribute__((address_space(3))) sa_v4f *q = SA_LDS(const sa_v4f, lp); t0 = *q; t1 = *(q + ls); t2 = *(q + 2 * ls); t3 = *(q + 3 * ls); }
    else { const __attribute__((address_space(1))) sa_v4f *q = SA_GLB(const sa_v4f, gp); t0 = *q; t1 = *(q + gs); t2 = *(q + 2 * gs); t3 = *(q + 3 * gs); }
    v[0] = SA_F4(t0); v[1] = SA_F4(t1); v[2] = SA_F4(t2); v[3] = SA_F4(t3);
}
__device__ __forceinline__ void sa_st4(bool in_lds, float4 *lp, size_t ls, float4 *gp, size_t gs, const float4 v[4])
{
    const sa_v4f t0 = SA_V4(v[0]), t1 = SA_V4(v[1]), t2 = SA_V4(v[2]), t3 = SA_V4(v[3]);
    if (in_lds) { __attribute__((address_space(3))) sa_v4f *q = SA_LDS(sa_v4f, lp); *q = t0; *(q + ls) = t1; *(q + 2 * ls) = t2; *(q + 3 * ls) = t3; }
    else { __attribute__((address_space(1))) sa_v4f *q = SA_GLB(sa_v4f, gp); *q = t0; *(q + gs) = t1; *(q + 2 * gs) = t2; *(q + 3 * gs) = t3; }
}
#define SA_R2_LD(in_, f_, v_) sa_ld4((in_), l_r2 + (f_), (size_t)LC, reinterpret_cast<const float4 *>(r2) + (f_), (size_t)A.cells, (v_))
#define SA_R2_ST(in_, f_, v_) sa_st4((in_), l_r2 + (f_), (size_t)LC, reinterpret_cast<float4 *>(r2) + (f_), (size_t)A.cells, (v_))
#define SA_PATCH_LD(in_, f_, v_) sa_ld4((in_), l_patch + (f_), (size_t)PC, reinterpret_cast<const float4 *>(patch_g) + 4 * (size_t)(f_), (size_t)1, (v_))
#define SA_PATCH_ST(in_, f_, v_) sa_st4((in_), l_patch + (f_), (size_t)PC, reinterpret_cast<float4 *>(patch_g) + 4 * (size_t)(f_), (size_t)1, (v_))
#ifdef YGZ_SA_TIMERS
#define SA_PHASE(k) do { if (tid == 0) { const long long tn_ = clock64(); tph[k] += tn_ - tlast; tlast = tn_; } } while (0)
#define SA_COUNT(k, v) do { if (tid == 0) tph[k] += (v); } while (0)
#else
#define SA_PHASE(k) do { } while (0)
#define SA_COUNT(k, v) do { } while (0)
#endif

// =====================================================================================================================
// The kernel (second form, round 4; the first form -- chain maps by integer arithmetic, patches and Hf in HBM, one workgroup per problem: 1.09 ms
// per 512 VGA pairs against 0.77 -- was removed in round 5, DESIGN.md Appendix B keeps its numbers).  What the second form changed, each measured:
//   * the 16-term parity map of a feature comes from the float adder itself: the chain started at 2^E (even mantissa) and at
//     2^E + ulp (odd mantissa) and pushed through the 16 terms gives the two increments as differences of bit patterns --
//     2 x 16 v_add_f32 instead of 16 x ~20 integer instructions (sa_chain_term) per feature and iteration;
//   * the reference patches of the first `pcap` features live in LDS next to the per-iteration scratch (64 of the 218 bytes an
//     iteration re-read per feature);
//   * the first iteration of a level is fused with precomputeReferencePatches: the patch, its gradients and the feature's
//     contribution to H are still in registers when the first residual pass needs them, so they are not re-read, and Hf -- the
//     21 doubles per feature and level that were written once and read once -- is not stored at all: when a feature enters or
//     leaves the image in a later iteration (rare) its block is recomputed from the stored gradients, bit for bit the same.
struct SaLevel { const uint8_t *ref_img, *cur_img; int cols, rows; float scale; double fl; };

#define SA_BIL(a, b, c, d) __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w_tl, (float)(a)), __fmul_rn(w_tr, (float)(b))), __fmul_rn(w_bl, (float)(c))), __fmul_rn(w_br, (float)(d)))

// cvutils::JacobXYZ2Cam (CVUtils.h:77-99) of the reference feature at xyz_ref = (x, y, dep)
__device__ __forceinline__ void sa_frame_jacobian(double x, double y, double dep, double fj[12])
{
    const double z_inv = 1. / dep, z_inv_2 = z_inv * z_inv;
    fj[0] = -z_inv; fj[1] = 0.0; fj[2] = x * z_inv_2; fj[3] = y * fj[2]; fj[4] = -(1.0 + x * fj[2]); fj[5] = y * z_inv;
    fj[6] = 0.0; fj[7] = -z_inv; fj[8] = y * z_inv_2; fj[9] = 1.0 + y * fj[8]; fj[10] = -fj[3]; fj[11] = -x * z_inv;
}

// the feature's contribution to H = sum over its 16 pixels of J J^T, J = (dx fj_row0 + dy fj_row1) fl (SparseImageAlign.cpp:116-117, :209).
// With f = fj_row0, g = fj_row1:  J_a J_b = fl^2 (dx^2 f_a f_b + dx dy (f_a g_b + g_a f_b) + dy^2 g_a g_b), so the 16 outer products collapse
// into three sums over the patch (Sxx, Sxy, Syy; the products of two floats are exact in double) and 21 combinations: ~360 instead of
// ~1060 FP64 operations per feature and level.  Same quantity, different rounding (1e-16 relative; the pose tolerance of the path is 1e-9).
__device__ __forceinline__ void sa_feature_h(const double fj[12], double fl, const float dxv[16], const float dyv[16], double hf[21])
{
    double sxx = 0.0, sxy = 0.0, syy = 0.0;
#pragma unroll
    for (int pc = 0; pc < 16; ++pc) {
        const double dx = (double)dxv[pc], dy = (double)dyv[pc];
        sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
    }
    const double fl2 = fl * fl;
    sxx *= fl2; sxy *= fl2; syy *= fl2;
    int q = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = a; b < 6; ++b)
            hf[q++] = sxx * (fj[a] * fj[b]) + sxy * (fj[a] * fj[6 + b] + fj[6 + a] * fj[b]) + syy * (fj[6 + a] * fj[6 + b]);
    }
}

// computeResiduals for one feature (SparseImageAlign.cpp:147-207): warp, bounds, bilinear window of the current image, residuals.
// Returns whether the feature is used by this iterate; res stays 0 otherwise.
// (xr, yr, dep) = Pixel2Camera of the reference pixel: the same for every level and iterate, so the fused pass stores it once per level
// and the later passes read it instead of redoing its two FP64 divisions
__device__ __forceinline__ bool sa_feature_residual(const SaArgs &A, const SaLevel &L, const Se3 &T, double xr, double yr, double dep,
                                                    const float refp[16], float res[16])
{
    const int border = 3;
#pragma unroll
    for (int k = 0; k < 16; ++k) res[k] = 0.f;
    const double xyz_ref[3] = { xr, yr, dep };
    double xyz_cur[3];
    se3_act_d(&T, xyz_ref, xyz_cur);
    const double pu = A.fx * xyz_cur[0] / xyz_cur[2] + A.cx, pv = A.fy * xyz_cur[1] / xyz_cur[2] + A.cy;
    const float u_cur = __fmul_rn((float)pu, L.scale), v_cur = __fmul_rn((float)pv, L.scale);
    const int ui = (int)floorf(u_cur), vi = (int)floorf(v_cur);
    if (u_cur != u_cur || v_cur != v_cur || ui < 0 || vi < 0 || ui - border < 0 || vi - border < 0 ||
        ui + border >= L.cols || vi + border >= L.rows) return false;
    const float su = __fsub_rn(u_cur, (float)ui), sv = __fsub_rn(v_cur, (float)vi);
    const float w_tl = (float)((1.0 - (double)su) * (1.0 - (double)sv)), w_tr = (float)((double)su * (1.0 - (double)sv));
    const float w_bl = (float)((1.0 - (double)su) * (double)sv), w_br = __fmul_rn(su, sv);
    uint32_t wl[5], wh[5];            // rows vi-2..vi+2, columns ui-2..ui+2
#pragma unroll
    for (int r = 0; r < 5; ++r) ygz_load5(L.cur_img + (size_t)(vi - 2 + r) * L.cols + (ui - 2), wl[r], wh[r]);
#pragma unroll
    for (int yy = 0; yy < 4; ++yy) {
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
            const float ic = SA_BIL(YGZ_BYTE(wl[yy], wh[yy], xx), YGZ_BYTE(wl[yy], wh[yy], xx + 1),
                                    YGZ_BYTE(wl[yy + 1], wh[yy + 1], xx), YGZ_BYTE(wl[yy + 1], wh[yy + 1], xx + 1));
            res[4 * yy + xx] = __fsub_rn(ic, refp[4 * yy + xx]);
        }
    }
    return true;
}

// Jres_ -= J * res (:210) with J factored as (dx fj_row0 + dy fj_row1) fl
__device__ __forceinline__ void sa_feature_jres(const double fj[12], double fl, const float gxv[16], const float gyv[16], const float res[16], double *accJ)
{
    double gA = 0.0, gB = 0.0;
#pragma unroll
    for (int pc = 0; pc < 16; ++pc) {
        gA += (double)gxv[pc] * (double)res[pc];
        gB += (double)gyv[pc] * (double)res[pc];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) accJ[k] -= (fj[k] * gA + fj[6 + k] * gB) * fl;
}

// H_INLINE: the fused pass adds the block of a used feature to H while its gradients are in registers (96 more live registers: fine for
// 256 lanes with 512 registers each, spills at 512 lanes); otherwise it only marks the feature and the loop that handles entering /
// leaving features recomputes the block from the gradients it just stored (same bits).
template <int SA_THREADS>
__device__ __forceinline__ void sa2_problem(const SaArgs &A, const int pair)
{
    __shared__ double red[SA_THREADS / 64][28];
    __shared__ Se3 sT;
    __shared__ int s_ctl;                 // 0 continue, 1 leave level
    __shared__ int s_nmeas_w[SA_THREADS / 64];
    __shared__ float s_chi2;
    __shared__ double s_ldlt[36 + 6 + 6];       // lane-0 solver workspace (LDS instead of private scratch)
    __shared__ int s_tr[6];
    __shared__ Se3 s_Tn;                        // candidate model of the iteration
    __shared__ double s_nmx;
    __shared__ int s_solve_ok;
    __shared__ double s_H[21];                  // H of the current level (updated by the change per iteration)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    ygz_raise_prio(A.prio);
    const int n = A.trk_n[pair];
    double *out = A.out + 16 * (size_t)pair;
    if (n <= 0) {                                      // run() returns 0 and leaves the pose (:25-29)
        if (tid == 0) { out[7] = 0; for (int l = 0; l < YGZ_MAX_LEVELS; ++l) out[8 + l] = 0; }
        return;
    }
    const int ref_slot = A.pair_t[pair], cur_slot = A.pair_q[pair];
    const double *px = A.trk_px + 2 * (size_t)pair * A.cells, *depth = A.trk_depth + (size_t)pair * A.cells;
    const uint8_t *has_mp = A.trk_has_mp + (size_t)pair * A.cells;
    uint8_t *wk = A.work + (size_t)pair * A.work_stride;
    // global work arrays of the pair (the layout of the first form; its Hf block is unused here)
    double *xy = (double *)wk;                                          // [cells][2] Pixel2Camera x, y of the reference pixel (in the first form's Hf block)
    float *dxy = (float *)((double *)wk + 33 * (size_t)A.cells);       // [cells][32] dx[16], dy[16]
    float *patch_g = (float *)((double *)wk + 96 * (size_t)A.cells);    // [cells][16] reference patches of the features beyond pcap
    float *r2 = patch_g + 16 * (size_t)A.cells;                         // the per-iteration scratch of the features beyond lcap: see the first form
    float *pre = r2 + 16 * (size_t)A.cells + A.cells;                    // (the chunk totals of the first form lay in between)
    int4 *fmap = reinterpret_cast<int4 *>(pre + A.cells);
    int2 *pmap = reinterpret_cast<int2 *>(fmap + A.cells);
    uint8_t *flags = (uint8_t *)(pmap + A.cells);                       // [cells] bit 0: visible (visible_fts_, never reset: :35), bit 1: part of the running H
    extern __shared__ __attribute__((aligned(16))) unsigned char sa_dyn[];
    const int LC = A.lcap, PC = A.pcap;
    float4 *const l_r2 = reinterpret_cast<float4 *>(sa_dyn);                              // [4][LC]
    int4 *const l_fmap = reinterpret_cast<int4 *>(l_r2 + 4 * (size_t)LC);                  // [LC]
    int2 *const l_pmap = reinterpret_cast<int2 *>(l_fmap + LC);                            // [LC]
    float *const l_pre = reinterpret_cast<float *>(l_pmap + LC);                           // [LC]
    float *const l_ctot = l_pre + LC;                                                      // [chunks + 1]: the totals of EVERY chunk of 64 features (lanes of one wavefront read across the tiers)
    float4 *const l_patch = reinterpret_cast<float4 *>(sa_dyn + (((size_t)LC * 92 + (size_t)((A.cells + 63) / 64 + 1) * 4 + 15) & ~(size_t)15));   // [4][PC]
    Se3 T_ref;
    for (int k = 0; k < 4; ++k) T_ref.q[k] = A.pair_T[14 * (size_t)pair + k];
    for (int k = 0; k < 3; ++k) T_ref.t[k] = A.pair_T[14 * (size_t)pair + 4 + k];
    // thread-0 solver state (NLLSSolver::reset, NLSSolver_impl.hpp:283-293)
    double chi2_ = 1e10; bool stop_ = false;
    Se3 old_model;
    int n_meas_last = 0;
#ifdef YGZ_SA_TIMERS
    long long tph[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, tlast = clock64();
#endif
    if (tid == 0) {
        Se3 T_cur, Tri;
        for (int k = 0; k < 4; ++k) T_cur.q[k] = out[k];
        for (int k = 0; k < 3; ++k) T_cur.t[k] = out[4 + k];
        se3_inv_d(&T_ref, &Tri);
        if (A.rel) sT = T_cur;
        else se3_mul_d(&T_cur, &Tri, &sT);              // T_cur_from_ref (SparseImageAlign.cpp:37)
        for (int l = 0; l < YGZ_MAX_LEVELS; ++l) out[8 + l] = 0;
    }
    for (int f = tid; f < n; f += SA_THREADS) {          // a lane only ever touches the flags, patches and gradients of its own features
        flags[f] = 0;
        const float4 z4[4] = { make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f) };
        SA_PATCH_ST((f & ~63) < PC, f, z4);
    }
    __syncthreads();

    for (int level = A.max_level; level >= A.min_level; --level) {
        SaLevel L;
        L.cols = A.w[level]; L.rows = A.h[level];
        L.ref_img = A.lvl[level] + (size_t)ref_slot * L.cols * L.rows;
        L.cur_img = A.lvl[level] + (size_t)cur_slot * L.cols * L.rows;
        L.scale = 1.0f / (float)(1 << level);
        L.fl = (double)((A.fx + A.fy) / 2) / (double)(1 << level);
        const int border = 3;
        if (tid < 21) s_H[tid] = 0.0;
        __syncthreads();
        if (tid == 0) old_model = sT;
        int it = 0;
        for (; it < A.n_iter; ++it) {
            SA_PHASE(0);
            const Se3 T = sT;
            int my_meas = 0;
            double acc[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) acc[k] = 0.0;
            bool any_chg = false;                                   // some feature of this lane entered / left this iteration: bits 2 (changed) and 3 (entered) of its flags byte say which
            if (it == 0) {
                // ---- precomputeReferencePatches (:59-122) fused with the first computeResiduals of the level: jacobian_cache_.setZero()
                // (:42) = zero gradients for the features this level does not refill; H gets the blocks of the used features
#pragma unroll 1
                for (int f0_ = 0; f0_ < n; f0_ += SA_THREADS) {
                    const int f = f0_ + tid;
                    const bool in_l = f0_ + 64 * wv < LC, in_p = f0_ + 64 * wv < PC;      // wave-uniform
                    float sq = 0.f;
                    if (f < n) {
                        const uint8_t fl0 = flags[f], hm = has_mp[f];
                        const double pxx = px[2 * f], pxy = px[2 * f + 1], dep = depth[f];
                        const float u_ref = (float)(pxx * L.scale), v_ref = (float)(pxy * L.scale);
                        const int ui = (int)floorf(u_ref), vi = (int)floorf(v_ref);
                        const bool refill = hm && !(ui - border < 0 || vi - border < 0 || ui + border >= L.cols || vi + border >= L.rows);
                        const bool vis = refill || (fl0 & 1);
                        const double xr = (pxx - A.cx) * dep / A.fx, yr = (pxy - A.cy) * dep / A.fy;      // Pixel2Camera (Camera.h:53-59)
                        if (vis) { xy[2 * f] = xr; xy[2 * f + 1] = yr; }
                        float pv[16], dxv[16], dyv[16];
#pragma unroll
                        for (int k = 0; k < 16; ++k) { pv[k] = 0.f; dxv[k] = 0.f; dyv[k] = 0.f; }
                        if (refill) {
                            const float su = __fsub_rn(u_ref, (float)ui), sv = __fsub_rn(v_ref, (float)vi);
                            const float w_tl = (float)((1.0 - (double)su) * (1.0 - (double)sv)), w_tr = (float)((double)su * (1.0 - (double)sv));
                            const float w_bl = (float)((1.0 - (double)su) * (double)sv), w_br = __fmul_rn(su, sv);
                            // the 7x7 reference window rows vi-3..vi+3, columns ui-3..ui+3: W(r, c) = byte c of row r
                            uint32_t wl[7], wh[7];
#pragma unroll
                            for (int r = 0; r < 7; ++r) ygz_load8(L.ref_img + (size_t)(vi - 3 + r) * L.cols + (ui - 3), wl[r], wh[r]);
#define W(r, c) YGZ_BYTE(wl[r], wh[r], c)
#pragma unroll
                            for (int yy = 0; yy < 4; ++yy) {
#pragma unroll
                                for (int xx = 0; xx < 4; ++xx) {
                                    const int pc = 4 * yy + xx;          // p = &W(yy + 1, xx + 1)
                                    pv[pc] = SA_BIL(W(yy + 1, xx + 1), W(yy + 1, xx + 2), W(yy + 2, xx + 1), W(yy + 2, xx + 2));
                                    dxv[pc] = __fmul_rn(0.5f, __fsub_rn(SA_BIL(W(yy + 1, xx + 2), W(yy + 1, xx + 3), W(yy + 2, xx + 2), W(yy + 2, xx + 3)),
                                                                         SA_BIL(W(yy + 1, xx), W(yy + 1, xx + 1), W(yy + 2, xx), W(yy + 2, xx + 1))));
                                    dyv[pc] = __fmul_rn(0.5f, __fsub_rn(SA_BIL(W(yy + 2, xx + 1), W(yy + 2, xx + 2), W(yy + 3, xx + 1), W(yy + 3, xx + 2)),
                                                                         SA_BIL(W(yy, xx + 1), W(yy, xx + 2), W(yy + 1, xx + 1), W(yy + 1, xx + 2))));
                                }
                            }
#undef W
                            const float4 p4[4] = { make_float4(pv[0], pv[1], pv[2], pv[3]), make_float4(pv[4], pv[5], pv[6], pv[7]),
                                                   make_float4(pv[8], pv[9], pv[10], pv[11]), make_float4(pv[12], pv[13], pv[14], pv[15]) };
                            SA_PATCH_ST(in_p, f, p4);
                        } else if (fl0 & 1) {
                            // visible from a coarser level (visible_fts_ is never reset, :35): the residual pass still visits it with the
                            // patch of that level and the zero columns setZero() left
                            float4 c4[4];
                            SA_PATCH_LD(in_p, f, c4);
#pragma unroll
                            for (int k = 0; k < 4; ++k) { pv[4 * k] = c4[k].x; pv[4 * k + 1] = c4[k].y; pv[4 * k + 2] = c4[k].z; pv[4 * k + 3] = c4[k].w; }
                        }
                        if (vis) {
                            float4 *o4 = reinterpret_cast<float4 *>(dxy + 32 * (size_t)f);
#pragma unroll
                            for (int k = 0; k < 4; ++k) { o4[k] = make_float4(dxv[4 * k], dxv[4 * k + 1], dxv[4 * k + 2], dxv[4 * k + 3]);
                                                          o4[4 + k] = make_float4(dyv[4 * k], dyv[4 * k + 1], dyv[4 * k + 2], dyv[4 * k + 3]); }
                        }
                        float res[16];
#pragma unroll
                        for (int k = 0; k < 16; ++k) res[k] = 0.f;
                        bool use = false;
                        if (vis) use = sa_feature_residual(A, L, T, xr, yr, dep, pv, res);
                        float xs[16];
#pragma unroll
                        for (int k = 0; k < 16; ++k) { xs[k] = __fmul_rn(__fmul_rn(res[k], res[k]), 1.0f); sq += xs[k]; }      // res*res*weight (:213)
                        const float4 x4[4] = { make_float4(xs[0], xs[1], xs[2], xs[3]), make_float4(xs[4], xs[5], xs[6], xs[7]),
                                               make_float4(xs[8], xs[9], xs[10], xs[11]), make_float4(xs[12], xs[13], xs[14], xs[15]) };
                        SA_R2_ST(in_l, f, x4);
                        flags[f] = (uint8_t)((vis ? 1 : 0) | (use ? 2 : 0) | (use && refill ? 12 : 0));
                        if (use) {
                            my_meas += 16;
                            double fj[12];
                            sa_frame_jacobian(xr, yr, dep, fj);
                            sa_feature_jres(fj, L.fl, dxv, dyv, res, acc + 21);
                            if (refill) any_chg = true;        // (its block of H is added by the enter / leave loop below: keeping the blocks in registers here made the kernel faster alone and the step slower)
                        }
                    }
                    const float incl = ygz_wave_scan_f(sq);
                    if (f < n) { if (in_l) *SA_LDS(float, l_pre + f) = incl - sq; else *SA_GLB(float, pre + f) = incl - sq; }
                    if (lane == 63) l_ctot[f >> 6] = incl;
                }
            } else {
                // ---- computeResiduals(model, linearize=true) (:124-223) of a later iteration: everything that does not depend on the
                // iterate is fetched up front, unconditionally and in one batch
#pragma unroll 1
                for (int f0_ = 0; f0_ < n; f0_ += SA_THREADS) {
                    const int f = f0_ + tid;
                    const bool in_l = f0_ + 64 * wv < LC, in_p = f0_ + 64 * wv < PC;      // wave-uniform
                    float sq = 0.f;
                    if (f < n) {
                        const uint8_t fl0 = flags[f];
                        const double xr = xy[2 * f], yr = xy[2 * f + 1], dep = depth[f];
                        float4 c4[4];
                        SA_PATCH_LD(in_p, f, c4);
                        const float4 c0 = c4[0], c1 = c4[1], c2 = c4[2], c3 = c4[3];
                        const float4 *gp = reinterpret_cast<const float4 *>(dxy + 32 * (size_t)f);
                        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0, g3 = g0, g4 = g0, g5 = g0, g6 = g0, g7 = g0;
                        if (fl0 & 1) { g0 = gp[0]; g1 = gp[1]; g2 = gp[2]; g3 = gp[3]; g4 = gp[4]; g5 = gp[5]; g6 = gp[6]; g7 = gp[7]; }
                        const float refp[16] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w };
                        float res[16];
#pragma unroll
                        for (int k = 0; k < 16; ++k) res[k] = 0.f;
                        bool use = false;
                        if (fl0 & 1) use = sa_feature_residual(A, L, T, xr, yr, dep, refp, res);
                        float xs[16];
#pragma unroll
                        for (int k = 0; k < 16; ++k) { xs[k] = __fmul_rn(__fmul_rn(res[k], res[k]), 1.0f); sq += xs[k]; }      // res*res*weight (:213)
                        const float4 x4[4] = { make_float4(xs[0], xs[1], xs[2], xs[3]), make_float4(xs[4], xs[5], xs[6], xs[7]),
                                               make_float4(xs[8], xs[9], xs[10], xs[11]), make_float4(xs[12], xs[13], xs[14], xs[15]) };
                        SA_R2_ST(in_l, f, x4);
                        const bool pu = (fl0 & 2) != 0;
                        if (use != pu) {                                     // rare: H changes by +-(the feature's block), added in the second loop below
                            flags[f] = (uint8_t)((fl0 & 1) | (use ? 2 : 0) | 4 | (use ? 8 : 0));
                            any_chg = true;
                        }
                        if (use) {
                            my_meas += 16;
                            const float gxv[16] = { g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w, g3.x, g3.y, g3.z, g3.w };
                            const float gyv[16] = { g4.x, g4.y, g4.z, g4.w, g5.x, g5.y, g5.z, g5.w, g6.x, g6.y, g6.z, g6.w, g7.x, g7.y, g7.z, g7.w };
                            double fj[12];
                            sa_frame_jacobian(xr, yr, dep, fj);
                            sa_feature_jres(fj, L.fl, gxv, gyv, res, acc + 21);
                        }
                    }
                    const float incl = ygz_wave_scan_f(sq);
                    if (f < n) { if (in_l) *SA_LDS(float, l_pre + f) = incl - sq; else *SA_GLB(float, pre + f) = incl - sq; }
                    if (lane == 63) l_ctot[f >> 6] = incl;
                }
            }
            SA_PHASE(8);      // feature loop
            if (__ballot(any_chg) != 0ull) {                        // wave-uniform skip: no feature of this wavefront changed state
#pragma unroll 1
                for (int c = 0; c * SA_THREADS < n; ++c) {
                    const int f = c * SA_THREADS + tid;
                    if (!any_chg || f >= n) continue;
                    const uint8_t flc = flags[f];
                    if (!(flc & 4)) continue;
                    flags[f] = (uint8_t)(flc & 3);
                    const bool add = (flc & 8) != 0;
                    const float4 *gp = reinterpret_cast<const float4 *>(dxy + 32 * (size_t)f);
                    const float4 g0 = gp[0], g1 = gp[1], g2 = gp[2], g3 = gp[3], g4 = gp[4], g5 = gp[5], g6 = gp[6], g7 = gp[7];
                    const float gxv[16] = { g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w, g3.x, g3.y, g3.z, g3.w };
                    const float gyv[16] = { g4.x, g4.y, g4.z, g4.w, g5.x, g5.y, g5.z, g5.w, g6.x, g6.y, g6.z, g6.w, g7.x, g7.y, g7.z, g7.w };
                    double fj[12], hf[21];
                    sa_frame_jacobian(xy[2 * f], xy[2 * f + 1], depth[f], fj);
                    sa_feature_h(fj, L.fl, gxv, gyv, hf);      // the block the fused pass of this level added (or would have added)
#pragma unroll
                    for (int k = 0; k < 21; ++k) acc[k] += add ? hf[k] : -hf[k];
                }
            }
            SA_PHASE(9);      // enter / leave blocks
            // the 27 wavefront sums, stage by stage over all values (ygz_wave_sums_d: the additions of ygz_wave_sum_d, 27 independent chains per
            // stage; one after the other they were 6.2 of the 85 thousand cycles of an iteration): lanes 48..63 end up with the totals
            ygz_wave_sums_d(acc);
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < 27; ++k) red[wv][k] = acc[k];
            }
            {
                int m = my_meas;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) m += __shfl_xor(m, off);
                if (lane == 0) s_nmeas_w[wv] = m;
            }
            SA_PHASE(10);     // wave sums
            __syncthreads();
            SA_PHASE(1);      // barrier behind the residual pass
            // ---- chain, step 1 (all lanes, lane = feature, wavefront = chunk of 64): the 16-term map of every feature for its
            // predicted binade, then a SEGMENTED scan of the maps along the chunk (see the first form).  The map itself comes from
            // the float adder: starting at 2^E (m even) and at 2^E + ulp (m odd), 16 round-to-nearest-even adds ARE the increments
            // of the integer description, as long as the sum stays in the binade -- and if it does not, the exponent of the result
            // says so and the feature is marked `bad` (taken term by term by the walk), exactly the cases the integer form rejected.
            for (int f0_ = 0; f0_ < n; f0_ += SA_THREADS) {
                const int chunk = (f0_ >> 6) + wv;
                if (64 * chunk >= n) continue;                                // wave-uniform
                const int f = f0_ + tid, cnt = min(64, n - 64 * chunk);
                const bool in_l = 64 * chunk < LC;                             // wave-uniform
                float part = 0.f;
                for (int c2 = lane; c2 < chunk; c2 += 64) part += l_ctot[c2];
                const float base = ygz_wave_sum_f(part);
                int t0 = 0, t1 = 0, bad = 0, E = 0;
                if (f < n) {
                    float4 a4[4];
                    SA_R2_LD(in_l, f, a4);
                    const float4 a0 = a4[0], a1 = a4[1], a2 = a4[2], a3 = a4[3];
                    const float x[16] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w };
                    float pre_f;
                    if (in_l) pre_f = *SA_LDS(float, l_pre + f); else pre_f = *SA_GLB(float, pre + f);
                    E = (int)(__float_as_uint(base + pre_f) >> 23);
                    bad = !(E > 0 && E < 255);
                    const uint32_t e0 = (uint32_t)E << 23;
                    float ce = __uint_as_float(e0), co = __uint_as_float(e0 | 1u);
#pragma unroll
                    for (int k = 0; k < 16; ++k) { ce = __fadd_rn(ce, x[k]); co = __fadd_rn(co, x[k]); }
                    const uint32_t be = __float_as_uint(ce), bo = __float_as_uint(co);
                    bad |= ((be >> 23) != (uint32_t)E) | ((bo >> 23) != (uint32_t)E);
                    t0 = (int)(be - e0); t1 = (int)(bo - (e0 | 1u));
                }
                if (lane >= cnt) E = __builtin_amdgcn_readlane(E, cnt - 1);   // past the end: the empty map in the last feature's binade
                const int prev = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) << 2, E | (bad << 16));
                const int head = (lane == 0) | (prev != (E | (bad << 16))) | bad;      // bad features stand alone
                int p0 = t0, p1 = t1, hd = head;
#define SA_SCAN(ctrl, rmask, valid)                                                                                  \
                { const int l0_ = __builtin_amdgcn_update_dpp(0, p0, (ctrl), (rmask), 0xF, false),                   \
                            l1_ = __builtin_amdgcn_update_dpp(0, p1, (ctrl), (rmask), 0xF, false),                   \
                            lh_ = __builtin_amdgcn_update_dpp(0, hd, (ctrl), (rmask), 0xF, false);                   \
                  if (valid) { if (!hd) { const int n0_ = l0_ + ((l0_ & 1) ? p1 : p0), n1_ = l1_ + (((1 + l1_) & 1) ? p1 : p0); \
                                          p0 = min(n0_, 0x10000000); p1 = min(n1_, 0x10000000); }                    \
                               hd |= lh_; } }
                SA_SCAN(0x111, 0xF, (lane & 15) >= 1) SA_SCAN(0x112, 0xF, (lane & 15) >= 2)
                SA_SCAN(0x114, 0xF, (lane & 15) >= 4) SA_SCAN(0x118, 0xF, (lane & 15) >= 8)
                SA_SCAN(0x142, 0xA, (lane & 16) != 0)                          // row_bcast:15 -> rows 1, 3
                SA_SCAN(0x143, 0xC, lane >= 32)                                // row_bcast:31 -> rows 2, 3
#undef SA_SCAN
                if (f < n) {
                    const sa_v4i fm = { t0, t1, E, bad | (head << 1) }; const sa_v2i pm = { p0, p1 };
                    if (in_l) { *SA_LDS(sa_v4i, l_fmap + f) = fm; *SA_LDS(sa_v2i, l_pmap + f) = pm; }
                    else { *SA_GLB(sa_v4i, fmap + f) = fm; *SA_GLB(sa_v2i, pmap + f) = pm; }
                }
            }
            __syncthreads();
            SA_PHASE(13);     // maps + scan
            if (wv == 0) {
                // ---- wave 0: the walk over the segments (see the first form).  Per chunk of 64 features it needs the maps only (fmap, pmap: 24 bytes
                // per lane, the next chunk's in flight); the 16 terms of a feature are fetched when a feature has to be added term by term (about
                // ten per iteration), as one broadcast read.  A chunk that is ONE segment in the running sum's binade and stays inside it -- most
                // chunks -- is taken by a short straight-line test before the general loop.
                uint32_t cb = 0u;                                            // bits of c, wave-uniform
                const int J = (n + 63) >> 6;
                int4 nm = make_int4(0, 0, 0, 0); int2 np = make_int2(0, 0);
#define SA_LOAD_MAPS(j_)                                                                                                  \
                { const int fn_ = 64 * (j_) + lane;                                                                        \
                  nm = make_int4(0, 0, 0, 0); np = make_int2(0, 0);                                                        \
                  if (fn_ < n) {                                                                                           \
                      sa_v4i fm_; sa_v2i pm_;                                                                              \
                      if (64 * (j_) < LC) { fm_ = *SA_LDS(const sa_v4i, l_fmap + fn_); pm_ = *SA_LDS(const sa_v2i, l_pmap + fn_); } \
                      else { fm_ = *SA_GLB(const sa_v4i, fmap + fn_); pm_ = *SA_GLB(const sa_v2i, pmap + fn_); }           \
                      nm = make_int4(fm_.x, fm_.y, fm_.z, fm_.w); np = make_int2(pm_.x, pm_.y); } }
                SA_LOAD_MAPS(0)
#define SA_TRY(m0_, m1_, mE_, mbad_, taken)                                                                            \
                    { taken = false;                                                                                   \
                      if (!(mbad_) && (int)(cb >> 23) == (mE_)) {                                                      \
                          const uint32_t m_ = (cb & 0x7fffffu) | 0x800000u, mn_ = m_ + (uint32_t)((m_ & 1u) ? (m1_) : (m0_)); \
                          if (mn_ < 0x1000000u) { cb = (cb & 0xff800000u) | (mn_ & 0x7fffffu); taken = true; } } }
                for (int j = 0; j < J; ++j) {
                    const int f0 = nm.x, f1 = nm.y, E = nm.z, bad = nm.w & 1, p0 = np.x, p1 = np.y;
                    const int cnt = min(64, n - 64 * j);
                    const unsigned long long H = __ballot(((nm.w >> 1) & 1) && lane < cnt);
                    SA_LOAD_MAPS(j + 1)
                    int pos = 0;
                    if (H == 1ull) {                                         // one segment (lane 0 is always a head): the first step of the loop below, alone
                        bool taken;
                        SA_TRY(__builtin_amdgcn_readlane(p0, cnt - 1), __builtin_amdgcn_readlane(p1, cnt - 1), __builtin_amdgcn_readlane(E, 0),
                               __builtin_amdgcn_readlane(bad, 0), taken)
                        if (taken) continue;
                    }
                    while (pos < cnt) {
                        bool taken;
                        if ((H >> pos) & 1ull) {
                            const unsigned long long rest = pos < 63 ? (H >> (pos + 1)) : 0ull;
                            const int e = rest ? pos + (int)__builtin_ctzll(rest) : cnt - 1;        // last feature of the segment
                            const int sE = __builtin_amdgcn_readlane(E, pos), sbad = __builtin_amdgcn_readlane(bad, pos);
                            SA_TRY(__builtin_amdgcn_readlane(p0, e), __builtin_amdgcn_readlane(p1, e), sE, sbad, taken)
                            if (taken) { pos = e + 1; continue; }
                            if (!sbad && (int)(cb >> 23) == sE) {
                                // the segment leaves the binade: the first feature whose prefix does (the prefixes are monotone)
                                const uint32_t m_ = (cb & 0x7fffffu) | 0x800000u;
                                const int inc = (m_ & 1u) ? p1 : p0;
                                unsigned long long over = __ballot(m_ + (uint32_t)inc >= 0x1000000u);
                                over &= (~0ull << pos) & (e < 63 ? ((2ull << e) - 1ull) : ~0ull);
                                if (over) {
                                    const int jx = (int)__builtin_ctzll(over);
                                    if (jx > pos) {
                                        SA_TRY(__builtin_amdgcn_readlane(p0, jx - 1), __builtin_amdgcn_readlane(p1, jx - 1), sE, 0, taken)
                                        pos = jx;
                                    }
                                }
                            }
                        }
                        // one feature: its own map if c is in its binade and stays there, else its 16 terms in hardware floats
                        SA_TRY(__builtin_amdgcn_readlane(f0, pos), __builtin_amdgcn_readlane(f1, pos), __builtin_amdgcn_readlane(E, pos),
                               __builtin_amdgcn_readlane(bad, pos), taken)
                        if (!taken) {
                            SA_COUNT(7, 1);
                            float4 v4[4];                                   // every lane reads the feature's 16 terms (one address: a broadcast)
                            SA_R2_LD(64 * j < LC, 64 * j + pos, v4);
                            const float x[16] = { v4[0].x, v4[0].y, v4[0].z, v4[0].w, v4[1].x, v4[1].y, v4[1].z, v4[1].w,
                                                  v4[2].x, v4[2].y, v4[2].z, v4[2].w, v4[3].x, v4[3].y, v4[3].z, v4[3].w };
                            float cc = __uint_as_float(cb);
#pragma unroll
                            for (int k = 0; k < 16; ++k) cc = __fadd_rn(cc, x[k]);
                            cb = __builtin_amdgcn_readfirstlane(__float_as_uint(cc));
                        }
                        ++pos;
                    }
                }
#undef SA_LOAD_MAPS
#undef SA_TRY
                const float c = __uint_as_float(cb);
                if (lane == 0) s_chi2 = c;
                SA_COUNT(2, clock64() - tlast); SA_COUNT(5, 1);
            } else if (tid == 64) {
                // ---- meanwhile, one lane of wave 1: the 6x6 solve and the candidate model (NLSSolver_impl.hpp:40-52, 66-75)
                double Hm[36], Jr[6], x[6];
                int q = 0;
                for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) {
                    double sm = 0; for (int w2 = 0; w2 < SA_THREADS / 64; ++w2) sm += red[w2][q];
                    sm = s_H[q] + sm; s_H[q] = sm;                // running H of the level
                    Hm[6 * a + b] = sm; Hm[6 * b + a] = sm; ++q;
                }
                for (int a = 0; a < 6; ++a) { double sm = 0; for (int w2 = 0; w2 < SA_THREADS / 64; ++w2) sm += red[w2][21 + a]; Jr[a] = sm; }
                if (A.lin) {                                     // H_ and Jres_ as computeResiduals left them (ygz_hip_sparse_align_residuals)
                    double *lo = A.lin + 32 * (size_t)pair;
                    for (int k = 0; k < 21; ++k) lo[2 + k] = s_H[k];
                    for (int k = 0; k < 6; ++k) lo[23 + k] = Jr[k];
                }
                const bool okx = ldlt6_solve_ws(Hm, Jr, x, s_ldlt, s_ldlt + 36, s_ldlt + 42, s_tr);
                double mx[6]; for (int k = 0; k < 6; ++k) mx[k] = -x[k];
                Se3 E, Tn;
                se3_exp_d(mx, &E);
                se3_mul_d(&T, &E, &Tn);
                double nmx = -1; for (int k = 0; k < 6; ++k) if (fabs(x[k]) > nmx) nmx = fabs(x[k]);
                s_Tn = Tn; s_nmx = nmx; s_solve_ok = okx ? 1 : 0;
            }
            __syncthreads();
            SA_PHASE(3);      // chain || accumulation
            // ---- lane 0: accept / stop decision (NLSSolver_impl.hpp:53-63, 85-87)
            if (tid == 0) {
                int nm = 0; for (int w2 = 0; w2 < SA_THREADS / 64; ++w2) nm += s_nmeas_w[w2];
                n_meas_last = nm;
                if (A.lin) { A.lin[32 * (size_t)pair] = (double)s_chi2; A.lin[32 * (size_t)pair + 1] = (double)nm; }
                const double new_chi2 = (double)__fdiv_rn(s_chi2, (float)nm);
                if (!s_solve_ok) stop_ = true;
                int ctl = 0;
                if ((it > 0 && new_chi2 > chi2_) || stop_) { sT = old_model; ctl = 1; }
                else {
                    old_model = T; sT = s_Tn;
                    chi2_ = new_chi2;
                    if (s_nmx <= 0.000001) ctl = 2;        // eps_ (SparseImageAlign.cpp:18)
                }
                s_ctl = ctl;
            }
            __syncthreads();
            const int ctl = s_ctl;
            SA_PHASE(4);      // solve + update
            __syncthreads();
            if (ctl == 2) { ++it; break; }
            if (ctl == 1) break;
        }
        if (tid == 0 && level < YGZ_MAX_LEVELS) out[8 + level] = (double)it;
    }
    if (tid == 0) {
        Se3 o;
        if (A.rel) o = sT;
        else se3_mul_d(&sT, &T_ref, &o);                   // cur->_TCW = T_cur_from_ref * ref->_TCW (:48)
        for (int k = 0; k < 4; ++k) out[k] = o.q[k];
        for (int k = 0; k < 3; ++k) out[4 + k] = o.t[k];
        out[7] = (double)n_meas_last;
        if (A.out_host) for (int k = 0; k < 16; ++k) A.out_host[16 * (size_t)pair + k] = out[k];
#ifdef YGZ_SA_TIMERS
        if (A.dbg) for (int k = 0; k < 16; ++k) A.dbg[16 * (size_t)pair + k] = (double)tph[k];
#endif
    }
}
#undef SA_BIL

// A workgroup takes the problems blockIdx.x, blockIdx.x + gridDim.x, ...  With one problem per workgroup (gridDim.x = pairs) a launch of
// 512 problems needs two rounds on 256 CUs, and the second round only starts where a whole CU's worth of registers and LDS becomes free:
// beside a kernel of small workgroups (LK: four wavefronts of 112 registers, refilled as fast as they retire) it starves -- the device
// timeline of the step showed the alignment stretched from 1.1 to 4.8 ms, ending after everything else.  Resident workgroups that loop
// over their problems (gridDim.x = CUs) keep the slot they got when the GPU was empty.
template <int SA_THREADS>
__global__ __launch_bounds__(SA_THREADS, 1) void k_sparse_align2(SaArgs A)
{
    for (int pair = blockIdx.x; pair < A.n_pairs; pair += gridDim.x) {
        sa2_problem<SA_THREADS>(A, pair);
        __syncthreads();
    }
}

int ygz_launch_sparse_align(ygz_hip_ctx *ctx, int n_pairs, int max_level, int min_level, int n_iter)
{
    SaArgs A;
    for (int L = 0; L < YGZ_MAX_LEVELS; ++L) { A.lvl[L] = ctx->lvl[L]; A.w[L] = ctx->lw[L]; A.h[L] = ctx->lh[L]; }
    A.fx = ctx->prm.fx; A.fy = ctx->prm.fy; A.cx = ctx->prm.cx; A.cy = ctx->prm.cy;
    A.cells = ctx->cells; A.max_level = max_level; A.min_level = min_level; A.n_iter = n_iter;
    A.pair_q = ctx->pair_q; A.pair_t = ctx->pair_t; A.trk_n = ctx->trk_n; A.pair_T = ctx->pair_T;
    A.trk_px = ctx->trk_px; A.trk_depth = ctx->trk_depth; A.trk_has_mp = ctx->trk_has_mp;
    A.work = ctx->sa_work; A.work_stride = ctx->sa_work_stride; A.out = ctx->sa_out;
    A.dbg = nullptr; A.prio = ctx->wave_prio_mask & 1; A.n_pairs = n_pairs;
    A.lin = ctx->sa_lin; A.rel = ctx->sa_rel ? 1 : 0;
    A.out_host = ctx->sa_out_host;
#ifdef YGZ_SA_TIMERS
    { void *dd = nullptr; if (ygz_scratch(ctx, SCR_GEN_0 + 1, (size_t)n_pairs * 16 * 8, &dd) == YGZ_OK) A.dbg = (double *)dd; }
#endif
    // a problem gets 512 lanes when the launch leaves CUs idle anyway (few pairs: the single-frame surface calls) or when the grid is too big
    // for the per-lane feature loop of the 256-lane form; else 256 (one wavefront per SIMD: the VALU-bound stages of other streams fill its
    // idle issue slots)
    // (YGZ_SA_THREADS=256 / 512 pins the shape where the grid allows both: how the tests reach the 256-lane form with a handful of pairs)
    const char *env_t = getenv("YGZ_SA_THREADS");
    int threads = env_t ? (atoi(env_t) == 256 ? 256 : 512) : (2 * n_pairs <= ctx->n_cu ? 512 : 256);
    if (ctx->cells > 32 * 256) threads = 512;
    // per-iteration scratch of the first lcap features in LDS: 4 x 16 (r2) + 16 (fmap) + 8 (pmap) + 4 (pre) bytes each + chunk totals, then
    // 64 bytes (the reference patch) for the first pcap features from what the scratch leaves
    // (a 512-lane problem owns its CU -- nothing else fits beside 512 x 256 registers -- so it may take nearly all of the LDS)
    const int cells64 = (ctx->cells + 63) / 64 * 64;
    // (a single-frame call knows its feature count: scratch for exactly those, the rest of the LDS for their patches)
    const int lcap_want = ctx->sa_n_hint > 0 ? std::min(1600, std::max(64, ctx->sa_n_hint)) : (threads == 512 ? 1600 : 1024);
    A.lcap = ((lcap_want < ctx->cells ? lcap_want : ctx->cells) + 63) / 64 * 64;
    if (ctx->lds_per_block <= 0) {                                                 // dynamic + static LDS must fit the device's per-block limit (static: < 3 KB, see -Rpass-analysis=kernel-resource-usage)
        int v = 64 * 1024;
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device);
        ctx->lds_per_block = v;
    }
    const int lim = ctx->lds_per_block;
    const size_t ctot_b = (size_t)(cells64 / 64 + 1) * 4;
    {
        const int room = (int)(((long)lim - 4096 - 16 - (long)ctot_b) / 92);       // 92 bytes per feature
        if (A.lcap > room) A.lcap = room > 0 ? room / 64 * 64 : 0;
    }
    size_t dyn = (size_t)A.lcap * 92 + ctot_b + 16;
    dyn = (dyn + 15) & ~(size_t)15;
    {
        const long left = (long)lim - 4096 - 16 - (long)dyn;
        int pc = left > 0 ? (int)(left / 64) / 64 * 64 : 0;
        if (pc > cells64) pc = cells64;
        A.pcap = pc;
        dyn += (size_t)pc * 64;
    }
    void (*const k2_512)(SaArgs) = k_sparse_align2<512>;
    void (*const k2_256)(SaArgs) = k_sparse_align2<256>;
    if (!ctx->sa_attr_set) {                                                       // function attributes are per device: once per context
        YGZ_HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k2_512), hipFuncAttributeMaxDynamicSharedMemorySize, lim - 4096));
        YGZ_HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k2_256), hipFuncAttributeMaxDynamicSharedMemorySize, lim - 4096));
        ctx->sa_attr_set = true;
    }
    // resident workgroups: one per CU, each loops over its problems (see k_sparse_align2)
    const int grid = n_pairs > ctx->n_cu ? ctx->n_cu : n_pairs;
    if (threads == 512) YGZ_LAUNCH_DYN(ctx, KID_SPARSE_ALIGN, k2_512, dim3(grid), dim3(512), dyn, A);
    else YGZ_LAUNCH_DYN(ctx, KID_SPARSE_ALIGN, k2_256, dim3(grid), dim3(256), dyn, A);
    YGZ_HIPCHK(ctx, hipGetLastError());
#ifdef YGZ_SA_TIMERS
    if (A.dbg) {                                                                   // cycles of lane 0 per phase, problem 0 (a -DYGZ_SA_TIMERS build only)
        double h[16];
        (void)hipMemcpyAsync(h, A.dbg, sizeof(h), hipMemcpyDeviceToHost, ctx->stream); (void)hipStreamSynchronize(ctx->stream);
        fprintf(stderr, "[sa-timers] %d lanes: loop top %.0f; feature loop %.0f; enter / leave %.0f; wave sums %.0f; barrier %.0f; maps + scan %.0f; walk %.0f (of chain || solve %.0f); decide %.0f; walks %.0f, float-chain features %.0f\n",
                threads, h[0], h[8], h[9], h[10], h[1], h[13], h[2], h[3], h[4], h[5], h[7]);
    }
#endif
    return YGZ_OK;
}

// single pair, host arrays: fills track set 0 and runs the batched kernel with one pair
extern "C" int ygz_hip_sparse_align(ygz_hip_ctx *ctx, int ref_slot, const double T_ref[7], int cur_slot, double T_cur[7],
                                    const double *px, const double *depth, const uint8_t *has_mappoint, int n,
                                    int max_level, int min_level, int n_iter, int *n_meas_out, int *iters_out)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !T_ref || !T_cur || n < 0 || ref_slot < 0 || ref_slot >= ctx->prm.max_frames || cur_slot < 0 ||
        cur_slot >= ctx->prm.max_frames || min_level < 0 || max_level < min_level || max_level >= ctx->prm.pyramid_levels || n_iter < 0)
        return YGZ_E_INVALID;
    if (n_meas_out) *n_meas_out = 0;
    if (n == 0) return YGZ_OK;                                // run() returns 0 without touching the pose (:25-29)
    if (!px || !depth || !has_mappoint) return YGZ_E_INVALID;
    if (n > ctx->cells) return YGZ_E_CAPACITY;
    if (!ctx->pyr_valid[ref_slot] || !ctx->pyr_valid[cur_slot]) return YGZ_E_STATE;
    // everything the call uploads -- the pair tables and the reference features -- as ONE packed transfer + one scatter kernel (ygz_pack_*: nine
    // copies on the stream before), the 16 result doubles back into the page-locked arena: one wait per call
    const size_t N = (size_t)n;
    YgzPack pk;
    int rc = ygz_pack_begin(ctx, &pk, N * 25 + 64 + 14 * 8 + 16 + (size_t)ctx->prm.max_frames * 8 + 64, SCR_GEN_0 + 6);
    if (rc != YGZ_OK) return rc;
    if ((rc = ygz_track_set_pairs(ctx, &cur_slot, &ref_slot, T_cur, T_ref, 1, &pk)) != YGZ_OK) return rc;
    double *h_px = (double *)ygz_pack_add(&pk, ctx->trk_px, N * 16), *h_dep = (double *)ygz_pack_add(&pk, ctx->trk_depth, N * 8);
    uint8_t *h_mp = (uint8_t *)ygz_pack_add(&pk, ctx->trk_has_mp, N);
    int32_t *h_n = (int32_t *)ygz_pack_add(&pk, ctx->trk_n, 4);
    double *h_T = (double *)ygz_pack_add(&pk, ctx->sa_out, 7 * 8);
    if (!h_px || !h_dep || !h_mp || !h_n || !h_T) return YGZ_E_CAPACITY;
    memcpy(h_px, px, N * 16); memcpy(h_dep, depth, N * 8); memcpy(h_mp, has_mappoint, N); memcpy(h_T, T_cur, 56); *h_n = n;
    if ((rc = ygz_pack_upload(ctx, &pk)) != YGZ_OK) return rc;
    double *h_out = (double *)ygz_stage(ctx, 16 * 8);
    if (!h_out) return YGZ_E_HIP;
    // the kernel stores its 16 result doubles into the page-locked block itself (mapped into the device's address space): no copy back on the stream
    const bool direct = ygz_zero_copy();
    ctx->sa_n_hint = n; ctx->sa_out_host = direct ? h_out : nullptr;
    rc = ygz_launch_sparse_align(ctx, 1, max_level, min_level, n_iter);
    ctx->sa_n_hint = 0; ctx->sa_out_host = nullptr;
    if (rc != YGZ_OK) return rc;
    if (!direct) YGZ_HIPCHK(ctx, hipMemcpyAsync(h_out, ctx->sa_out, 16 * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->wait_hook) {                                     // the caller's own host work while the kernel runs (ygz_hip_set_wait_hook), once
        void (*const fn)(void *) = ctx->wait_hook; void *const user = ctx->wait_hook_user;
        ctx->wait_hook = nullptr; ctx->wait_hook_user = nullptr;
        fn(user);
    }
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 7; ++k) T_cur[k] = h_out[k];
    if (n_meas_out) *n_meas_out = (int)(h_out[7] / 16);        // run() returns n_meas_/patch_area_ (:49)
    if (iters_out) for (int l = 0; l < ctx->prm.pyramid_levels; ++l) iters_out[l] = (int)h_out[8 + l];
    return YGZ_OK;
}

// One NLLSSolver::computeResiduals(model, linearize_system = true, false) of SparseImgAlign (src/Algorithm/SparseImageAlign.cpp:124-223, with
// precomputeReferencePatches :59-122 for the level) at a model the CALLER holds: what a solver other than the resident Gauss-Newton loop needs --
// the class surface's Levenberg-Marquardt (NLSSolver_impl.hpp:91-212) drives its trials with it.  One launch of the same kernel with one level and
// one iteration; the model goes in as T_cur_from_ref itself.  chi2_sum = the reference's float running sum (exact), n_meas = measurements
// (16 per feature used), H (6 x 6, symmetric) and Jres as H_ / Jres_ after the call.
extern "C" int ygz_hip_sparse_align_residuals(ygz_hip_ctx *ctx, int ref_slot, int cur_slot, const double T_cur_from_ref[7], const double *px,
                                              const double *depth, const uint8_t *has_mappoint, int n, int level, double *chi2_sum, int *n_meas,
                                              double *H, double *Jres)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !T_cur_from_ref || n < 0 || ref_slot < 0 || ref_slot >= ctx->prm.max_frames || cur_slot < 0 || cur_slot >= ctx->prm.max_frames ||
        level < 0 || level >= ctx->prm.pyramid_levels || !chi2_sum || !n_meas) return YGZ_E_INVALID;
    *chi2_sum = 0.0; *n_meas = 0;
    if (H) for (int k = 0; k < 36; ++k) H[k] = 0.0;
    if (Jres) for (int k = 0; k < 6; ++k) Jres[k] = 0.0;
    if (n == 0) return YGZ_OK;
    if (!px || !depth || !has_mappoint) return YGZ_E_INVALID;
    if (n > ctx->cells) return YGZ_E_CAPACITY;
    if (!ctx->pyr_valid[ref_slot] || !ctx->pyr_valid[cur_slot]) return YGZ_E_STATE;
    const size_t N = (size_t)n;
    YgzPack pk;
    int rc = ygz_pack_begin(ctx, &pk, N * 25 + 64 + 14 * 8 + 16 + (size_t)ctx->prm.max_frames * 8 + 64, SCR_GEN_0 + 6);
    if (rc != YGZ_OK) return rc;
    const double I7[7] = { 0, 0, 0, 1, 0, 0, 0 };
    if ((rc = ygz_track_set_pairs(ctx, &cur_slot, &ref_slot, T_cur_from_ref, I7, 1, &pk)) != YGZ_OK) return rc;
    double *h_px = (double *)ygz_pack_add(&pk, ctx->trk_px, N * 16), *h_dep = (double *)ygz_pack_add(&pk, ctx->trk_depth, N * 8);
    uint8_t *h_mp = (uint8_t *)ygz_pack_add(&pk, ctx->trk_has_mp, N);
    int32_t *h_n = (int32_t *)ygz_pack_add(&pk, ctx->trk_n, 4);
    double *h_T = (double *)ygz_pack_add(&pk, ctx->sa_out, 7 * 8);
    if (!h_px || !h_dep || !h_mp || !h_n || !h_T) return YGZ_E_CAPACITY;
    memcpy(h_px, px, N * 16); memcpy(h_dep, depth, N * 8); memcpy(h_mp, has_mappoint, N); memcpy(h_T, T_cur_from_ref, 56); *h_n = n;
    if ((rc = ygz_pack_upload(ctx, &pk)) != YGZ_OK) return rc;
    void *d_lin = nullptr;
    if ((rc = ygz_scratch(ctx, SCR_GEN_0 + 3, 32 * 8, &d_lin)) != YGZ_OK) return rc;
    YGZ_HIPCHK(ctx, hipMemsetAsync(d_lin, 0, 32 * 8, ctx->stream));
    double *h_lin = (double *)ygz_stage(ctx, 32 * 8);
    if (!h_lin) return YGZ_E_HIP;
    ctx->sa_n_hint = n; ctx->sa_lin = (double *)d_lin; ctx->sa_rel = true;
    rc = ygz_launch_sparse_align(ctx, 1, level, level, 1);
    ctx->sa_n_hint = 0; ctx->sa_lin = nullptr; ctx->sa_rel = false;
    if (rc != YGZ_OK) return rc;
    if ((rc = ygz_kcopy(ctx, h_lin, d_lin, 32 * 8, hipMemcpyDeviceToHost)) != YGZ_OK) return rc;
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *chi2_sum = h_lin[0]; *n_meas = (int)h_lin[1];
    if (H) { int q = 0; for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) { H[6 * a + b] = h_lin[2 + q]; H[6 * b + a] = h_lin[2 + q]; ++q; } }
    if (Jres) for (int k = 0; k < 6; ++k) Jres[k] = h_lin[23 + k];
    return YGZ_OK;
}
