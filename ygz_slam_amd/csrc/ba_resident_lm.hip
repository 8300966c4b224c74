// SURVEY 8f-1 -- the whole Levenberg-Marquardt loop of ba::LocalBAG2O (src/Algorithm/BA.cpp:390-395,501-502: g2o
// OptimizationAlgorithmLevenberg + BlockSolver_6_3 with marginalised points + a Cholesky solve of the reduced pose system)
// resident on the GPU: a team of 1-32 workgroups per BA window (k_ba_lm_team, see there) runs linearisation, Schur complement,
// L D L^T, back-substitution, state update, trial evaluation and the lambda policy without a host round trip; hundreds of
// windows optimise concurrently (one workgroup each), a handful use 32 CUs each.  The host-loop form (ba_lm.hip::ygz_hip_ba_optimize) keeps
// the same arithmetic with the reduced system on the CPU; oracle/ceres_ba.c::yo_g2o_lm restates it for the tests [frozen spec of g2o, see there].
//
// Per LM trial and window (K <= 16 poses, 14 of them free; P points; E edges):
//   1. S = blockdiag(Hpp + lambda I) - sum_l Y_a(l) Hpl_b(l)^T with Y_a(l) = Hpl_a(l) (Hll_l + lambda I)^-1: one wavefront per (pose pair
//      a <= b, part of the points) sweeps its points 128 at a time through the (point, pose) -> edge table, inverts the point blocks it
//      meets, accumulates the 6x6 block in registers and reduces it in a fixed order (no floating-point atomics).
//   2. S = L D L^T by 6x6 block columns in LDS (<= 84 x 84 doubles), the two triangular solves inside one wavefront.
//   3. x_l = Dinv_l (b_l - sum_e Hpl_e^T x_p) (lane = point), T <- exp(x_p) T on the poses (lane = pose), trial chi2, rho, lambda.
// The per-edge blocks (Hpl, residuals) are re-read twice per trial by the same few CUs: plain stores keep them in L2 (the
// streaming stores of the one-shot linearisation, ba.hip, would send every re-read to HBM).
#define YGZ_BA_PLAIN_STORES
#include "ba_dev.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>
#include <float.h>

#define LM_THREADS 256
#define LM_WAVES   (LM_THREADS / 64)
#define LM_MAXKF   20                                   // free poses of a window of k_ba_lm_team: its reduced system ((6 Kf)^2 doubles) is DYNAMIC LDS, sized by the launch's largest window
#define LM_MAXN    (6 * LM_MAXKF)
#define LM_CERES_MAXKF 14                               // k_ba_ceres keeps its reduced system in static LDS
#define LM_CERES_MAXN  (6 * LM_CERES_MAXKF)
#define LM_RB      4                                    // rows (edges of a point) whose 6 x 3 blocks a lane requests together (8: one round trip for a window of 8 keyframes, but 512 registers and spills)
#define LM_RC      8                                    // rows whose observation records (5 values) a lane requests together: one round trip for a window of 8 keyframes
#define LM_LDSK    16                                   // windows of up to 16 poses keep their per-pose tables in LDS

#define lm_wave_sum ygz_wave_sum_d
// block-wide sum in a fixed order (xor tree inside the wavefronts, then the wavefronts left to right); result on every lane
__device__ __forceinline__ double lm_block_sum(double v, double *red /*[LM_WAVES]*/)
{
    v = lm_wave_sum(v);
    __syncthreads();                                   // red is free again
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < LM_WAVES; ++w) s += red[w];
    return s;
}
__device__ __forceinline__ double lm_block_max(double v, double *red)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < LM_WAVES; ++w) s = fmax(s, red[w]);
    return s;
}

__device__ __forceinline__ bool lm_inv3(const double *m, double *r)
{
    const double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
    if (!(fabs(det) > 0) || !isfinite(det)) return false;
    const double id = 1.0 / det;
    r[0] = c0 * id; r[1] = (m[2] * m[7] - m[1] * m[8]) * id; r[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    r[3] = c1 * id; r[4] = (m[0] * m[8] - m[2] * m[6]) * id; r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    r[6] = c2 * id; r[7] = (m[1] * m[6] - m[0] * m[7]) * id; r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

// ba_point_chi2 with the inputs of LM_RC rows requested together: in the resident loop a thread owns one point and its
// edge loop is a chain of dependent round trips to L2 (measured: 53 us per trial for 16 such iterations), not arithmetic
__device__ __forceinline__ double lm_point_chi2_pf(const BaDev &B, int il)
{
    const int lane = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
    const double pt[3] = { B.points[3 * (size_t)il], B.points[3 * (size_t)il + 1], B.points[3 * (size_t)il + 2] };
    double sum = 0.0;
    for (int c0 = 0; c0 < rows; c0 += LM_RC) {
        int ipg[LM_RC], eng[LM_RC]; double oxg[LM_RC], oyg[LM_RC], hubg[LM_RC];
#pragma unroll
        for (int u = 0; u < LM_RC; ++u) {
            const bool in_ = c0 + u < rows;
            const size_t row = (size_t)(row0 + c0 + u);
            ipg[u] = in_ ? B.pose_c[row * 64 + lane] : -1; eng[u] = in_ ? B.enable_c[row * 64 + lane] : 0;
            oxg[u] = in_ ? BA_EC(B.obs_c, row, 2, 0, lane) : 0.0; oyg[u] = in_ ? BA_EC(B.obs_c, row, 2, 1, lane) : 0.0;
            hubg[u] = in_ ? B.huber_c[row * 64 + lane] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < LM_RC; ++u) {
            if (ipg[u] < 0 || !eng[u]) continue;
            double p[3], r[2], rho0, rho1;
            ba_project(B, B.posed + BA_POSED * (size_t)ipg[u], pt, oxg[u], oyg[u], p, r);
            ba_robust(r[0] * r[0] + r[1] * r[1], hubg[u], &rho0, &rho1);
            sum += rho0;
        }
    }
    return sum;
}

// computeActiveErrors + buildSystem at the current state; returns the robustified chi2 (block-uniform)
__device__ double lm_linearize(const BaDev &B, double (*red27)[28], double *red)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) *B.n_behind = 0;
    if (tid < B.K) ba_pose_prep_one(B, tid);
    __syncthreads();
    double chi = 0.0;
    for (int il = tid; il < B.P; il += LM_THREADS) chi += ba_point_edges(B, il);
    chi = lm_block_sum(chi, red);                        // (barriers inside: the per-edge records are visible below)
    for (int a = 0; a < B.Kf; ++a) {                     // pose blocks: every point that sees free pose a contributes
        const int k = B.free_pose[a];
        double acc[27];
#pragma unroll
        for (int i = 0; i < 27; ++i) acc[i] = 0.0;
        for (int il = tid; il < B.P; il += LM_THREADS) ba_pose_contrib(B, il, a, acc);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 27; ++i) { const double v = lm_wave_sum(acc[i]); if (lane == 0) red27[wv][i] = v; }
        __syncthreads();
        if (tid < 27) {
            double s = 0.0;
            for (int w = 0; w < LM_WAVES; ++w) s += red27[w][tid];
            if (tid < 21) {
                int u = 0, rem = tid;
                while (rem >= 6 - u) { rem -= 6 - u; ++u; }
                const int v = u + rem;
                B.Hpp[36 * (size_t)k + 6 * u + v] = s; B.Hpp[36 * (size_t)k + 6 * v + u] = s;
            } else B.bp[6 * (size_t)k + (tid - 21)] = s;
        }
    }
    __syncthreads();
    return chi;
}

// computeActiveErrors at the (trial) state
__device__ double lm_errors(const BaDev &B, double *red)
{
    const int tid = threadIdx.x;
    if (tid < B.K) ba_pose_prep_one(B, tid);
    __syncthreads();
    double chi = 0.0;
    for (int il = tid; il < B.P; il += LM_THREADS) chi += ba_point_chi2(B, il);
    return lm_block_sum(chi, red);
}

// Schur complement of the point blocks: S(a,b) -= sum_l Y_a(l) W_b(l)^T and bs_a -= sum_l Y_a(l) b_l over the points l seen by both
// free poses a <= b.  One wavefront per pose pair, lanes over the points, a fixed-order wavefront sum at the end.  The sweep is
// bound by the latency of its two dependent loads (edge slot of the point in pose a / b, then the 36 block entries), so every lane
// keeps TWO points in flight and the slots of the next two are requested before the blocks of the current ones are used.
// sc_p / sc_l (SC: ceres' Jacobi scaling): W and b_l are taken column-scaled, (Hpl * s_pose) * s_point as the host-loop form does.
// YM: where Y_a = Hpl_a Dinv comes from.  0: read (Y_c, written by the caller's Dinv phase: k_ba_ceres).  1: formed here from the point's stored
// inverse.  2: formed here from Hll + lambda I itself (lm_inv3, the same expressions and bits as a separate phase would store): the team
// kernel then has NO Dinv phase -- 2 MB of Y per trial and window neither written nor read back, one team barrier less per trial.
template <bool SC, int YM = 0>
__device__ __forceinline__ void lm_sweep_point(const BaDev &B, int l, int ca, int cb, bool diag, const double *spb, const double *sc_l,
                                               double *acc /*[36]*/, double *accb /*[6]*/, double lambda = 0.0, int *bad = nullptr)
{
    const bool v = ca >= 0 && cb >= 0;
    const int row0 = v ? B.slot_off[l >> 6] : 0, ln = l & 63;
    double Ya[18], Wb[18];
    if (YM != 0) {
        double Wa[18], Dv[9];
#pragma unroll
        for (int i = 0; i < 18; ++i) { Wa[i] = v ? BA_EC(B.Hpl_c, row0 + ca, 18, i, ln) : 0.0; Wb[i] = v ? BA_EC(B.Hpl_c, row0 + cb, 18, i, ln) : 0.0; }
        if (YM == 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) Dv[i] = v ? B.Dinv[9 * (size_t)l + i] : 0.0;
        } else {
            double D[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) D[i] = v ? BA_PC(B.Hll_c, l, 9, i) : ((i % 4 == 0) ? 1.0 : 0.0);
            D[0] += lambda; D[4] += lambda; D[8] += lambda;
            if (!lm_inv3(D, Dv)) { if (v && !B.point_fixed[l]) *bad = 1; for (int i = 0; i < 9; ++i) Dv[i] = 0.0; }
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) Ya[3 * r + cc] = Wa[3 * r] * Dv[cc] + Wa[3 * r + 1] * Dv[3 + cc] + Wa[3 * r + 2] * Dv[6 + cc];
    } else {
#pragma unroll
        for (int i = 0; i < 18; ++i) { Ya[i] = v ? BA_EC(B.Y_c, row0 + ca, 18, i, ln) : 0.0; Wb[i] = v ? BA_EC(B.Hpl_c, row0 + cb, 18, i, ln) : 0.0; }
    }
    double s0 = 1.0, s1 = 1.0, s2 = 1.0;
    if (SC && v) { s0 = sc_l[3 * (size_t)l]; s1 = sc_l[3 * (size_t)l + 1]; s2 = sc_l[3 * (size_t)l + 2]; }
    if (SC) {
#pragma unroll
        for (int c = 0; c < 6; ++c) { Wb[3 * c] = Wb[3 * c] * spb[c] * s0; Wb[3 * c + 1] = Wb[3 * c + 1] * spb[c] * s1; Wb[3 * c + 2] = Wb[3 * c + 2] * spb[c] * s2; }
    }
    if (v) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[6 * r + c] = fma(Ya[3 * r + 2], Wb[3 * c + 2], fma(Ya[3 * r + 1], Wb[3 * c + 1], fma(Ya[3 * r], Wb[3 * c], acc[6 * r + c])));
        }
        if (diag) {
            double g0 = BA_PC(B.bl_c, l, 3, 0), g1 = BA_PC(B.bl_c, l, 3, 1), g2 = BA_PC(B.bl_c, l, 3, 2);
            if (SC) { g0 *= s0; g1 *= s1; g2 *= s2; }
#pragma unroll
            for (int r = 0; r < 6; ++r) accb[r] += Ya[3 * r] * g0 + Ya[3 * r + 1] * g1 + Ya[3 * r + 2] * g2;
        }
    }
}

template <bool SC>
__device__ __forceinline__ void lm_pair_sweep(const BaDev &B, double *S, double *bs, int n, const double *sc_p, const double *sc_l)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int Kf = B.Kf, P = B.P, npairs = Kf * (Kf + 1) / 2;
    for (int pr = wv; pr < npairs; pr += LM_WAVES) {
        int a = 0, rem = pr;
        while (rem >= Kf - a) { rem -= Kf - a; ++a; }
        const int b = a + rem;
        double acc[36], accb[6], spb[6];
#pragma unroll
        for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) accb[i] = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) spb[c] = SC ? sc_p[6 * (size_t)B.free_pose[b] + c] : 1.0;
        int l = lane;
        int ca0 = l < P ? B.ppc[(size_t)l * Kf + a] : -1, cb0 = l < P ? B.ppc[(size_t)l * Kf + b] : -1;
        int ca1 = l + 64 < P ? B.ppc[(size_t)(l + 64) * Kf + a] : -1, cb1 = l + 64 < P ? B.ppc[(size_t)(l + 64) * Kf + b] : -1;
        for (int base = 0; base < P; base += 128, l += 128) {           // wave-uniform trip count
            const int na0 = l + 128 < P ? B.ppc[(size_t)(l + 128) * Kf + a] : -1, nb0 = l + 128 < P ? B.ppc[(size_t)(l + 128) * Kf + b] : -1;
            const int na1 = l + 192 < P ? B.ppc[(size_t)(l + 192) * Kf + a] : -1, nb1 = l + 192 < P ? B.ppc[(size_t)(l + 192) * Kf + b] : -1;
            lm_sweep_point<SC>(B, l, ca0, cb0, a == b, spb, sc_l, acc, accb);
            lm_sweep_point<SC>(B, l + 64, ca1, cb1, a == b, spb, sc_l, acc, accb);
            ca0 = na0; cb0 = nb0; ca1 = na1; cb1 = nb1;
        }
#pragma unroll
        for (int i = 0; i < 36; ++i) acc[i] = lm_wave_sum(acc[i]);
        if (a == b) {
#pragma unroll
            for (int i = 0; i < 6; ++i) accb[i] = lm_wave_sum(accb[i]);
        }
        if (lane == 0) {
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
                S[(6 * a + r) * n + 6 * b + c] -= acc[6 * r + c];
                if (a != b) S[(6 * b + c) * n + 6 * a + r] -= acc[6 * r + c];
            }
            if (a == b) for (int r = 0; r < 6; ++r) bs[6 * a + r] -= accb[r];
        }
    }
}

// =====================================================================================================================================
// The Levenberg-Marquardt loop by a TEAM of G workgroups per window.  One workgroup per window leaves 255 of the 256 CUs idle
// when a launch holds a handful of windows (the BA round of the offline run) and is bound by what ONE CU can read: the Schur sweep
// alone moves 16 MB per trial through one vector memory pipeline.  Work is dealt out in units whose arithmetic does not depend on who
// runs them, so the result is the same for every G (1 ... 32: the launch picks G from the number of windows; a sharded and an unsharded
// offline run stay bit-identical, and a window whose team timed out is solved again by one workgroup to the same bits):
//   * per-point work (linearisation, point update, trial chi2) = one WAVEFRONT per chunk of 64 points; a chunk leaves a record (chi2,
//     scale, the 27 sums of every free pose: (chunk, pose) tasks of their own) and everybody adds the records in chunk order;
//   * the Schur sweep = one wavefront per (pose pair, PART of the points), LM_V = 8 fixed parts of whole chunks; the wavefront that
//     delivers the last part of a pair adds the parts in part order (an arrival counter per pair).
// Tasks go to wavefront (task mod 4 G), member-minor: consecutive tasks run on different CUs.  Per iteration: linearise | barrier |
// combine.  Per trial: sweep | barrier | EVERY member loads the finished blocks, factors S and substitutes (identical arithmetic: x_p is not
// published) | pose update, point update, trial errors | barrier (chunk records: scale, chi2) | every member takes the same accept /
// reject decision.  The pose state -- the prepared poses T = (q, t, R) -- is private to a workgroup (identical copies): a shared copy
// updated by everybody would be a read-modify-write race.
// Barriers: one monotonic counter per window, lane 0 releases at agent scope before it arrives and acquires after the wait (MI355X: per-CU
// L1 and per-XCD L2 are not coherent); what members exchange is written with agent-scope stores and read with plain loads after the
// barrier's acquire (the parts of a pose pair, read by whoever delivers the last one without a barrier in between: agent-scope loads).
// All blocks of a team must be resident: the launch keeps windows x G <= half of the CUs (a CU holds one of these;
// ygz_hip_ba_set_team_budget), every wait is bounded and a timeout aborts the whole team with YGZ_E_HIP.  blockIdx -> (XCD slot, team
// member): the members of a team share an XCD / L2 when the dispatcher places block b on XCD b % 8 (speed only).
#ifndef LM_V
#define LM_V      8
#endif
#define LM_MAXG   32                           // members of a team (the launch never picks more)
#define LM_NPAIR  (LM_MAXKF * (LM_MAXKF + 1) / 2)
#define LM_HDR    2048                         // per-window header of the scratch (zeroed by the launch): barrier counter + abort flag (bar[0..3]), one
                                               // behind-camera count per member (bar[4..4+LM_MAXG)), one arrival counter per pose pair (bar[64..64+LM_NPAIR))
#define LM_PARTW  (8 + 27 * LM_MAXKF)          // per chunk of 64 points: chi2, max |diag|, singular flag, scale, trial chi2, -, -, -, pose sums [Kf][27]
#define LM_SPW    42                           // per (pair, part): 36 block entries + 6 of the right-hand side
static_assert(16 + 4 * LM_MAXG <= 256 && 256 + 4 * LM_NPAIR <= LM_HDR, "scratch header");

// The per-pair hand-off of the Schur sweep (parts as relaxed agent-scope = write-through stores, s_waitcnt vmcnt(0), a workgroup-scope fence for
// the compiler, a relaxed agent-scope arrival counter; the last arriver re-reads the parts with agent-scope loads) is NOT a release / acquire
// pair in the HSA memory model: it relies on two properties of gfx942 / gfx950 -- stores are counted by vmcnt (there is no separate vscnt)
// and sc1 stores / loads go through to the device-coherent level.  On any other target this must be rewritten, not recompiled.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "ba_resident_lm.hip: the hand-off protocol of the Schur sweep is written for gfx942 / gfx950 (vmcnt counts stores, sc1 write-through)"
#endif
struct LmTeamArgs {
    const BaDev *wins; int n_windows, G, max_iterations; ygz_ba_stats *stats;
    unsigned char *scratch; size_t stride;     // per window: header (LM_HDR) | xpub | part | Sp | Sfin | private pose state of the G members
    int Kmax, prio, Qcap;
    int spread;                                // team placement, see k_ba_lm_team
    int xcd_barrier;                           // 1: teams whose members share an XCD take the barrier without the L2 write-back (YGZ_LM_XCD_BARRIER=0: never)
    long long *dbg;                            // YGZ_LM_DEBUG: [16] wall-clock ticks (10 ns) per phase of member 0 of the first window
};
#define LM_TICK(k) do { if (A.dbg && blockIdx.x == 0 && tid == 0) { const long long tn_ = wall_clock64(); s_t[k] += tn_ - t_prev; t_prev = tn_; } } while (0)

__device__ __forceinline__ double tl_ld(const double *p)
{ return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void tl_st(double *p, double v)
{ __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#define LM_NB 32
// sum (or maximum) over the n chunk records of a window, in chunk order, LM_NB loads in flight (one round trip for a window of up to 2048 points) (a plain loop was a chain of n dependent L2
// round trips: 12 us per iteration for 31 chunks).  The padding adds + 0.0, which leaves the sum as it is.
__device__ __forceinline__ double lm_sum_records(const double *p, int n)
{
    double t = 0.0;
    for (int c0 = 0; c0 < n; c0 += LM_NB) {
        double v[LM_NB];
#pragma unroll
        for (int u = 0; u < LM_NB; ++u) v[u] = c0 + u < n ? p[(size_t)(c0 + u) * LM_PARTW] : 0.0;
#pragma unroll
        for (int u = 0; u < LM_NB; ++u) t += v[u];
    }
    return t;
}
// two sums at once (both sets of loads in flight together)
__device__ __forceinline__ void lm_sum_records2(const double *p, const double *q, int n, double *sp, double *sq)
{
    double t = 0.0, w = 0.0;
    for (int c0 = 0; c0 < n; c0 += LM_NB) {
        double v[LM_NB], x[LM_NB];
#pragma unroll
        for (int u = 0; u < LM_NB; ++u) { v[u] = c0 + u < n ? p[(size_t)(c0 + u) * LM_PARTW] : 0.0; x[u] = c0 + u < n ? q[(size_t)(c0 + u) * LM_PARTW] : 0.0; }
#pragma unroll
        for (int u = 0; u < LM_NB; ++u) { t += v[u]; w += x[u]; }
    }
    *sp = t; *sq = w;
}
__device__ __forceinline__ double lm_max_records(const double *p, int n)
{
    double t = 0.0;
    for (int c0 = 0; c0 < n; c0 += LM_NB) {
        double v[LM_NB];
#pragma unroll
        for (int u = 0; u < LM_NB; ++u) v[u] = c0 + u < n ? p[(size_t)(c0 + u) * LM_PARTW] : 0.0;
#pragma unroll
        for (int u = 0; u < LM_NB; ++u) t = fmax(t, v[u]);
    }
    return t;
}

// false: a member did not arrive in time (or another member gave up): the caller returns, the host reports YGZ_E_HIP
// same_xcd: every member of the team runs on ONE XCD (checked at run time, see k_ba_lm_team).  Then they share that XCD's L2: a member's
// stores are visible to the others once they are acknowledged (vmcnt) and the reader has dropped its CU's L1 (the acquire below), so the
// RELEASE -- buffer_wbl2 sc1, which writes every dirty line of the XCD's L2 back to memory, those of the kernels that share the XCD
// included: 96 MB written per launch for an 11 MB working set (profiles/r04_lm_pmc.md), and the tracking kernels of an offline run slowed
// down beside it -- is not needed.  Members on different XCDs (the L2s are not coherent with each other) keep the full form.
__device__ __forceinline__ bool lm_team_barrier(unsigned *bar, unsigned &epoch, int G, int *s_ok, bool same_xcd)
{
    if (G == 1) { __syncthreads(); return true; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // every wavefront drains its own stores
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
        if (!same_xcd) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * (unsigned)G;
        int ok = 1;
        for (unsigned spins = 0; __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins) {
            if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
            if (spins > (1u << 22)) { __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
}


// Sums of N <= 64 per-lane FP64 values over the 64 lanes of a wavefront, all at once: ba_reduce32's butterfly with one more level
// (32 + 16 + 8 + 4 + 2 + 1 exchanges instead of N separate 6-step reductions).  On return every lane holds the total of value
// *idx = its lane number bit-reversed (6 bits); the order of the additions is fixed.
template <int N>
__device__ __forceinline__ double lm_reduce64(const double (&acc)[N], int lane, int *idx)
{
    static_assert(N <= 64, "one value per lane");
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8, b4 = lane & 16, b5 = lane & 32;
    double v1[32], v2[16], v3[8], v4[4], v5[2];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const double lo = j < N ? acc[j < N ? j : 0] : 0.0, hi = 32 + j < N ? acc[32 + j < N ? 32 + j : 0] : 0.0;
        const double keep = b0 ? hi : lo, send = b0 ? lo : hi;
        v1[j] = keep + ba_dpp_d(send, 1);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) { const double keep = b1 ? v1[16 + j] : v1[j], send = b1 ? v1[j] : v1[16 + j]; v2[j] = keep + ba_dpp_d(send, 0); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const double keep = b2 ? v2[8 + j] : v2[j], send = b2 ? v2[j] : v2[8 + j]; v3[j] = keep + __shfl_xor(send, 4); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const double keep = b3 ? v3[4 + j] : v3[j], send = b3 ? v3[j] : v3[4 + j]; v4[j] = keep + ba_ror8_d(send); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const double keep = b4 ? v4[2 + j] : v4[j], send = b4 ? v4[j] : v4[2 + j]; v5[j] = keep + __shfl_xor(send, 16); }
    const double keep = b5 ? v5[1] : v5[0], send = b5 ? v5[0] : v5[1];
    *idx = (b0 ? 32 : 0) + (b1 ? 16 : 0) + (b2 ? 8 : 0) + (b3 ? 4 : 0) + (b4 ? 2 : 0) + (b5 ? 1 : 0);
    return keep + __shfl_xor(send, 32);
}

// 1 / d to the last bit or two: v_rcp_f64 (about 2^-23) and two Newton steps -- five dependent instructions where the IEEE
// division is a chain of fifteen; the pivots of the reduced system are far from the limits of the exponent range
__device__ __forceinline__ double lm_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}

// The reduced pose system S x = b of one trial, n = 6 nb unknowns, by member 0 of a team: S = L D L^T by 6 x 6 BLOCK columns (nb <= 14
// of them, two workgroup barriers each, where the scalar right-looking form took 3 n), then the two triangular solves by one
// wavefront, again a block at a time.  Per block column: every thread factors the diagonal block for itself in registers (21 LDS
// broadcasts, no barrier); thread i takes row i of the panel below it (X = A L^-T D^-1, and T = X D kept aside for the update);
// the trailing matrix loses T X^T.  S holds the lower triangle, row-major with stride n; on return S holds L below the diagonal,
// bs holds x.  *fail is set when a pivot is not positive (the matrix is not positive definite: the trial is rejected).
__device__ __forceinline__ void lm_factor_blocked(double *S, double *dgv /*[n]*/, double *rdg /*[n]*/, double (*Tb)[6], int n, int *fail)
{
    const int tid = threadIdx.x, nb = n / 6;
    for (int kb = 0; kb < nb; ++kb) {
        const int j0 = 6 * kb, m = n - j0 - 6;
        double Lk[15], d[6], rd[6];                                  // unit lower triangle (r, c < r) at r (r - 1) / 2 + c
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double t[5];                                             // t[k] = L(c, k) d[k]
            double dc = S[(j0 + c) * n + j0 + c];
#pragma unroll
            for (int k = 0; k < c; ++k) { t[k] = Lk[c * (c - 1) / 2 + k] * d[k]; dc -= Lk[c * (c - 1) / 2 + k] * t[k]; }
            if (!(dc > 0) || !isfinite(dc)) { bad = true; dc = 1.0; }
            d[c] = dc; rd[c] = lm_rcp(dc);
#pragma unroll
            for (int r = c + 1; r < 6; ++r) {
                double a = S[(j0 + r) * n + j0 + c];
#pragma unroll
                for (int k = 0; k < c; ++k) a -= Lk[r * (r - 1) / 2 + k] * t[k];
                Lk[r * (r - 1) / 2 + c] = a * rd[c];
            }
        }
        if (m > 0 && tid < m) {                                      // panel: row i of the rows below the block
            const int i = j0 + 6 + tid;
            double T[6], X[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double a = S[i * n + j0 + c];
#pragma unroll
                for (int k = 0; k < c; ++k) a -= T[k] * Lk[c * (c - 1) / 2 + k];
                T[c] = a; X[c] = a * rd[c];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) { S[i * n + j0 + c] = X[c]; Tb[i][c] = T[c]; }
        }
        __syncthreads();                                             // every thread has read the diagonal block; the panel is in place
        if (tid == 0) {
            if (bad) *fail = 1;
#pragma unroll
            for (int r = 1; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < r; ++c) S[(j0 + r) * n + j0 + c] = Lk[r * (r - 1) / 2 + c];
#pragma unroll
            for (int c = 0; c < 6; ++c) { dgv[j0 + c] = d[c]; rdg[j0 + c] = rd[c]; }
        }
        if (m > 0) {                                                 // trailing matrix (its lower triangle); (ii, kk) steps by LM_THREADS without a division
            const int dq = LM_THREADS / m, dr = LM_THREADS - dq * m;
            int ii = tid / m, kk = tid - ii * m;
            for (; ii < m; ii += dq, kk += dr) {
                if (kk >= m) { kk -= m; ++ii; if (ii >= m) break; }
                if (kk > ii) continue;
                const int i = j0 + 6 + ii, k = j0 + 6 + kk;
                double a = S[i * n + k];
#pragma unroll
                for (int c = 0; c < 6; ++c) a -= Tb[i][c] * S[k * n + j0 + c];
                S[i * n + k] = a;
            }
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void lm_subst_blocked(const double *S, double *bs, const double *rdg, int n)
{
    const int tid = threadIdx.x, lane = tid & 63, nb = n / 6;
    if (tid < 64) {
#define LM_WSYNC_() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        for (int kb = 0; kb < nb; ++kb) {                            // L z = b
            const int j0 = 6 * kb;
            double z[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double a = bs[j0 + c];
#pragma unroll
                for (int k = 0; k < c; ++k) a -= S[(j0 + c) * n + j0 + k] * z[k];
                z[c] = a;
            }
            LM_WSYNC_()                                              // every lane has read b of the block
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) bs[j0 + c] = z[c];
            }
            for (int i = j0 + 6 + lane; i < n; i += 64) {
                double a = bs[i];
#pragma unroll
                for (int c = 0; c < 6; ++c) a -= S[i * n + j0 + c] * z[c];
                bs[i] = a;
            }
            LM_WSYNC_()
        }
        for (int i = lane; i < n; i += 64) bs[i] *= rdg[i];          // y = D^-1 z
        LM_WSYNC_()
        for (int kb = nb - 1; kb >= 0; --kb) {                       // L^T x = y
            const int j0 = 6 * kb;
            double x[6];
#pragma unroll
            for (int c = 5; c >= 0; --c) {
                double a = bs[j0 + c];
#pragma unroll
                for (int k = 5; k > c; --k) a -= S[(j0 + k) * n + j0 + c] * x[k];
                x[c] = a;
            }
            LM_WSYNC_()
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) bs[j0 + c] = x[c];
            }
            for (int i = lane; i < j0; i += 64) {
                double a = bs[i];
#pragma unroll
                for (int c = 0; c < 6; ++c) a -= S[(j0 + c) * n + i] * x[c];
                bs[i] = a;
            }
            LM_WSYNC_()
        }
#undef LM_WSYNC_
    }
}

__global__ __launch_bounds__(LM_THREADS) void k_ba_lm_team(LmTeamArgs A)
{
    extern __shared__ __attribute__((aligned(16))) double lm_dyn_S[];          // the reduced system, [n][n] with n = 6 Kf of THIS window (the launch sizes it for its largest)
    double *const S = lm_dyn_S;
    __shared__ double bs[LM_MAXN], xp[LM_MAXN], dg[LM_MAXN], rdg[LM_MAXN];   // dg / rdg: D of S = L D L^T and its reciprocals
    __shared__ double Tb[LM_MAXN][6];              // the panel of the current block column times D (lm_solve_blocked)
    __shared__ double sH[LM_MAXKF][28];            // Hpp upper triangle (21) + bp (6) of the free poses at the linearisation point
    __shared__ double s_posed[LM_LDSK][BA_POSED], s_posed_bk[LM_LDSK][BA_POSED];   // the prepared poses (q, t, R) ARE the pose state of the loop; backup for pop()
    __shared__ int32_t s_free_idx[LM_LDSK], s_free_pose[LM_LDSK];
    __shared__ uint8_t s_fixed[LM_LDSK];
    __shared__ int s_fail, s_ok, s_same;
    __shared__ long long s_t[16];
    const int G = A.G;
    // Placement of a team (the dispatcher deals block b to XCD b % 8 -- observed, not promised; nothing below depends on it for correctness):
    //   compact (default): the members of window w are blocks xslot + 8 j: ONE XCD -> the barrier without the L2 write-back, 10 % faster alone;
    //                      but the team then owns all 32 CUs of that XCD for the whole launch (a member's four wavefronts take a CU's
    //                      registers), and every kernel of another stream has workgroups dealt to that XCD: beside a team the tracking
    //                      kernels of the offline run made NO progress until the LM was done (profiles/r05_offline128_g1_timeline.md);
    //   spread (ygz_hip_ba_set_team_placement): consecutive blocks = consecutive members: four CUs of every XCD, the others stay free.
    int g, w;
    if (A.spread) { g = (int)(blockIdx.x % (unsigned)G); w = (int)(blockIdx.x / (unsigned)G); }
    else { const int xslot = blockIdx.x & 7, j = blockIdx.x >> 3; g = j % G; w = (j / G) * 8 + xslot; }
    if (w >= A.n_windows) return;
    ygz_raise_prio(A.prio);                                 // a latency chain of barriers and short phases on a few CUs
    BaDev B = A.wins[w];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (B.P <= 0 || B.E <= 0 || B.Kf <= 0) {
        // a degenerate window (device-built: no anchor feature with depth was seen in another keyframe): nothing to optimise.  Reported as a
        // finished run of ZERO iterations (every member leaves here, nobody waits at a barrier) -- not as twenty "iterations" of failed pivots
        if (g == 0 && tid == 0) {
            ygz_ba_stats st;
            st.iterations = 0; st.lm_trials = 0; st.chi2_initial = 0.0; st.chi2_final = 0.0; st.lambda_final = 0.0;
            A.stats[w] = st;
            *reinterpret_cast<ygz_ba_stats *>(B.lm_out) = st;
            for (int i = 4; i < 8; ++i) B.lm_out[i] = -1.0;
        }
        return;
    }
    long long t_prev = 0;
    if (A.dbg && blockIdx.x == 0 && tid == 0) { for (int i = 0; i < 16; ++i) s_t[i] = 0; t_prev = wall_clock64(); }
    const int K = B.K, P = B.P, Kf = B.Kf, n = 6 * Kf, Q = B.Q, npairs = Kf * (Kf + 1) / 2;
    const int wid = wv * G + g, nwv = G * LM_WAVES;            // this wavefront's place among the team's (member-minor: consecutive tasks on different CUs)
    unsigned char *scr = A.scratch + (size_t)w * A.stride;
    unsigned *bar = reinterpret_cast<unsigned *>(scr);
    double *xpub = reinterpret_cast<double *>(scr + LM_HDR);                   // [LM_MAXN + 2]
    double *crec = xpub + LM_MAXN + 2;                                          // [Qcap][LM_PARTW]: one record per chunk of 64 points
    double *Sp = crec + (size_t)A.Qcap * LM_PARTW;                              // [npairs][LM_V][LM_SPW]
    double *Sfin = Sp + (size_t)LM_NPAIR * LM_V * LM_SPW;                       // [npairs][LM_SPW]: the assembled blocks of the reduced system
    double *priv = Sfin + (size_t)LM_NPAIR * LM_SPW + (size_t)g * A.Kmax * (6 + 2 * BA_POSED);
    unsigned *pair_cnt = bar + 64;                                              // arrivals per pose pair (monotonic: LM_V per trial)
    double *my_poses = priv, *my_posed = priv + 6 * (size_t)A.Kmax, *posed_bk = my_posed + (size_t)BA_POSED * A.Kmax;
    double *const out_poses = B.poses_w;
    int32_t *const n_behind = B.n_behind;
    B.n_behind = reinterpret_cast<int32_t *>(bar + 4 + g);                     // the member's own count of edges behind the camera
    // the member's private pose state replaces the window's arrays in everything below
    for (int i = tid; i < 6 * K; i += LM_THREADS) my_poses[i] = B.poses[i];
    B.poses = my_poses; B.poses_w = my_poses; B.poses_bk = my_poses; B.posed = my_posed;
    // the per-pose tables every edge looks up through its pose index (prepared pose, free index, constant flag) live in LDS: in the
    // per-point edge loops they were a dependent global load per edge on top of the edge's own data
    if (K <= LM_LDSK) {
        for (int i = tid; i < K; i += LM_THREADS) { s_free_idx[i] = B.free_idx[i]; s_free_pose[i] = B.free_pose[i]; s_fixed[i] = B.fixed[i]; }
        B.posed = &s_posed[0][0]; B.free_idx = s_free_idx; B.free_pose = s_free_pose; B.fixed = s_fixed;
        posed_bk = &s_posed_bk[0][0];
    }
    unsigned epoch = 0;
    bool same_xcd = false;
    if (G > 1 && A.xcd_barrier && tid == 0) {
        // which XCD this member runs on (bar[2] = max id + 1, bar[3] = max (255 - id) + 1 over the members: equal ids <=> one XCD).  The
        // launch deals members of a team to blocks b, b + 8, b + 16 ... and the dispatcher deals block b to XCD b % 8 -- observed, not
        // promised: the light barrier is only taken when the register says so
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 255u;
        __hip_atomic_fetch_max(bar + 2, x + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(bar + 3, 256u - x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    // The pose state of the loop is the PREPARED pose T = (q, t, R): oplus is T <- exp(update) T (VertexSE3Sophus::oplusImpl, G2oTypes.h:38-45,
    // stores log(exp(update) exp(estimate)) and every later use takes exp of it again -- the same T up to the rounding of log and exp; two
    // SE3::exp and one SE3::log per trial and pose less on a path every member waits for).  The estimates are written once, at the end.
    if (tid < K) ba_pose_prep_one(B, tid);
    __syncthreads();

    double lambda = 0.0, ni = 2.0, currentChi = 0.0, chi_initial = 0.0;
    int iterations = 0, trials = 0;
#define LM_PART_RANGE(v) const int p0_ = 64 * (int)(((long long)(v) * Q) / LM_V), p1_ = min(P, 64 * (int)(((long long)((v) + 1) * Q) / LM_V));

    for (int it = 0; it < A.max_iterations; ++it) {
        // ---- computeActiveErrors + buildSystem: per chunk chi2, max |diag Hll|, the 27 sums of every free pose
        if (tid == 0) __hip_atomic_store(B.n_behind, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        // tasks of ONE WAVEFRONT each: (chunk c of 64 points, -1) = the edges of its points (residuals, Hll / bl, Hpl), (c, a) = the 27 sums of
        // free pose a over the chunk's points.  Task t goes to wavefront wid = t mod (4 G), member-minor: the Q heavy tasks land on Q
        // different CUs (one lane per point, a chain of dependent L2 round trips per edge: four chunks on one CU shared its memory pipeline),
        // the pose tasks fill the other wavefronts -- the helper members idled through this phase before.  Whoever runs a task, its sums are
        // formed inside one wavefront in a fixed order and combined in chunk order: the result does not depend on G.
        for (int t = wid; t < Q * (Kf + 1); t += nwv) {
            const bool heavy = t < Q;
            const int c = heavy ? t : (t - Q) / Kf, a = heavy ? -1 : (t - Q) - c * Kf;
            const int il = 64 * c + lane;
            double *cr = crec + (size_t)c * LM_PARTW;
            if (heavy) {
                double chi = 0.0, mx = 0.0;
                if (il < P) {
                    chi = ba_point_edges(B, il);
                    if (it == 0) for (int d = 0; d < 3; ++d) mx = fmax(mx, fabs(BA_PC(B.Hll_c, il, 9, 4 * d)));
                }
                chi = lm_wave_sum(chi);
                if (it == 0) {
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
                }
                if (lane == 0) { tl_st(cr, chi); if (it == 0) tl_st(cr + 1, mx); }
            } else {
                double acc[27];
#pragma unroll
                for (int i = 0; i < 27; ++i) acc[i] = 0.0;
                if (il < P) ba_pose_contrib(B, il, a, acc);
                int idx;                                                   // all 27 sums of the wavefront in one butterfly (ba_dev.h), fixed order
                const double tot = ba_reduce32(acc, lane, &idx);
                if (lane < 32 && idx < 27) tl_st(cr + 8 + 27 * a + idx, tot);
            }
        }
        LM_TICK(0);
        if (!lm_team_barrier(bar, epoch, G, &s_ok, same_xcd)) return;
        if (it == 0 && G > 1 && A.xcd_barrier) {                  // behind the first (full) barrier every member has published its XCD
            if (tid == 0) s_same = __hip_atomic_load(bar + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __hip_atomic_load(bar + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 257u;
            __syncthreads();
            same_xcd = s_same != 0;
        }
        LM_TICK(1);
        {   // every member adds the chunk records in chunk order: identical sH, chi2 (and lambda at the first iteration)
            for (int i = tid; i < 27 * Kf; i += LM_THREADS) sH[i / 27][i % 27] = lm_sum_records(crec + 8 + i, Q);
            double mx = it == 0 ? lm_max_records(crec + 1, Q) : 0.0;
            currentChi = lm_sum_records(crec, Q);
            __syncthreads();
            if (g == 0 && tid == LM_THREADS - 1) { int nb = 0; for (int m = 0; m < G; ++m) nb += (int)__hip_atomic_load(bar + 4 + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *n_behind = nb; }
            if (g == 0) for (int ti = tid; ti < 27 * Kf; ti += LM_THREADS) {   // the window's Hpp / bp as the ABI exposes them
                const int a = ti / 27, i = ti % 27, k = B.free_pose[a];
                if (i < 21) { int u = 0, rem = i; while (rem >= 6 - u) { rem -= 6 - u; ++u; } const int vv = u + rem;
                              B.Hpp[36 * (size_t)k + 6 * u + vv] = sH[a][i]; B.Hpp[36 * (size_t)k + 6 * vv + u] = sH[a][i]; }
                else B.bp[6 * (size_t)k + (i - 21)] = sH[a][i];
            }
            if (it == 0) {
                chi_initial = currentChi;
                for (int a = 0; a < Kf; ++a) { int q = 0; for (int u = 0; u < 6; ++u) { mx = fmax(mx, fabs(sH[a][q])); q += 6 - u; } }   // diagonal entries of the packed triangle
                lambda = 1e-5 * mx; ni = 2.0;
            }
        }
        double rho = 0.0; int qmax = 0;
        LM_TICK(2);
        do {
            // ---- push() (poses; the points are saved where they are updated).  There is no Dinv phase: the sweep inverts the point blocks itself
            for (int i = tid; i < BA_POSED * K; i += LM_THREADS) posed_bk[i] = B.posed[i];
            if (tid == 0) s_fail = 0;
            __syncthreads();
            LM_TICK(3);
            // ---- 2. one wavefront per (pose pair, part): that part's share of sum_l Y_a(l) W_b(l)^T and sum_l Y_a(l) b_l
            for (int task = wid; task < npairs * LM_V; task += nwv) {          // member-minor like the chunk tasks: 224 tasks leave 7 on every CU, not 8 on three quarters of them
                // tasks in the order diagonal pairs, first off-diagonal, second ...: the pairs near the diagonal share the most points and take
                // the longest, so the first round of tasks (one per wavefront) holds the long ones and the second the short ones
                const int ks = task / LM_V, v = task - ks * LM_V;
                LM_PART_RANGE(v)
                int dd = 0, a = ks;
                while (a >= Kf - dd) { a -= Kf - dd; ++dd; }
                const int b = a + dd;
                const int pr = a * Kf - a * (a - 1) / 2 + dd;              // the pair's place in the a-major enumeration (Sp, Sfin, arrival counters)
                double acc[36], accb[6];
                int bad = 0;
#pragma unroll
                for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i) accb[i] = 0.0;
                int l = p0_ + lane;
                int ca0 = l < p1_ ? B.ppc[(size_t)l * Kf + a] : -1, cb0 = l < p1_ ? B.ppc[(size_t)l * Kf + b] : -1;
                int ca1 = l + 64 < p1_ ? B.ppc[(size_t)(l + 64) * Kf + a] : -1, cb1 = l + 64 < p1_ ? B.ppc[(size_t)(l + 64) * Kf + b] : -1;
                for (int base = p0_; base < p1_; base += 128, l += 128) {
                    const int na0 = l + 128 < p1_ ? B.ppc[(size_t)(l + 128) * Kf + a] : -1, nb0 = l + 128 < p1_ ? B.ppc[(size_t)(l + 128) * Kf + b] : -1;
                    const int na1 = l + 192 < p1_ ? B.ppc[(size_t)(l + 192) * Kf + a] : -1, nb1 = l + 192 < p1_ ? B.ppc[(size_t)(l + 192) * Kf + b] : -1;
                    lm_sweep_point<false, 2>(B, l, ca0, cb0, a == b, nullptr, nullptr, acc, accb, lambda, &bad);
                    lm_sweep_point<false, 2>(B, l + 64, ca1, cb1, a == b, nullptr, nullptr, acc, accb, lambda, &bad);
                    ca0 = na0; cb0 = nb0; ca1 = na1; cb1 = nb1;
                }
                if (__ballot(bad != 0) != 0ull && lane == 0) tl_st(xpub + LM_MAXN + 1, (double)(trials + 1));   // a singular point block: every member rejects this trial (the trial's number: nobody has to clear it)
                // the 42 sums of the wavefront in one butterfly, one coalesced store; the wavefront that delivers the LAST part of a pair
                // adds the parts -- in part order, whoever it is -- and leaves the finished block of S (and of the right-hand side) in Sfin
                double all[LM_SPW];
#pragma unroll
                for (int i = 0; i < 36; ++i) all[i] = acc[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) all[36 + i] = accb[i];
                int idx;
                const double tot = lm_reduce64<LM_SPW>(all, lane, &idx);
                if (idx < LM_SPW) tl_st(Sp + ((size_t)pr * LM_V + v) * LM_SPW + idx, tot);
                // the parts travel through agent-scope stores and loads (write-through / re-read at the scope of the device) and have been
                // acknowledged (vmcnt) before the arrival is counted: no agent-scope fence here -- a release writes the WHOLE L2 of the XCD
                // back and an acquire drops its lines, per task and for every kernel that shares the XCD (measured on the offline run:
                // + 2.4 ms of tracking per 1024 frames beside the two BA launches)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                unsigned arrived = 0;
                if (lane == 0) arrived = __hip_atomic_fetch_add(pair_cnt + pr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived);
                if ((arrived % LM_V) == LM_V - 1) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (lane < LM_SPW) {
                        double pv[LM_V], sub = 0.0;                            // all parts requested before the first is added
#pragma unroll
                        for (int vv = 0; vv < LM_V; ++vv) pv[vv] = tl_ld(Sp + ((size_t)pr * LM_V + vv) * LM_SPW + lane);
#pragma unroll
                        for (int vv = 0; vv < LM_V; ++vv) sub += pv[vv];
                        double t = 0.0;
                        if (a == b) {
                            if (lane < 36) { const int rr = lane / 6, cc = lane - 6 * rr, u = min(rr, cc), w2 = max(rr, cc);
                                             t = sH[a][u * 6 - u * (u - 1) / 2 + (w2 - u)]; if (rr == cc) t += lambda; }
                            else t = sH[a][21 + (lane - 36)];
                        }
                        tl_st(Sfin + (size_t)pr * LM_SPW + lane, t - sub);
                    }
                }
            }
            // (two pairs per wavefront walking the points together -- to overlap their load chains -- needs 84 accumulators: 110 spilled
            // registers, 2040 instead of 1680 us per 24 sweeps)
            LM_TICK(5);
            if (!lm_team_barrier(bar, epoch, G, &s_ok, same_xcd)) return;
            LM_TICK(6);
            // ---- 3. EVERY member: the blocks of S from Sfin, L D L^T by block columns, substitutions -- the same 1176 values and the same
            //         arithmetic everywhere, so x_p needs no publishing and no team barrier (member 0 alone solved while 31 members waited for
            //         it and then for the barrier behind it: the wait was the same, the barrier came on top)
            {
                // (plain loads: the team barrier has made the finished blocks visible like every other array the members share)
                for (int i0 = tid; i0 < npairs * LM_SPW; i0 += 6 * LM_THREADS) {   // six loads in flight per thread (n = 42: 1176 values, one round)
                    double val[6];
#pragma unroll
                    for (int u = 0; u < 6; ++u) { const int i = i0 + u * LM_THREADS; val[u] = i < npairs * LM_SPW ? Sfin[i] : 0.0; }
#pragma unroll
                    for (int u = 0; u < 6; ++u) {                              // block (b <= a) of pair pr: rows of b x columns of a -> the LOWER triangle of S
                        const int i = i0 + u * LM_THREADS;
                        if (i >= npairs * LM_SPW) break;
                        const int pr = i / LM_SPW, e = i - pr * LM_SPW;
                        int b = 0, rem = pr;
                        while (rem >= Kf - b) { rem -= Kf - b; ++b; }
                        const int a = b + rem;
                        if (e >= 36) { if (a == b) bs[6 * a + (e - 36)] = val[u]; continue; }
                        const int x = e / 6, y = e - 6 * x;                    // entry (6 b + x, 6 a + y) = (6 a + y, 6 b + x)
                        if (a == b) { if (x >= y) S[(6 * a + x) * n + 6 * a + y] = val[u]; }
                        else S[(6 * a + y) * n + 6 * b + x] = val[u];
                    }
                }
                if (tid == 0) s_fail = xpub[LM_MAXN + 1] == (double)(trials + 1);      // a sweep task of THIS trial met a singular point block
                __syncthreads();
                LM_TICK(7);
                lm_factor_blocked(S, dg, rdg, Tb, n, &s_fail);
                LM_TICK(8);
                lm_subst_blocked(S, bs, rdg, n);
            }
            LM_TICK(9);
            __syncthreads();
            for (int i = tid; i < n; i += LM_THREADS) xp[i] = bs[i];
            const bool ok2 = s_fail == 0;
            __syncthreads();
            LM_TICK(10);
            // ---- 4. update(x): T <- exp(x_p) T, x_l per point; computeScale; then computeActiveErrors at the trial state -- per chunk
            if (ok2 && tid < Kf) {
                double *o = B.posed + BA_POSED * (size_t)B.free_pose[tid];
                const double v[6] = { xp[6 * tid + 3], xp[6 * tid + 4], xp[6 * tid + 5], xp[6 * tid], xp[6 * tid + 1], xp[6 * tid + 2] };   // [omega; t] -> [t; omega]
                Se3 U, T0, T1;
                se3_exp_d(v, &U);
                for (int d = 0; d < 4; ++d) T0.q[d] = o[d];
                for (int d = 0; d < 3; ++d) T0.t[d] = o[4 + d];
                se3_mul_d(&U, &T0, &T1);
                for (int d = 0; d < 4; ++d) o[d] = T1.q[d];
                for (int d = 0; d < 3; ++d) o[4 + d] = T1.t[d];
                quat_to_R_d(T1.q, o + 7);
            }
            __syncthreads();
            LM_TICK(14);
            for (int c = wid; c < Q; c += nwv) {                       // one wavefront per chunk of 64 points
                const int il = 64 * c + lane;
                double scale = 0.0, chi = 0.0;
                if (ok2) {
                    if (il < P) {
                        for (int d = 0; d < 3; ++d) B.points_bk[3 * (size_t)il + d] = B.points_w[3 * (size_t)il + d];
                        if (B.point_fixed[il]) { B.xl[3 * (size_t)il] = B.xl[3 * (size_t)il + 1] = B.xl[3 * (size_t)il + 2] = 0.0; }
                        else {
                            double r3[3] = { BA_PC(B.bl_c, il, 3, 0), BA_PC(B.bl_c, il, 3, 1), BA_PC(B.bl_c, il, 3, 2) };
                            const int ln = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
                            for (int c0 = 0; c0 < rows; c0 += LM_RB) {         // the blocks of LM_RB rows requested together (see step 1), used in row order
                                double Wg[LM_RB][18]; int ipg[LM_RB];
#pragma unroll
                                for (int u = 0; u < LM_RB; ++u) {
                                    const bool in_ = c0 + u < rows;
                                    ipg[u] = in_ ? B.pose_c[(size_t)(row0 + c0 + u) * 64 + ln] : -1;
#pragma unroll
                                    for (int i = 0; i < 18; ++i) Wg[u][i] = in_ ? BA_EC(B.Hpl_c, row0 + c0 + u, 18, i, ln) : 0.0;
                                }
#pragma unroll
                                for (int u = 0; u < LM_RB; ++u) {
                                    if (ipg[u] < 0) continue;
                                    const int a = B.free_idx[ipg[u]];
                                    if (a < 0) continue;
                                    for (int cc = 0; cc < 3; ++cc) for (int r = 0; r < 6; ++r) r3[cc] -= Wg[u][3 * r + cc] * xp[6 * a + r];
                                }
                            }
                            double D[9], Di[9];                                // the inverse the sweep used (the same expressions)
                            for (int i = 0; i < 9; ++i) D[i] = BA_PC(B.Hll_c, il, 9, i);
                            D[0] += lambda; D[4] += lambda; D[8] += lambda;
                            if (!lm_inv3(D, Di)) { for (int i = 0; i < 9; ++i) Di[i] = 0.0; chi += __longlong_as_double(0x7FF8000000000000ll); }   // (a point no pose pair covers: the trial is rejected through its chi2)
                            double x3[3];
                            for (int cc = 0; cc < 3; ++cc) x3[cc] = Di[3 * cc] * r3[0] + Di[3 * cc + 1] * r3[1] + Di[3 * cc + 2] * r3[2];
                            for (int cc = 0; cc < 3; ++cc) {
                                B.xl[3 * (size_t)il + cc] = x3[cc];
                                scale += x3[cc] * (lambda * x3[cc] + BA_PC(B.bl_c, il, 3, cc));
                                B.points_w[3 * (size_t)il + cc] += x3[cc];
                            }
                        }
                        chi += lm_point_chi2_pf(B, il);                          // the lane's own point at its trial position
                    }
                }
                scale = lm_wave_sum(scale);
                chi = lm_wave_sum(chi);
                if (lane == 0) { tl_st(crec + (size_t)c * LM_PARTW + 3, scale); tl_st(crec + (size_t)c * LM_PARTW + 4, chi); }
            }
            LM_TICK(15);
            LM_TICK(11);
            if (!lm_team_barrier(bar, epoch, G, &s_ok, same_xcd)) return;
            LM_TICK(12);
            double scale = 0.0, tempChi = DBL_MAX;
            if (ok2) {
                tempChi = 0.0;
                lm_sum_records2(crec + 3, crec + 4, Q, &scale, &tempChi);
                for (int a = 0; a < Kf; ++a) for (int d = 0; d < 6; ++d) scale += xp[6 * a + d] * (lambda * xp[6 * a + d] + sH[a][21 + d]);
            }
            rho = (currentChi - tempChi) / (scale + 1e-3);
            ++trials;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2; currentChi = tempChi;
            } else {                                                             // pop(): the member's own points and its pose copy
                lambda *= ni; ni *= 2;
                __syncthreads();
                for (int i = tid; i < BA_POSED * K; i += LM_THREADS) B.posed[i] = posed_bk[i];
                for (int c = wid; ok2 && c < Q; c += nwv) {                         // (a trial whose system could not be solved did not touch the points)
                    const int il = 64 * c + lane;
                    if (il < P) for (int d = 0; d < 3; ++d) B.points_w[3 * (size_t)il + d] = B.points_bk[3 * (size_t)il + d];
                }
                __syncthreads();
                if (!isfinite(lambda)) break;
            }
            qmax++;
            LM_TICK(13);
        } while (rho < 0 && qmax < 10);
        ++iterations;
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) break;
    }
    if (A.dbg && blockIdx.x == 0 && tid == 0) for (int i = 0; i < 16; ++i) A.dbg[i] = s_t[i];
    if (g == 0) {
        __syncthreads();
        for (int i = tid; i < 6 * K; i += LM_THREADS) out_poses[i] = my_poses[i];          // the constant poses as they came
        __syncthreads();
        if (tid < Kf) {                                                                       // the estimates of the free poses: log T, [omega; t]
            const int k = B.free_pose[tid];
            const double *o = B.posed + BA_POSED * (size_t)k;
            Se3 T; double r[6];
            for (int d = 0; d < 4; ++d) T.q[d] = o[d];
            for (int d = 0; d < 3; ++d) T.t[d] = o[4 + d];
            se3_log_d(&T, r);
            for (int d = 0; d < 3; ++d) { out_poses[6 * (size_t)k + d] = r[3 + d]; out_poses[6 * (size_t)k + 3 + d] = r[d]; }
        }
        if (tid == 0) {
            ygz_ba_stats st;
            st.iterations = iterations; st.lm_trials = trials; st.chi2_initial = chi_initial; st.chi2_final = currentChi; st.lambda_final = lambda;
            A.stats[w] = st;
            *reinterpret_cast<ygz_ba_stats *>(B.lm_out) = st;               // kept with the window (ygz_hip_ba_get_stats)
            for (int i = 4; i < 8; ++i) B.lm_out[i] = -1.0;                 // the outlier record of this state is not computed yet (ygz_hip_ba_mark_outliers)
        }
    }
#undef LM_PART_RANGE
}

// ---- the inlier statistics after optimize() (BA.cpp:503-515): every edge's UNROBUSTIFIED chi2 at the final state against the threshold
// (5.991); an edge above it is an outlier -- the reference sets Feature::_bad, which takes the observation out of every later BA
// (BA.cpp:436).  Here: counted, summed, and with disable != 0 switched off (enable = 0) so that another optimisation of the same window
// runs without it.  One workgroup per window; sums in a fixed order.  lm_out[4..7] = edges tested, outliers, chi2 of all tested edges,
// chi2 of the inliers.
struct OutlierArgs { const BaDev *wins; double thr; int disable; };
__global__ __launch_bounds__(256) void k_ba_outliers(OutlierArgs A)
{
    __shared__ double red[4][256];
    BaDev B = A.wins[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid < B.K) ba_pose_prep_one(B, tid);
    __syncthreads();
    double n_en = 0.0, n_out = 0.0, chi_all = 0.0, chi_in = 0.0;
    uint8_t *enable_c = const_cast<uint8_t *>(B.enable_c);
    for (int il = tid; il < B.P; il += 256) {
        const int lane = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
        const double pt[3] = { B.points[3 * (size_t)il], B.points[3 * (size_t)il + 1], B.points[3 * (size_t)il + 2] };
        for (int c = 0; c < rows; ++c) {
            const int row = row0 + c, ip = B.pose_c[(size_t)row * 64 + lane];
            if (ip < 0 || !B.enable_c[(size_t)row * 64 + lane]) continue;
            double p[3], r[2];
            ba_project(B, B.posed + BA_POSED * (size_t)ip, pt, BA_EC(B.obs_c, row, 2, 0, lane), BA_EC(B.obs_c, row, 2, 1, lane), p, r);
            const double e2 = r[0] * r[0] + r[1] * r[1];                       // edge->chi2() with identity information (BA.cpp:509)
            n_en += 1.0; chi_all += e2;
            if (e2 > A.thr) { n_out += 1.0; if (A.disable) enable_c[(size_t)row * 64 + lane] = 0; }
            else chi_in += e2;
        }
    }
    red[0][tid] = n_en; red[1][tid] = n_out; red[2][tid] = chi_all; red[3][tid] = chi_in;
    __syncthreads();
    if (tid < 4) { double t = 0.0; for (int i = 0; i < 256; ++i) t += red[tid][i]; B.lm_out[4 + tid] = t; }
}


// ---- self-test of the light (same-XCD) barrier, once per context before the first team launch.  The light form leaves out the
// agent-scope release and relies on three things the HSA memory model does not promise: HW_REG_XCC_ID names the XCD a workgroup runs on, the
// workgroups of one XCD share its L2, and a store is visible there once vmcnt has counted it.  All three hold on the MI355X boxes this was
// measured on (SPX mode); firmware, another partition mode or another part may differ, and a wrong assumption would be a SILENT stale read in
// the LM.  So: a message-passing litmus through lm_team_barrier itself -- pairs of workgroups (b, b + 8: the dispatcher's same-XCD pairs; pairs
// that land on different XCDs take the full barrier and count as such), 256 rounds each way: the writer stores a line of round-dependent words
// with plain stores, both take the barrier, the reader checks every word.  Any stale word (or no same-XCD pair at all) switches the light form
// off for the context; YGZ_LM_XCD_BARRIER=0 / 1 overrides the test either way.
__global__ __launch_bounds__(256) void k_lm_barrier_selftest(unsigned *__restrict__ bars, unsigned *__restrict__ data, unsigned *__restrict__ result)
{
    __shared__ int s_ok, s_same;
    const int pair = blockIdx.x & 7, member = blockIdx.x >> 3, tid = threadIdx.x;       // blocks b and b + 8 form pair b
    unsigned *bar = bars + 64 * pair, *buf = data + 1024 * pair;
    unsigned epoch = 0;
    if (tid == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 255u;
        __hip_atomic_fetch_max(bar + 2, x + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(bar + 3, 256u - x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!lm_team_barrier(bar, epoch, 2, &s_ok, false)) { if (tid == 0) atomicAdd(result + 2, 1u); return; }
    if (tid == 0) s_same = __hip_atomic_load(bar + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __hip_atomic_load(bar + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 257u;
    __syncthreads();
    const bool same = s_same != 0;
    unsigned stale = 0;
    for (unsigned r = 1; r <= 512; ++r) {
        const int writer = (int)(r & 1u);
        if (member == writer) for (int i = tid; i < 1024; i += 256) buf[i] = r * 0x9E3779B1u + (unsigned)i;      // plain stores, as the LM's tl_st
        if (!lm_team_barrier(bar, epoch, 2, &s_ok, same)) { if (tid == 0) atomicAdd(result + 2, 1u); return; }
        if (member != writer) for (int i = tid; i < 1024; i += 256) stale += buf[i] != r * 0x9E3779B1u + (unsigned)i;
        if (!lm_team_barrier(bar, epoch, 2, &s_ok, same)) { if (tid == 0) atomicAdd(result + 2, 1u); return; }  // the reader is done before the next round overwrites
    }
    if (stale) atomicAdd(result + 1, stale);
    if (same && member == 0 && tid == 0) atomicAdd(result, 1u);                                                     // pairs that exercised the light form
}

static int lm_light_barrier_allowed(ygz_hip_ctx *ctx)
{
    static const int forced = [] { const char *e = getenv("YGZ_LM_XCD_BARRIER"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }();
    if (forced >= 0) return ctx->lm_light_barrier = forced;
    if (ctx->lm_light_barrier >= 0) return ctx->lm_light_barrier;
    void *d = nullptr;
    const size_t bytes = (8 * 64 + 8 * 1024 + 4) * sizeof(unsigned);
    if (ygz_scratch(ctx, SCR_GEN_0 + 5, bytes, &d) != YGZ_OK) return ctx->lm_light_barrier = 0;
    unsigned *bars = (unsigned *)d, *data = bars + 8 * 64, *result = data + 8 * 1024;
    unsigned res[4] = { 0, 1, 1, 0 };
    if (hipMemsetAsync(d, 0, bytes, ctx->stream) == hipSuccess) {
        k_lm_barrier_selftest<<<dim3(16), dim3(256), 0, ctx->stream>>>(bars, data, result);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(res, result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { res[0] = 0; res[1] = 1; }
    }
    // passed: at least one pair ran the light form and nobody saw a stale word or a time-out
    return ctx->lm_light_barrier = (res[0] > 0 && res[1] == 0 && res[2] == 0) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_ba_lm_reset(const BaDev *__restrict__ wins, unsigned char *__restrict__ scratch, size_t stride,
                                                       ygz_ba_stats *__restrict__ stats)
{
    static_assert(LM_HDR % (256 * 4) == 0 && sizeof(ygz_ba_stats) == 32, "whole dwords per thread; four 8-byte words per record");
    const int w = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < LM_HDR / 4; i += 256) reinterpret_cast<uint32_t *>(scratch + (size_t)w * stride)[i] = 0u;
    if (tid < 4) reinterpret_cast<unsigned long long *>(stats + w)[tid] = ~0ull;
    else if (tid < 8) reinterpret_cast<unsigned long long *>(wins[w].lm_out)[tid - 4] = 0xFEFEFEFEFEFEFEFEull;
    else if (tid == 8) reinterpret_cast<double *>(scratch + (size_t)w * stride + LM_HDR)[LM_MAXN + 1] = 0.0;   // "a point block was singular" (set by the sweep, cleared by member 0)
}

// the per-window records of a range in one piece: lm_out[0..8) (statistics of the last resident run, outlier record) and the sizes of the
// graph as the device table holds them -> rec[w][LM_REC] (ygz_hip_ba_get_stats / ygz_hip_ba_get_outlier_stats: one copy instead of one per window)
#define LM_REC 12
__global__ void k_ba_records(const BaDev *__restrict__ wins, int n, double *__restrict__ rec)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    const BaDev &B = wins[w];
    for (int i = 0; i < 8; ++i) rec[(size_t)w * LM_REC + i] = B.lm_out[i];
    rec[(size_t)w * LM_REC + 8] = (double)B.K; rec[(size_t)w * LM_REC + 9] = (double)B.P;
    rec[(size_t)w * LM_REC + 10] = (double)B.E; rec[(size_t)w * LM_REC + 11] = (double)B.Kf;
}

extern "C" {

int ygz_hip_ba_set_team_budget(ygz_hip_ctx *ctx, int workgroups)
{
    if (!ctx || workgroups < 0) return YGZ_E_INVALID;
    ctx->lm_team_budget = workgroups > ctx->n_cu ? ctx->n_cu : workgroups;
    return YGZ_OK;
}

int ygz_hip_ba_set_team_placement(ygz_hip_ctx *ctx, int spread)
{
    if (!ctx) return YGZ_E_INVALID;
    ctx->lm_spread = spread != 0;
    return YGZ_OK;
}

int ygz_hip_ba_optimize_resident(ygz_hip_ctx *ctx, int window_begin, int n_windows, int max_iterations, ygz_ba_stats *stats)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || window_begin < 0 || n_windows < 1 || window_begin + n_windows > (int)ctx->ba.size() || max_iterations < 0) return YGZ_E_INVALID;
    for (int i = window_begin; i < window_begin + n_windows; ++i) {
        if (!ctx->ba[i]) return YGZ_E_INVALID;
        if (ctx->ba[i]->formulation != 0) return YGZ_E_INVALID;      // the g2o path of the live tree
        if (ctx->ba[i]->Kf > LM_MAXKF || ctx->ba[i]->K > LM_THREADS) return YGZ_E_CAPACITY;   // reduced system must fit LDS
        if (ctx->ba[i]->has_dup) return YGZ_E_INVALID;            // the pair sweep of the Schur step takes ONE edge per (point, pose): use ygz_hip_ba_optimize
    }
    int rc = YGZ_OK;
    const BaDev *table = ygz_ba_table(ctx, &rc);
    if (!table) return rc;
    // team size: windows x G <= half of the device's CUs by default (every member spins at the team barriers, so all of them must be
    // resident together -- also beside the kernels of other streams and of the offline run's tracking lanes), G a power of two
    int G = 1, Kmax = 1;
    static const bool single = [] { const char *e = getenv("YGZ_BA_LM_TEAM"); return e && e[0] == '1' && e[1] == 0; }();   // A/B switch: one workgroup per window
    const int wg_budget = ctx->lm_team_budget > 0 ? ctx->lm_team_budget : (ctx->n_cu / 2 > 8 ? ctx->n_cu / 2 : 8);
    // up to LM_MAXG members: every phase is a pool of one-wavefront tasks (chunks of points, (chunk, pose) sums, (pose pair, part) sweeps:
    // 28 pairs x 8 parts over the 128 wavefronts of 32 members are 1.75 rounds of the longest phase of a trial)
    if (!single) while (G < LM_MAXG && n_windows * (2 * G) <= wg_budget) G *= 2;           // windows x G <= budget
    for (int i = window_begin; i < window_begin + n_windows; ++i) Kmax = ctx->ba[i]->K > Kmax ? ctx->ba[i]->K : Kmax;
    int Qcap = 1;
    for (int i = window_begin; i < window_begin + n_windows; ++i) Qcap = std::max(Qcap, (ctx->ba[i]->P + 63) / 64);      // (host fields: the capacities)
    size_t stride = LM_HDR + sizeof(double) * ((size_t)LM_MAXN + 2 + (size_t)Qcap * LM_PARTW + (size_t)LM_NPAIR * LM_V * LM_SPW + (size_t)LM_NPAIR * LM_SPW
                                           + (size_t)G * Kmax * (6 + 2 * BA_POSED));
    stride = (stride + 255) & ~(size_t)255;
    const size_t stats_bytes = (((size_t)n_windows * sizeof(ygz_ba_stats)) + 255) & ~(size_t)255;
    void *d_scr = nullptr;
    if ((rc = ygz_scratch(ctx, SCR_BA_0, stats_bytes + (size_t)n_windows * stride, &d_scr)) != YGZ_OK) return rc;
    LmTeamArgs A;
    A.wins = table + window_begin; A.n_windows = n_windows; A.G = G; A.max_iterations = max_iterations; A.stats = (ygz_ba_stats *)d_scr;
    A.scratch = (unsigned char *)d_scr + stats_bytes; A.stride = stride; A.Kmax = Kmax; A.Qcap = Qcap;
    A.dbg = nullptr; A.prio = (ctx->wave_prio_mask >> 3) & 1; A.spread = ctx->lm_spread ? 1 : 0;
    A.xcd_barrier = lm_light_barrier_allowed(ctx);               // self-tested once per context (k_lm_barrier_selftest); YGZ_LM_XCD_BARRIER=0 / 1 overrides
    static const bool lm_debug = getenv("YGZ_LM_DEBUG") != nullptr;
    if (lm_debug) { void *d = nullptr; if (ygz_scratch(ctx, SCR_GEN_0 + 4, 16 * 8, &d) == YGZ_OK) A.dbg = (long long *)d; }
    YgzAuxScope aux(ctx, 1);
    // barrier counters and abort flags zeroed, iterations = -1 until a team finishes (0xFF in the launch's statistics, 0xFE in the windows' own
    // records; 0xFF there = never run, ba_carve): one small launch (a 2-D memset, a memset and one memset per window took 0.15 ms of the
    // serial tail of an offline run)
    YGZ_LAUNCH(ctx, KID_BA_LM, k_ba_lm_reset, dim3(n_windows), dim3(256), A.wins, A.scratch, stride, A.stats);
    // the reduced system of the launch's largest window as dynamic LDS: 14 KB for the 7 free poses of an offline window (the static 56 KB of
    // rounds 3-4 held 14 poses whatever the window had; beside a team other kernels now find that much more LDS on its CUs), 115 KB for 20
    int nmax = 6;
    for (int i = window_begin; i < window_begin + n_windows; ++i) nmax = std::max(nmax, 6 * ctx->ba[i]->Kf);     // (host field: the capacity)
    const size_t dyn_lds = (size_t)nmax * nmax * sizeof(double);
    void (*const k_team)(LmTeamArgs) = k_ba_lm_team;
    if (!ctx->lm_attr_set) {
        YGZ_HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_team), hipFuncAttributeMaxDynamicSharedMemorySize, LM_MAXN * LM_MAXN * (int)sizeof(double)));
        ctx->lm_attr_set = true;
    }
    YGZ_LAUNCH_DYN(ctx, KID_BA_LM, k_team, dim3(A.spread ? G * n_windows : 8 * G * ((n_windows + 7) / 8)), dim3(LM_THREADS), dyn_lds, A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    if (A.dbg) {
        long long h[16];
        static const char *const nm[16] = { "linearise + pose sums", "barrier", "combine parts", "Dinv, Y = Hpl Dinv", "barrier", "Schur sweep", "barrier",
                                            "load S", "L D L^T", "substitutions", "barrier", "block sums of the update", "barrier", "accept / reject",
                                            "pose update (oplus, SE3::exp)", "point update + trial chi2" };
        YGZ_HIPCHK(ctx, hipMemcpyAsync(h, A.dbg, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        long long tot = 0; for (int i = 0; i < 16; ++i) tot += h[i];
        fprintf(stderr, "[lm-debug] %d windows, team of %d: member 0 of window 0, %.1f us in all:", n_windows, G, tot * 0.01);
        for (int i = 0; i < 16; ++i) fprintf(stderr, " %s %.1f;", nm[i], h[i] * 0.01);
        fprintf(stderr, "\n");
    }
    if (stats) {
        // a team whose member never reached a barrier (not co-resident within the spin bound) leaves without writing its record: the
        // caller gets YGZ_E_HIP whether it asked for statistics here or reads them later with ygz_hip_ba_get_stats
        YGZ_HIPCHK(ctx, hipMemcpyAsync(stats, d_scr, (size_t)n_windows * sizeof(ygz_ba_stats), hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < n_windows; ++i)
            if (stats[i].iterations < 0) { ctx->last_hip_error = (int)hipErrorLaunchTimeOut; return YGZ_E_HIP; }
    }
    return YGZ_OK;
}

// statistics of the last ygz_hip_ba_optimize_resident run of every window in the range (which may have been asynchronous), and the
// actual graph sizes dims [n][4] = K, P, E, free poses (what ygz_hip_ba_build_windows assembled; the uploaded sizes otherwise).
// YGZ_E_HIP when a window's team timed out at a barrier, YGZ_E_STATE when no resident LM has run on one.  Synchronises.
int ygz_hip_ba_get_stats(ygz_hip_ctx *ctx, int window_begin, int n_windows, ygz_ba_stats *stats, int32_t *dims)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || window_begin < 0 || n_windows < 1 || window_begin + n_windows > (int)ctx->ba.size() || (!stats && !dims)) return YGZ_E_INVALID;
    for (int i = window_begin; i < window_begin + n_windows; ++i) if (!ctx->ba[i]) return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    int rc = YGZ_OK;
    const BaDev *table = ygz_ba_table(ctx, &rc);
    if (!table) return rc;
    // one gather kernel and ONE copy for the whole range (a copy per window cost 20 us each behind the last LM launch of a run)
    double *d_rec = nullptr;
    if ((rc = ygz_scratch(ctx, SCR_GEN_0 + 5, (size_t)n_windows * LM_REC * 8, (void **)&d_rec)) != YGZ_OK) return rc;
    YGZ_LAUNCH(ctx, KID_BA_LM, k_ba_records, dim3(ygz_div_up(n_windows, 64)), dim3(64), table + window_begin, n_windows, d_rec);
    std::vector<double> hr((size_t)n_windows * LM_REC);
    YGZ_HIPCHK(ctx, hipMemcpyAsync(hr.data(), d_rec, hr.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_windows; ++i) {
        const double *r = hr.data() + (size_t)i * LM_REC;
        if (stats) memcpy(stats + i, r, sizeof(ygz_ba_stats));
        if (dims) { dims[4 * i] = (int32_t)r[8]; dims[4 * i + 1] = (int32_t)r[9]; dims[4 * i + 2] = (int32_t)r[10]; dims[4 * i + 3] = (int32_t)r[11]; }
    }
    if (stats) for (int i = 0; i < n_windows; ++i) if (stats[i].iterations < 0) { ctx->last_hip_error = (int)hipErrorLaunchTimeOut; return stats[i].lm_trials == -1 ? YGZ_E_STATE : YGZ_E_HIP; }
    return YGZ_OK;
}

// BA.cpp:503-515 for the windows of the range, behind whatever optimised them (asynchronous): see k_ba_outliers.  disable != 0 switches the
// outlier edges off for later runs on the same windows (the effect of Feature::_bad = true on the next LocalBAG2O, BA.cpp:436).
int ygz_hip_ba_mark_outliers(ygz_hip_ctx *ctx, int window_begin, int n_windows, double chi2_threshold, int disable)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || window_begin < 0 || n_windows < 1 || window_begin + n_windows > (int)ctx->ba.size() || !(chi2_threshold >= 0)) return YGZ_E_INVALID;
    for (int i = window_begin; i < window_begin + n_windows; ++i) if (!ctx->ba[i] || ctx->ba[i]->K > 256) return YGZ_E_INVALID;
    int rc = YGZ_OK;
    const BaDev *table = ygz_ba_table(ctx, &rc);
    if (!table) return rc;
    OutlierArgs A; A.wins = table + window_begin; A.thr = chi2_threshold; A.disable = disable;
    YgzAuxScope aux(ctx, 1);                                   // the stream the resident LM runs on
    YGZ_LAUNCH(ctx, KID_BA_LM, k_ba_outliers, dim3(n_windows), dim3(256), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

// out [n_windows][4]: edges tested, outliers, chi2 of all tested edges, chi2 of the inliers, as ygz_hip_ba_mark_outliers left them
// (-1: not computed for the window's current state).  Synchronises.
int ygz_hip_ba_get_outlier_stats(ygz_hip_ctx *ctx, int window_begin, int n_windows, double *out)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !out || window_begin < 0 || n_windows < 1 || window_begin + n_windows > (int)ctx->ba.size()) return YGZ_E_INVALID;
    for (int i = window_begin; i < window_begin + n_windows; ++i) if (!ctx->ba[i]) return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    int rc = YGZ_OK;
    const BaDev *table = ygz_ba_table(ctx, &rc);
    if (!table) return rc;
    double *d_rec = nullptr;
    if ((rc = ygz_scratch(ctx, SCR_GEN_0 + 5, (size_t)n_windows * LM_REC * 8, (void **)&d_rec)) != YGZ_OK) return rc;
    YGZ_LAUNCH(ctx, KID_BA_LM, k_ba_records, dim3(ygz_div_up(n_windows, 64)), dim3(64), table + window_begin, n_windows, d_rec);
    std::vector<double> hr((size_t)n_windows * LM_REC);
    YGZ_HIPCHK(ctx, hipMemcpyAsync(hr.data(), d_rec, hr.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_windows; ++i) for (int k = 0; k < 4; ++k) out[4 * (size_t)i + k] = hr[(size_t)i * LM_REC + 4 + k];
    return YGZ_OK;
}

int ygz_hip_ba_get_state(ygz_hip_ctx *ctx, int window, double *poses, double *points)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || window < 0 || window >= (int)ctx->ba.size() || !ctx->ba[window]) return YGZ_E_INVALID;
    auto *w = ctx->ba[window];
    if (poses) YGZ_HIPCHK(ctx, hipMemcpyAsync(poses, w->poses, (size_t)w->K * 48, hipMemcpyDeviceToHost, ctx->stream));
    if (points) YGZ_HIPCHK(ctx, hipMemcpyAsync(points, w->points, (size_t)w->P * 24, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

}  // extern "C"

// =====================================================================================================================================
// SURVEY 8f-1, second half -- ceres::Solve as ba::LocalBA / OptimizeCurrent / OptimizeCurrentPointOnly / TwoViewBACeres configure it
// (src/Algorithm/BA.cpp:58-62,136-140,308-312,372-375: default options = trust-region Levenberg-Marquardt, Jacobi scaling, Schur
// elimination of the points) resident on the GPU: one workgroup per formulation-2 window runs IterationZero, every
// LevenbergMarquardtStrategy::ComputeStep (scaled blocks, clamped diagonal / radius, Schur complement, Cholesky, back-substitution),
// the step-validity and model-cost tests, the candidate evaluation and the radius policy without a host round trip.  It follows
// ygz_hip_ba_solve_ceres (ba_lm.hip: the same loop with the reduced system on the host) and oracle/ceres_ba.c::yo_ceres_solve
// [frozen spec of ceres-solver 1.13] decision by decision; the parameter update is the additive one of the ceres functors.
__device__ __forceinline__ double ce_point_cost(const BaDev &B, int il, int *behind)
{   // ba_point_chi2 + the PoseOnly functor's failure condition (p_z < 0 on an enabled edge)
    const int lane = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
    const double pt[3] = { B.points[3 * (size_t)il], B.points[3 * (size_t)il + 1], B.points[3 * (size_t)il + 2] };
    double sum = 0.0;
    for (int c = 0; c < rows; ++c) {
        const int row = row0 + c, ip = B.pose_c[(size_t)row * 64 + lane];
        if (ip < 0 || !B.enable_c[(size_t)row * 64 + lane]) continue;
        double p[3], r[2], rho0, rho1;
        ba_project(B, B.posed + BA_POSED * (size_t)ip, pt, BA_EC(B.obs_c, row, 2, 0, lane), BA_EC(B.obs_c, row, 2, 1, lane), p, r);
        if (p[2] < 0) *behind = 1;
        ba_robust(r[0] * r[0] + r[1] * r[1], B.huber_c[(size_t)row * 64 + lane], &rho0, &rho1);
        sum += rho0;
    }
    return sum;
}

__global__ __launch_bounds__(LM_THREADS) void k_ba_ceres(const BaDev *__restrict__ wins, ygz_ceres_options o, ygz_ceres_summary *__restrict__ sums)
{
    __shared__ double S[LM_CERES_MAXN * LM_CERES_MAXN];
    __shared__ double bs[LM_CERES_MAXN], xp[LM_CERES_MAXN];
    __shared__ double red27[LM_WAVES][28];
    __shared__ double red[LM_WAVES];
    __shared__ double dxp[16 * 6];
    __shared__ int s_fail;
    const BaDev B = wins[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = B.K, P = B.P, Kf = B.Kf, n = 6 * Kf;
    ygz_ceres_summary R;
    R.iterations = 0; R.successful_steps = 0; R.unsuccessful_steps = 0; R.termination = YGZ_CERES_NO_CONVERGENCE;
    R.initial_cost = 0.0; R.final_cost = 0.0; R.final_radius = 0.0;
    double x_cost = 0.0, radius = o.initial_trust_region_radius, decrease_factor = 2.0, x_norm = 0.0, gmax = 0.0;
    int invalid_run = 0, term = YGZ_CERES_NO_CONVERGENCE;

    // x_norm over the free parameters, max |gradient| (block-uniform)
#define CE_NORM_GRADIENT()                                                                                                           \
    {   double s2_ = 0.0, g_ = 0.0;                                                                                                  \
        for (int i = tid; i < 6 * Kf; i += LM_THREADS) { const int k_ = B.free_pose[i / 6], d_ = i % 6;                              \
            const double v_ = B.poses_w[6 * (size_t)k_ + d_]; s2_ += v_ * v_; g_ = fmax(g_, fabs(B.bp[6 * (size_t)k_ + d_])); }     \
        for (int i = tid; i < 3 * P; i += LM_THREADS) { const int l_ = i / 3, d_ = i % 3; if (B.point_fixed[l_]) continue;           \
            const double v_ = B.points_w[i]; s2_ += v_ * v_; g_ = fmax(g_, fabs(BA_PC(B.bl_c, l_, 3, d_))); }                        \
        x_norm = sqrt(lm_block_sum(s2_, red)); gmax = lm_block_max(g_, red); }

    do {
        // ---- IterationZero: residuals, Jacobians (as blocks), cost at the start
        const double chi0 = lm_linearize(B, red27, red);
        const int nb0 = *B.n_behind;                              // ba_point_edges counted the enabled edges with p_z < 0
        x_cost = 0.5 * chi0;
        if ((o.fail_behind_camera && nb0 > 0) || !isfinite(chi0)) { term = YGZ_CERES_FAILURE; break; }
        R.initial_cost = x_cost;
        for (int i = tid; i < 6 * K; i += LM_THREADS) B.sc_p[i] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(B.Hpp[36 * (size_t)(i / 6) + 7 * (i % 6)])) : 1.0;
        for (int i = tid; i < 3 * P; i += LM_THREADS) B.sc_l[i] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(BA_PC(B.Hll_c, i / 3, 9, 4 * (i % 3)))) : 1.0;
        __syncthreads();
        CE_NORM_GRADIENT()
        for (;;) {
            if (R.iterations >= o.max_num_iterations) { term = YGZ_CERES_NO_CONVERGENCE; break; }
            if (gmax <= o.gradient_tolerance) { term = YGZ_CERES_GRADIENT_TOLERANCE; break; }
            if (radius <= o.min_trust_region_radius) { term = YGZ_CERES_MIN_RADIUS; break; }
            ++R.iterations;
            __syncthreads();
            if (tid == 0) s_fail = 0;
            __syncthreads();
            // ---- 1. per point: D = sHll + diag(clamp(diag sHll) / radius), Dinv, Y_e = sHpl_e Dinv (sH = column-scaled blocks)
            for (int il = tid; il < P; il += LM_THREADS) {
                double *Di = B.Dinv + 9 * (size_t)il;
                const int ln = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
                double Dv[9];
                bool fixedl = B.point_fixed[il] != 0;
                if (!fixedl) {
                    const double sl[3] = { B.sc_l[3 * (size_t)il], B.sc_l[3 * (size_t)il + 1], B.sc_l[3 * (size_t)il + 2] };
                    double D[9];
                    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) D[3 * r + c] = BA_PC(B.Hll_c, il, 9, 3 * r + c) * sl[r] * sl[c];
                    for (int r = 0; r < 3; ++r) { const double dg = fmin(fmax(D[4 * r], o.min_lm_diagonal), o.max_lm_diagonal); D[4 * r] += dg / radius; }
                    if (!lm_inv3(D, Dv)) { s_fail = 1; fixedl = true; }
                }
                if (fixedl) for (int i = 0; i < 9; ++i) Dv[i] = 0.0;
                for (int i = 0; i < 9; ++i) Di[i] = Dv[i];
                for (int c = 0; c < rows; ++c) {
                    const int row = row0 + c, ip = B.pose_c[(size_t)row * 64 + ln];
                    if (ip < 0) continue;
                    double W[18];
                    for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 3; ++cc)
                        W[3 * r + cc] = BA_EC(B.Hpl_c, row, 18, 3 * r + cc, ln) * B.sc_p[6 * (size_t)ip + r] * B.sc_l[3 * (size_t)il + cc];
                    for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 3; ++cc)
                        BA_EC(B.Y_c, row, 18, 3 * r + cc, ln) = W[3 * r] * Dv[cc] + W[3 * r + 1] * Dv[3 + cc] + W[3 * r + 2] * Dv[6 + cc];
                }
            }
            // ---- 2. S = blockdiag(sHpp + dp) - sum_l Y_a sHpl_b^T, bs = sbp - sum_l Y_a sbl
            for (int i = tid; i < n * n; i += LM_THREADS) {
                const int r = i / n, c = i - r * n, a = r / 6, b = c / 6;
                double v = 0.0;
                if (a == b) {
                    const int k = B.free_pose[a], rr = r - 6 * a, cc = c - 6 * b;
                    v = B.Hpp[36 * (size_t)k + 6 * rr + cc] * B.sc_p[6 * (size_t)k + rr] * B.sc_p[6 * (size_t)k + cc];
                    if (r == c) { const double dg = fmin(fmax(v, o.min_lm_diagonal), o.max_lm_diagonal); v += dg / radius; }
                }
                S[i] = v;
            }
            for (int i = tid; i < n; i += LM_THREADS) { const int k = B.free_pose[i / 6]; bs[i] = B.bp[6 * (size_t)k + (i % 6)] * B.sc_p[6 * (size_t)k + (i % 6)]; }
            __syncthreads();
            lm_pair_sweep<true>(B, S, bs, n, B.sc_p, B.sc_l);
            __syncthreads();
            // ---- 3. Cholesky + substitutions (as in k_ba_lm_team)
            for (int j = 0; j < n; ++j) {
                if (tid == 0) { const double d = S[j * n + j]; if (!(d > 0) || !isfinite(d)) s_fail = 1; S[j * n + j] = sqrt(d > 0 ? d : 1.0); }
                __syncthreads();
                const double dj = S[j * n + j];
                for (int i = j + 1 + tid; i < n; i += LM_THREADS) S[i * n + j] = S[i * n + j] / dj;
                __syncthreads();
                const int m = n - j - 1;
                for (int t = tid; t < m * m; t += LM_THREADS) {
                    const int i = j + 1 + t / m, k = j + 1 + t % m;
                    if (k <= i) S[i * n + k] -= S[i * n + j] * S[k * n + j];
                }
                __syncthreads();
            }
            if (wv == 0) {
                for (int k = 0; k < n; ++k) {
                    if (lane == 0) bs[k] = bs[k] / S[k * n + k];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                    const double yk = bs[k];
                    for (int i = k + 1 + lane; i < n; i += 64) bs[i] -= S[i * n + k] * yk;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
                for (int k = n - 1; k >= 0; --k) {
                    if (lane == 0) bs[k] = bs[k] / S[k * n + k];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                    const double xk = bs[k];
                    for (int i = lane; i < k; i += 64) bs[i] -= S[k * n + i] * xk;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
                for (int i = lane; i < n; i += 64) xp[i] = bs[i];
            }
            for (int i = tid; i < 6 * K; i += LM_THREADS) dxp[i] = 0.0;
            __syncthreads();
            bool valid = s_fail == 0;
            // ---- 4. x_l = Dinv (sbl - sum sHpl^T xp); steps in the unscaled parameters; finiteness
            int bad = 0;
            if (valid) {
                for (int i = tid; i < n; i += LM_THREADS) { const int k = B.free_pose[i / 6]; const double v = xp[i] * B.sc_p[6 * (size_t)k + (i % 6)]; if (!isfinite(v)) bad = 1; dxp[6 * k + (i % 6)] = v; }
                for (int il = tid; il < P; il += LM_THREADS) {
                    double x3[3] = { 0.0, 0.0, 0.0 };
                    if (!B.point_fixed[il]) {
                        const double sl[3] = { B.sc_l[3 * (size_t)il], B.sc_l[3 * (size_t)il + 1], B.sc_l[3 * (size_t)il + 2] };
                        double r3[3] = { BA_PC(B.bl_c, il, 3, 0) * sl[0], BA_PC(B.bl_c, il, 3, 1) * sl[1], BA_PC(B.bl_c, il, 3, 2) * sl[2] };
                        const int ln = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
                        for (int c = 0; c < rows; ++c) {
                            const int row = row0 + c, ip = B.pose_c[(size_t)row * 64 + ln];
                            if (ip < 0) continue;
                            const int a = B.free_idx[ip];
                            if (a < 0) continue;
                            for (int cc = 0; cc < 3; ++cc) for (int r = 0; r < 6; ++r)
                                r3[cc] -= BA_EC(B.Hpl_c, row, 18, 3 * r + cc, ln) * B.sc_p[6 * (size_t)ip + r] * sl[cc] * xp[6 * a + r];
                        }
                        const double *Di = B.Dinv + 9 * (size_t)il;
                        for (int cc = 0; cc < 3; ++cc) { x3[cc] = (Di[3 * cc] * r3[0] + Di[3 * cc + 1] * r3[1] + Di[3 * cc + 2] * r3[2]) * sl[cc]; if (!isfinite(x3[cc])) bad = 1; }
                    }
                    for (int cc = 0; cc < 3; ++cc) B.xl[3 * (size_t)il + cc] = x3[cc];
                }
            }
            if (__syncthreads_or(bad)) valid = false;
            // ---- model_cost_change = d.b - d.H d / 2 from the (unscaled) blocks
            double model_cost_change = 0.0;
            if (valid) {
                double mcc = 0.0;
                for (int i = tid; i < n; i += LM_THREADS) {
                    const int k = B.free_pose[i / 6], r = i % 6;
                    double hd = 0.0;
                    for (int c = 0; c < 6; ++c) hd += B.Hpp[36 * (size_t)k + 6 * r + c] * dxp[6 * k + c];
                    mcc += dxp[6 * k + r] * (B.bp[6 * (size_t)k + r] - 0.5 * hd);
                }
                for (int il = tid; il < P; il += LM_THREADS) {
                    if (B.point_fixed[il]) continue;
                    const double d3[3] = { B.xl[3 * (size_t)il], B.xl[3 * (size_t)il + 1], B.xl[3 * (size_t)il + 2] };
                    for (int r = 0; r < 3; ++r) {
                        const double hd = BA_PC(B.Hll_c, il, 9, 3 * r) * d3[0] + BA_PC(B.Hll_c, il, 9, 3 * r + 1) * d3[1] + BA_PC(B.Hll_c, il, 9, 3 * r + 2) * d3[2];
                        mcc += d3[r] * (BA_PC(B.bl_c, il, 3, r) - 0.5 * hd);
                    }
                    const int ln = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
                    for (int c = 0; c < rows; ++c) {
                        const int row = row0 + c, ip = B.pose_c[(size_t)row * 64 + ln];
                        if (ip < 0) continue;
                        for (int r = 0; r < 6; ++r)
                            mcc -= dxp[6 * ip + r] * (BA_EC(B.Hpl_c, row, 18, 3 * r, ln) * d3[0] + BA_EC(B.Hpl_c, row, 18, 3 * r + 1, ln) * d3[1] + BA_EC(B.Hpl_c, row, 18, 3 * r + 2, ln) * d3[2]);
                    }
                }
                model_cost_change = lm_block_sum(mcc, red);
                if (!(model_cost_change > 0)) valid = false;
            }
            if (!valid) {                                                  // HandleInvalidStep
                if (++invalid_run >= o.max_num_consecutive_invalid_steps) { term = YGZ_CERES_FAILURE; break; }
                radius *= 0.5;
                ++R.unsuccessful_steps;
                continue;
            }
            invalid_run = 0;
            // ---- candidate = x + d (backup first), its cost
            double step2 = 0.0;
            for (int i = tid; i < 6 * K; i += LM_THREADS) { B.poses_bk[i] = B.poses_w[i]; B.poses_w[i] += dxp[i]; step2 += dxp[i] * dxp[i]; }
            for (int i = tid; i < 3 * P; i += LM_THREADS) { const double d = B.xl[i]; B.points_bk[i] = B.points_w[i]; B.points_w[i] += d; step2 += d * d; }
            step2 = lm_block_sum(step2, red);                              // (barriers inside: the candidate is visible)
            if (tid < K) ba_pose_prep_one(B, tid);
            __syncthreads();
            double cc = 0.0; int behind = 0;
            for (int il = tid; il < P; il += LM_THREADS) cc += ce_point_cost(B, il, &behind);
            cc = lm_block_sum(cc, red);
            const int any_behind = __syncthreads_or(behind);
            double cand_cost = 0.5 * cc;
            if ((o.fail_behind_camera && any_behind) || !isfinite(cc)) cand_cost = DBL_MAX;      // a failed evaluation = a step of very high cost
            bool accept = false, leave = false;
            if (sqrt(step2) <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { term = YGZ_CERES_PARAMETER_TOLERANCE; leave = true; }
            else {
                const double cost_change = x_cost - cand_cost;
                if (fabs(cost_change) <= o.function_tolerance * x_cost) { term = YGZ_CERES_FUNCTION_TOLERANCE; leave = true; }
                else {
                    const double relative_decrease = cost_change / model_cost_change;
                    if (relative_decrease > o.min_relative_decrease) {     // HandleSuccessfulStep
                        accept = true;
                        double t = 2.0 * relative_decrease - 1.0;
                        t = 1.0 - t * t * t;
                        radius = fmin(radius / fmax(1.0 / 3.0, t), o.max_trust_region_radius);
                        decrease_factor = 2.0;
                        ++R.successful_steps;
                    } else {                                               // StepRejected
                        radius = radius / decrease_factor;
                        decrease_factor *= 2.0;
                        ++R.unsuccessful_steps;
                    }
                }
            }
            if (!accept) {                                                 // the iterate stays (also on the two tolerance exits)
                __syncthreads();
                for (int i = tid; i < 6 * K; i += LM_THREADS) B.poses_w[i] = B.poses_bk[i];
                for (int i = tid; i < 3 * P; i += LM_THREADS) B.points_w[i] = B.points_bk[i];
                __syncthreads();
                if (leave) break;
                continue;
            }
            const double chi = lm_linearize(B, red27, red);                // blocks of the new iterate
            x_cost = 0.5 * chi;
            CE_NORM_GRADIENT()
        }
    } while (0);
#undef CE_NORM_GRADIENT
    if (tid == 0) { R.termination = term; R.final_cost = x_cost; R.final_radius = radius; sums[blockIdx.x] = R; }
}

extern "C" int ygz_hip_ba_solve_ceres_resident(ygz_hip_ctx *ctx, int window_begin, int n_windows, const ygz_ceres_options *opt_in,
                                               ygz_ceres_summary *summaries)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || window_begin < 0 || n_windows < 1 || window_begin + n_windows > (int)ctx->ba.size()) return YGZ_E_INVALID;
    for (int i = window_begin; i < window_begin + n_windows; ++i) {
        if (!ctx->ba[i]) return YGZ_E_INVALID;
        if (ctx->ba[i]->formulation != 2) return YGZ_E_INVALID;      // the ceres functors
        if (ctx->ba[i]->Kf > LM_CERES_MAXKF || ctx->ba[i]->K > 16) return YGZ_E_CAPACITY;
        if (ctx->ba[i]->has_dup) return YGZ_E_INVALID;
    }
    ygz_ceres_options opt;
    if (opt_in) opt = *opt_in; else ygz_hip_ceres_default_options(&opt);
    int rc = YGZ_OK;
    const BaDev *table = ygz_ba_table(ctx, &rc);
    if (!table) return rc;
    void *d_sum = nullptr;
    if ((rc = ygz_scratch(ctx, SCR_BA_0, (size_t)n_windows * sizeof(ygz_ceres_summary), &d_sum)) != YGZ_OK) return rc;
    YgzAuxScope aux(ctx, 1);
    YGZ_LAUNCH(ctx, KID_BA_LM, k_ba_ceres, dim3(n_windows), dim3(LM_THREADS), table + window_begin, opt, (ygz_ceres_summary *)d_sum);
    YGZ_HIPCHK(ctx, hipGetLastError());
    if (summaries) {
        YGZ_HIPCHK(ctx, hipMemcpyAsync(summaries, d_sum, (size_t)n_windows * sizeof(ygz_ceres_summary), hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return YGZ_OK;
}
