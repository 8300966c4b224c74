// M3 / M6 -- the "best distance -> clamp -> keep what is below a multiple of it" filters that follow the matcher:
//   M3  test/test_orb_match.cpp:97-104   min over the DMatches, clamp to [20, 50], keep distance < 3 * min_dis
//   M6  Matcher::CheckFrameDescriptors   src/Algorithm/Matcher.cpp:45-84: Hamming distance of given (index1, index2) pairs,
//       best clamped to [init_low, init_high], keep distance < initMatchRatio * best
// One workgroup per set: lanes stride over the rows, the minimum is a wavefront DPP min + 4 LDS partials (integers: order is
// irrelevant), the keep flags are written in place and counted with a ballot.  The resident form runs all frame pairs of the
// pair table in one launch, straight on the matcher's output in HBM.
#include "ygz_internal.h"

#define PF_THREADS 256

__device__ __forceinline__ int pf_wave_min_i(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0xB1, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x4E, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x141, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x140, 0xF, 0xF, false));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

__device__ __forceinline__ int pf_block_min(int v, int *red)
{
    const int m = pf_wave_min_i(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const int r = min(min(red[0], red[1]), min(red[2], red[3]));
    __syncthreads();
    return r;
}

__device__ __forceinline__ int pf_block_count(bool flag, int *red)
{
    const int c = __popcll(__ballot(flag));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    const int r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

// M3 on the matcher's per-pair output: idx/dist [pairs][stride]; good [pairs][stride]; stats [pairs][2] = (n_good, min_dis as
// the clamped integer-valued double stored in 2 x i32 is avoided: min_dis is returned as double in a separate array)
struct PfArgs {
    const int32_t *set_count, *pair_q;            // rows of pair p = set_count[pair_q[p]]  (pair_q == nullptr: set_count[p])
    const int32_t *idx, *dist; size_t stride;
    uint8_t *good; int32_t *n_good; double *min_dis;
    double lo, hi, factor;
};

__global__ __launch_bounds__(PF_THREADS) void k_match_postfilter(PfArgs A)
{
    __shared__ int red[4];
    const int p = blockIdx.x;
    const int n = A.set_count[A.pair_q ? A.pair_q[p] : p];
    const size_t o = (size_t)p * A.stride;
    int mn = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < n; i += PF_THREADS)
        if (A.idx[o + i] >= 0) mn = min(mn, A.dist[o + i]);
    mn = pf_block_min(mn, red);
    // min_dis = min_dis<20?20:min_dis; min_dis = min_dis>50?50:min_dis;   (double, test_orb_match.cpp:99-101)
    double md = (double)mn;                        // no match at all: 2147483647 -> clamped to hi, nothing is kept anyway
    md = md < A.lo ? A.lo : md;
    md = md > A.hi ? A.hi : md;
    const double th = A.factor * md;
    int cnt = 0;
    for (int base = 0; base < n; base += PF_THREADS) {
        const int i = base + threadIdx.x;
        bool k = false;
        if (i < n) { k = A.idx[o + i] >= 0 && (double)A.dist[o + i] < th; A.good[o + i] = (uint8_t)k; }
        cnt += pf_block_count(k, red);
    }
    if (threadIdx.x == 0) { A.n_good[p] = cnt; A.min_dis[p] = md; }
}

// M6: lane = index pair; descriptors [rows][8] u32 of the two sets, idx1/idx2 [n] (nullptr: row i of each set)
struct CkArgs {
    const uint32_t *d1, *d2; const int32_t *i1, *i2; int n, n1, n2;
    int low, high; float ratio;
    int32_t *dist; uint8_t *keep; int32_t *out;      // out[0] = cnt_good, out[1] = clamped best
};

__global__ __launch_bounds__(PF_THREADS) void k_check_pairs(CkArgs A)
{
    __shared__ int red[4];
    int mn = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < A.n; i += PF_THREADS) {
        const int a = A.i1 ? A.i1[i] : i, b = A.i2 ? A.i2[i] : i;
        int d = 0x7FFFFFFF;
        if (a >= 0 && a < A.n1 && b >= 0 && b < A.n2) {
            const uint4 *pa = reinterpret_cast<const uint4 *>(A.d1 + 8 * (size_t)a), *pb = reinterpret_cast<const uint4 *>(A.d2 + 8 * (size_t)b);
            const uint4 x0 = pa[0], x1 = pa[1], y0 = pb[0], y1 = pb[1];
            d = __popc(x0.x ^ y0.x) + __popc(x0.y ^ y0.y) + __popc(x0.z ^ y0.z) + __popc(x0.w ^ y0.w) +
                __popc(x1.x ^ y1.x) + __popc(x1.y ^ y1.y) + __popc(x1.z ^ y1.z) + __popc(x1.w ^ y1.w);   // Matcher::DescriptorDistance
        }
        A.dist[i] = d;
        mn = min(mn, d);
    }
    mn = pf_block_min(mn, red);
    int best = mn;
    best = best > A.low ? best : A.low;             // Matcher.cpp:65-66
    best = best < A.high ? best : A.high;
    const float th = A.ratio * (float)best;         // float * int -> float (:72)
    int cnt = 0;
    for (int base = 0; base < A.n; base += PF_THREADS) {
        const int i = base + threadIdx.x;
        bool k = false;
        if (i < A.n) { k = (float)A.dist[i] < th; A.keep[i] = (uint8_t)k; }
        cnt += pf_block_count(k, red);
    }
    if (threadIdx.x == 0) { A.out[0] = cnt; A.out[1] = best; }
}

int ygz_pf_ensure(ygz_hip_ctx *ctx)
{
    if (ctx->m_good) return YGZ_OK;
    const size_t F = (size_t)ctx->prm.max_frames, Cn = (size_t)ctx->cells;
    YGZ_HIPCHK(ctx, hipMalloc((void **)&ctx->m_good, F * Cn));
    YGZ_HIPCHK(ctx, hipMalloc((void **)&ctx->m_good_n, F * 4));
    YGZ_HIPCHK(ctx, hipMalloc((void **)&ctx->m_min_dis, F * 8));
    return YGZ_OK;
}

// the filter over the resident match rows of n_pairs pairs whose query-set sizes are set_count[pair_q[p]] (device arrays)
int ygz_launch_match_postfilter(ygz_hip_ctx *ctx, const int32_t *set_count, const int32_t *pair_q, int n_pairs, double lo, double hi, double factor)
{
    int rc = ygz_pf_ensure(ctx);
    if (rc != YGZ_OK) return rc;
    PfArgs A;
    A.set_count = set_count; A.pair_q = pair_q; A.idx = ctx->m_idx; A.dist = ctx->m_dist; A.stride = (size_t)ctx->cells;
    A.good = ctx->m_good; A.n_good = ctx->m_good_n; A.min_dis = ctx->m_min_dis;
    A.lo = lo; A.hi = hi; A.factor = factor;
    YGZ_LAUNCH(ctx, KID_MATCH_POSTFILTER, k_match_postfilter, dim3(n_pairs), dim3(PF_THREADS), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

extern "C" {

int ygz_hip_match_postfilter(ygz_hip_ctx *ctx, double min_floor, double min_ceil, double factor)
{
    if (!ctx || !(min_floor <= min_ceil) || !(factor > 0)) return YGZ_E_INVALID;
    if (ctx->n_pairs < 1) return YGZ_E_STATE;
    YgzDeviceGuard dg(ctx);
    int rc = ygz_pf_ensure(ctx);
    if (rc != YGZ_OK) return rc;
    YgzAuxScope aux(ctx, YGZ_AUX_MATCH);            // rides behind the matcher when that runs on its side stream
    PfArgs A;
    A.set_count = ctx->n_kp; A.pair_q = ctx->pair_q; A.idx = ctx->m_idx; A.dist = ctx->m_dist; A.stride = (size_t)ctx->cells;
    A.good = ctx->m_good; A.n_good = ctx->m_good_n; A.min_dis = ctx->m_min_dis;
    A.lo = min_floor; A.hi = min_ceil; A.factor = factor;
    YGZ_LAUNCH(ctx, KID_MATCH_POSTFILTER, k_match_postfilter, dim3(ctx->n_pairs), dim3(PF_THREADS), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    ctx->pf_valid = true;
    return YGZ_OK;
}

int ygz_hip_get_good_matches(ygz_hip_ctx *ctx, int pair, uint8_t *good, int capacity, int *nq_out, int *n_good, double *min_dis)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || pair < 0 || pair >= ctx->n_pairs || !nq_out) return YGZ_E_INVALID;
    if (!ctx->pf_valid || !ctx->m_good) return YGZ_E_STATE;
    YgzDeviceGuard dg(ctx);
    int32_t qslot = 0, nq = 0, ng = 0; double md = 0;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&qslot, ctx->pair_q + pair, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&nq, ctx->n_kp + qslot, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&ng, ctx->m_good_n + pair, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&md, ctx->m_min_dis + pair, 8, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *nq_out = nq;
    if (n_good) *n_good = ng;
    if (min_dis) *min_dis = md;
    if (nq > capacity) return YGZ_E_CAPACITY;
    if (nq > 0 && good) {
        YGZ_HIPCHK(ctx, hipMemcpyAsync(good, ctx->m_good + (size_t)pair * ctx->cells, (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return YGZ_OK;
}

// stand-alone form on host match arrays (what ygz_hip_hamming_match returned)
int ygz_hip_match_postfilter_host(ygz_hip_ctx *ctx, const int32_t *train_idx, const int32_t *dist, int nq, double min_floor,
                                  double min_ceil, double factor, uint8_t *good, int *n_good, double *min_dis)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || nq < 0 || !(min_floor <= min_ceil) || !(factor > 0) || (nq > 0 && (!train_idx || !dist || !good))) return YGZ_E_INVALID;
    if (nq == 0) {                                  // the reference dereferences end() here (UB): defined as "nothing kept"
        if (n_good) *n_good = 0;
        if (min_dis) *min_dis = min_ceil;
        return YGZ_OK;
    }
    YgzDeviceGuard dg(ctx);
    uint8_t *buf = nullptr;
    const size_t N = (size_t)nq, off_d = 64 + N * 4, off_g = off_d + N * 4, off_s = (off_g + N + 15) & ~(size_t)15;
    int rc = ygz_scratch(ctx, SCR_GEN_0, off_s + 64, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(buf, &nq, 4, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(buf + 64, train_idx, N * 4, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(buf + off_d, dist, N * 4, hipMemcpyHostToDevice, ctx->stream));
    PfArgs A;
    A.set_count = (const int32_t *)buf; A.pair_q = nullptr; A.idx = (const int32_t *)(buf + 64); A.dist = (const int32_t *)(buf + off_d);
    A.stride = N; A.good = buf + off_g; A.min_dis = (double *)(buf + off_s); A.n_good = (int32_t *)(buf + off_s + 8);
    A.lo = min_floor; A.hi = min_ceil; A.factor = factor;
    YGZ_LAUNCH(ctx, KID_MATCH_POSTFILTER, k_match_postfilter, dim3(1), dim3(PF_THREADS), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    struct { double md; int32_t ng; int32_t pad; } st;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(good, buf + off_g, N, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&st, buf + off_s, 16, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (n_good) *n_good = st.ng;
    if (min_dis) *min_dis = st.md;
    return YGZ_OK;
}

static int run_check(ygz_hip_ctx *ctx, const uint32_t *d1, int n1, const uint32_t *d2, int n2, const int32_t *h_i1, const int32_t *h_i2,
                     int n, int low, int high, float ratio, int32_t *dist, uint8_t *keep, int *n_good, int *best)
{
    uint8_t *buf = nullptr;
    const size_t N = (size_t)n, off_i2 = N * 4, off_d = 2 * N * 4, off_k = 3 * N * 4, off_o = (off_k + N + 15) & ~(size_t)15;
    int rc = ygz_scratch(ctx, SCR_GEN_0 + 1, off_o + 64, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    if (h_i1) YGZ_HIPCHK(ctx, hipMemcpyAsync(buf, h_i1, N * 4, hipMemcpyHostToDevice, ctx->stream));
    if (h_i2) YGZ_HIPCHK(ctx, hipMemcpyAsync(buf + off_i2, h_i2, N * 4, hipMemcpyHostToDevice, ctx->stream));
    CkArgs A;
    A.d1 = d1; A.d2 = d2; A.i1 = h_i1 ? (const int32_t *)buf : nullptr; A.i2 = h_i2 ? (const int32_t *)(buf + off_i2) : nullptr;
    A.n = n; A.n1 = n1; A.n2 = n2; A.low = low; A.high = high; A.ratio = ratio;
    A.dist = (int32_t *)(buf + off_d); A.keep = buf + off_k; A.out = (int32_t *)(buf + off_o);
    YGZ_LAUNCH(ctx, KID_MATCH_POSTFILTER, k_check_pairs, dim3(1), dim3(PF_THREADS), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    int32_t out[2] = { 0, 0 };
    if (dist) YGZ_HIPCHK(ctx, hipMemcpyAsync(dist, A.dist, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (keep) YGZ_HIPCHK(ctx, hipMemcpyAsync(keep, A.keep, N, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(out, A.out, 8, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (n_good) *n_good = out[0];
    if (best) *best = out[1];
    return YGZ_OK;
}

int ygz_hip_check_frame_descriptors(ygz_hip_ctx *ctx, int slot1, int slot2, const int32_t *idx1, const int32_t *idx2, int n,
                                    int init_low, int init_high, float ratio, int32_t *dist, uint8_t *keep, int *n_good, int *best_dist)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || slot1 < 0 || slot1 >= ctx->prm.max_frames || slot2 < 0 || slot2 >= ctx->prm.max_frames || n < 0 ||
        (n > 0 && (!idx1 || !idx2))) return YGZ_E_INVALID;
    if (n == 0) { if (n_good) *n_good = 0; if (best_dist) *best_dist = init_low; return YGZ_OK; }     // reference: UB (min_element of an empty vector)
    YgzDeviceGuard dg(ctx);
    int32_t cnt[2];
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&cnt[0], ctx->n_kp + slot1, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&cnt[1], ctx->n_kp + slot2, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; ++i) if (idx1[i] < 0 || idx1[i] >= cnt[0] || idx2[i] < 0 || idx2[i] >= cnt[1]) return YGZ_E_INVALID;
    return run_check(ctx, ctx->kp_desc + (size_t)slot1 * ctx->cells * 8, cnt[0], ctx->kp_desc + (size_t)slot2 * ctx->cells * 8, cnt[1],
                     idx1, idx2, n, init_low, init_high, ratio, dist, keep, n_good, best_dist);
}

int ygz_hip_check_descriptor_pairs(ygz_hip_ctx *ctx, const uint8_t *desc1, const uint8_t *desc2, int n, int init_low, int init_high,
                                   float ratio, int32_t *dist, uint8_t *keep, int *n_good, int *best_dist)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || n < 0 || (n > 0 && (!desc1 || !desc2))) return YGZ_E_INVALID;
    if (n == 0) { if (n_good) *n_good = 0; if (best_dist) *best_dist = init_low; return YGZ_OK; }
    YgzDeviceGuard dg(ctx);
    uint8_t *d = nullptr;
    const size_t N = (size_t)n;
    int rc = ygz_scratch(ctx, SCR_MATCH_Q, N * 64 + 64, (void **)&d);
    if (rc != YGZ_OK) return rc;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d, desc1, N * 32, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d + N * 32, desc2, N * 32, hipMemcpyHostToDevice, ctx->stream));
    return run_check(ctx, (const uint32_t *)d, n, (const uint32_t *)(d + N * 32), n, nullptr, nullptr, n, init_low, init_high, ratio,
                     dist, keep, n_good, best_dist);
}

}  // extern "C"
