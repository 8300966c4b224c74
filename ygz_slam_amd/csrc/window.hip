// The keyframe side of the batched offline run (BASELINE configs[4]) kept in HBM: what LocalMapping::LocalBA does on the host before
// it calls ba::LocalBAG2O (src/Module/LocalMapping.cpp:149-208 collects the window's keyframes, their map points and the features
// that observe them; src/Algorithm/BA.cpp:397-470 turns them into g2o vertices and edges) happens here on the device, so that a BA
// round needs no keypoint table on the host and no graph upload:
//   keyframe store   one fixed-size row per keyframe (pixels, depths, levels, descriptors, count) -- the rows of a rank's own
//                    keyframes are copied from the tracking context's slots device-to-device (k_kf_put); the rows of other ranks
//                    arrive through the caller's collective on the same memory (fixed-shape rows: one all-gather);
//   relative poses   T_rel of every frame (pose of frame f in the frame of f - 1, what pose-only BA left per pair);
//   window build     per window of K keyframes, anchor = keyframe 0 (held fixed like keyframe 0 at BA.cpp:404):
//                    k_win_select   the anchor's features with depth (first max_points, keypoint order) -> a compacted descriptor set;
//                    matcher        that set against every other keyframe of the window: BFMatcher(crossCheck) + the good-match rule
//                                   (test/test_orb_match.cpp:86-104) -- k_hamming_f4 / k_match_postfilter on the store's rows;
//                    k_win_edges    points seen by >= 2 keyframes, their observations as the chunked row layout of ba_dev.h, the
//                                   (point, pose) -> edge table, and the window's poses chained from the relative poses:
//                                   T(anchor) = identity, T(f) = T_rel(f) * T(f - 1), vertex estimate = log as [omega; upsilon];
//                                   map points = Pixel2Camera(pixel, depth) in the anchor's camera (Camera.h:56-62).
//                    A window is therefore a function of its own frames only: whichever rank builds it, and whenever, the graph and
//                    the LM result are bit-identical, and no window waits for the global trajectory.
//   state rows       [poses 6K | points 3P | K P E iterations trials chi2_0 chi2 lambda | edges tested, outliers, chi2, chi2 of inliers] per window, packed on the device for the
//                    exchange (owner -> everybody) and the host.
#include "ba_dev.h"
#include <string.h>
#include <vector>

size_t ygz_ba_zero_bytes(const ygz_hip_ctx::BaWindow *w);                          // ba.hip
int ygz_ba_reserve_window(ygz_hip_ctx *ctx, int window, int K, int P, double huber);

struct ygz_hip_ctx::KfStore {
    int n_kf = 0, n_frames = 0, max_windows = 0;
    size_t row_bytes = 0, off_px = 0, off_depth = 0, off_level = 0, off_desc = 0, off_count = 0;
    bool with_images = false;                              // rows also hold the keyframe's pyramid (levels 0 .. pyramid_levels - 1), for the direct-projection observations
    size_t off_img[YGZ_MAX_LEVELS] = {0};
    uint8_t *rows = nullptr; bool own_rows = false;       // [n_kf + max_windows] rows: keyframes, then the compacted anchor sets
    double *trel = nullptr;                                // [n_frames][7]
    int32_t *counts = nullptr;                             // [n_kf + max_windows] rows in use per row (contiguous: the matcher's set sizes)
};

struct KfView {
    uint8_t *rows; size_t row_bytes, off_px, off_depth, off_level, off_desc, off_count; int cells, n_kf;
    int32_t *counts; double *trel;
};
__device__ __forceinline__ double *kfv_px(const KfView &V, int r) { return reinterpret_cast<double *>(V.rows + (size_t)r * V.row_bytes + V.off_px); }
__device__ __forceinline__ double *kfv_depth(const KfView &V, int r) { return reinterpret_cast<double *>(V.rows + (size_t)r * V.row_bytes + V.off_depth); }
__device__ __forceinline__ int32_t *kfv_level(const KfView &V, int r) { return reinterpret_cast<int32_t *>(V.rows + (size_t)r * V.row_bytes + V.off_level); }
__device__ __forceinline__ uint4 *kfv_desc(const KfView &V, int r) { return reinterpret_cast<uint4 *>(V.rows + (size_t)r * V.row_bytes + V.off_desc); }
__device__ __forceinline__ int32_t *kfv_count(const KfView &V, int r) { return reinterpret_cast<int32_t *>(V.rows + (size_t)r * V.row_bytes + V.off_count); }

// off[0..4]: pixels, depth, level, descriptors, count; off_img[L] (with images): level L of the keyframe's pyramid, each 64-byte aligned
static void kf_layout(const ygz_hip_ctx *ctx, bool with_images, size_t *off, size_t *off_img, size_t *row_bytes)
{
    const size_t C = (size_t)ctx->cells;
    auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
    off[0] = 0; off[1] = al(16 * C); off[2] = al(off[1] + 8 * C); off[3] = al(off[2] + 4 * C); off[4] = al(off[3] + 32 * C);
    size_t end = off[4] + 64;
    for (int L = 0; L < YGZ_MAX_LEVELS; ++L) off_img[L] = 0;
    if (with_images)
        for (int L = 0; L < ctx->prm.pyramid_levels; ++L) { off_img[L] = al(end); end = off_img[L] + (size_t)ctx->lw[L] * ctx->lh[L]; }
    if (with_images) end += 64;                          // the row-wise window loads may touch a few bytes past a level
    *row_bytes = (end + 255) & ~(size_t)255;
}

static KfView kf_view(const ygz_hip_ctx *ctx)
{
    const auto *S = ctx->kfs;
    KfView V;
    V.rows = S->rows; V.row_bytes = S->row_bytes; V.off_px = S->off_px; V.off_depth = S->off_depth; V.off_level = S->off_level;
    V.off_desc = S->off_desc; V.off_count = S->off_count; V.cells = ctx->cells; V.n_kf = S->n_kf; V.counts = S->counts; V.trel = S->trel;
    return V;
}

// ---- keyframe rows from a tracking context's slots -------------------------------------------------------------------
#define KF_PUT_MAX 64
struct KfPutArgs {
    KfView V; int n;
    const int32_t *n_kp; const double *kp_px, *kp_depth; const int32_t *kp_level; const uint32_t *kp_desc;
    int32_t slot[KF_PUT_MAX], kf[KF_PUT_MAX];
};
__global__ __launch_bounds__(256) void k_kf_put(KfPutArgs A)
{
    const int b = blockIdx.y, slot = A.slot[b], kf = A.kf[b], i = blockIdx.x * 256 + threadIdx.x;
    const int n = A.n_kp[slot];
    if (i == 0) { *kfv_count(A.V, kf) = n; A.V.counts[kf] = n; }
    if (i >= n) return;
    const size_t s = (size_t)slot * A.V.cells + i;
    double *px = kfv_px(A.V, kf);
    px[2 * i] = A.kp_px[2 * s]; px[2 * i + 1] = A.kp_px[2 * s + 1];
    kfv_depth(A.V, kf)[i] = A.kp_depth[s];
    kfv_level(A.V, kf)[i] = A.kp_level[s];
    const uint4 *sd = reinterpret_cast<const uint4 *>(A.kp_desc + 8 * s);
    uint4 *dd = kfv_desc(A.V, kf) + 2 * (size_t)i;
    dd[0] = sd[0]; dd[1] = sd[1];
}
// the keyframe's pyramid levels into its row (16 bytes per lane; blockIdx.y = keyframe of the call, blockIdx.z = level)
struct KfImgArgs {
    uint8_t *rows; size_t row_bytes, off_img[YGZ_MAX_LEVELS], bytes[YGZ_MAX_LEVELS];
    const uint8_t *lvl[YGZ_MAX_LEVELS];
    int32_t slot[KF_PUT_MAX], kf[KF_PUT_MAX];
};
__global__ __launch_bounds__(256) void k_kf_put_img(KfImgArgs A)
{
    const int b = blockIdx.y, L = blockIdx.z;
    const size_t n = A.bytes[L], i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i >= n) return;
    const uint8_t *src = A.lvl[L] + (size_t)A.slot[b] * n + i;
    uint8_t *dst = A.rows + (size_t)A.kf[b] * A.row_bytes + A.off_img[L] + i;
    if (i + 16 <= n && (n & 15) == 0) *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
    else for (size_t k = 0; k < 16 && i + k < n; ++k) dst[k] = src[k];
}
__global__ __launch_bounds__(256) void k_trel_put(double *__restrict__ trel, const double *__restrict__ po_T, int n)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < 7 * n) trel[t] = po_T[t];
}
__global__ __launch_bounds__(256) void k_kf_counts(KfView V)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < V.n_kf) V.counts[r] = *kfv_count(V, r);
}

// ---- window build -----------------------------------------------------------------------------------------------------
#define WIN_THREADS 1024
#define WIN_WAVES   (WIN_THREADS / 64)
#define WIN_MAXQ    256                      // chunks of 64 points per window: max_points <= 16384

struct WinArgs {
    KfView V;
    const int32_t *kf_index, *kf_frame, *n_kfs, *pair_of;     // [n_win][Kcap] x 2, [n_win], [n_win][Kcap] (pair of (window, keyframe j), -1: none)
    int Kcap, Pcap, set_base;                                 // anchor set of window w = store row set_base + w
    BaDev *wins;                                              // table entries of the windows being built
    int32_t *sel_idx, *sc_new, *sc_cnt, *sc_e0;               // [n_win][Pcap] scratch
    const int32_t *m_idx; const uint8_t *m_good; size_t m_stride;
    // direct != 0: the observations come from FindDirectProjection (k_win_project) instead of the Hamming matches: o_ok / o_px [pairs][m_stride]
    int direct; const uint8_t *o_ok; const double *o_px;
    double *Tj;                                               // [n_win][Kcap][7] pose of keyframe j relative to the anchor (k_win_poses)
};

__device__ __forceinline__ int win_wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
    return v;
}
// exclusive prefix of v over the block and the block total; red: [WIN_WAVES] ints
__device__ __forceinline__ int win_block_excl_scan(int v, int *red, int *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int inc = win_wave_incl_scan(v, lane);
    __syncthreads();                                       // red is free
    if (lane == 63) red[wv] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < WIN_WAVES; ++k) { const int r = red[k]; if (k < wv) off += r; tot += r; }
    *total = tot;
    return off + inc - v;
}

// the anchor's features with a depth, first Pcap in keypoint order -> store row set_base + w (pixels, depths, levels, descriptors)
__global__ __launch_bounds__(WIN_THREADS) void k_win_select(WinArgs A)
{
    __shared__ int red[WIN_WAVES];
    const int w = blockIdx.x, tid = threadIdx.x;
    const int a = A.kf_index[(size_t)w * A.Kcap], S = A.set_base + w;
    const int cnt = A.n_kfs[w] >= 1 ? A.V.counts[a] : 0;
    const double *apx = kfv_px(A.V, a), *adp = kfv_depth(A.V, a);
    const int32_t *alv = kfv_level(A.V, a);
    const uint4 *ads = kfv_desc(A.V, a);
    double *spx = kfv_px(A.V, S), *sdp = kfv_depth(A.V, S);
    int32_t *slv = kfv_level(A.V, S);
    uint4 *sds = kfv_desc(A.V, S);
    int running = 0;
    for (int base = 0; base < cnt && running < A.Pcap; base += WIN_THREADS) {
        const int i = base + tid;
        const bool f = i < cnt && adp[i] > 0;
        int tot;
        const int pos = running + win_block_excl_scan(f ? 1 : 0, red, &tot);
        if (f && pos < A.Pcap) {
            spx[2 * pos] = apx[2 * i]; spx[2 * pos + 1] = apx[2 * i + 1];
            sdp[pos] = adp[i]; slv[pos] = alv[i];
            sds[2 * (size_t)pos] = ads[2 * (size_t)i]; sds[2 * (size_t)pos + 1] = ads[2 * (size_t)i + 1];
            A.sel_idx[(size_t)w * A.Pcap + pos] = i;
        }
        running += tot;
    }
    if (tid == 0) { const int n = running < A.Pcap ? running : A.Pcap; A.V.counts[S] = n; *kfv_count(A.V, S) = n; }
}

// keyframe j's pose relative to the anchor, chained from the frames' relative poses: T(anchor) = identity, T(f) = T_rel(f) * T(f - 1)
__device__ __forceinline__ void win_chain(const WinArgs &A, int w, int j, Se3 &T)
{
    const int32_t *kff = A.kf_frame + (size_t)w * A.Kcap;
    T.q[0] = T.q[1] = T.q[2] = 0; T.q[3] = 1; T.t[0] = T.t[1] = T.t[2] = 0;
    for (int f = kff[0] + 1; f <= kff[j]; ++f) {
        Se3 Rl, C;
        const double *r = A.V.trel + 7 * (size_t)f;
        Rl.q[0] = r[0]; Rl.q[1] = r[1]; Rl.q[2] = r[2]; Rl.q[3] = r[3]; Rl.t[0] = r[4]; Rl.t[1] = r[5]; Rl.t[2] = r[6];
        se3_mul_d(&Rl, &T, &C);
        T = C;
    }
}
__global__ __launch_bounds__(64) void k_win_poses(WinArgs A)
{
    const int w = blockIdx.x, j = threadIdx.x;
    if (j >= A.Kcap) return;
    Se3 T; T.q[0] = T.q[1] = T.q[2] = 0; T.q[3] = 1; T.t[0] = T.t[1] = T.t[2] = 0;
    if (j >= 1 && j < A.n_kfs[w]) win_chain(A, w, j, T);
    double *o = A.Tj + 7 * ((size_t)w * A.Kcap + j);
    o[0] = T.q[0]; o[1] = T.q[1]; o[2] = T.q[2]; o[3] = T.q[3]; o[4] = T.t[0]; o[5] = T.t[1]; o[6] = T.t[2];
}

__global__ __launch_bounds__(WIN_THREADS) void k_win_edges(WinArgs A)
{
    __shared__ int red[WIN_WAVES];
    __shared__ int chunk_max[WIN_MAXQ], s_slot[WIN_MAXQ + 1];
    const int w = blockIdx.x, tid = threadIdx.x;
    const int nk = A.n_kfs[w], S = A.set_base + w, n_sel = nk >= 2 ? A.V.counts[S] : 0;
    BaDev B = A.wins[w];
    const int Kf = nk > 1 ? nk - 1 : 0;
    const double *spx = kfv_px(A.V, S), *sdp = kfv_depth(A.V, S);
    int32_t *sc_new = A.sc_new + (size_t)w * A.Pcap, *sc_cnt = A.sc_cnt + (size_t)w * A.Pcap, *sc_e0 = A.sc_e0 + (size_t)w * A.Pcap;
    const int32_t *kfi = A.kf_index + (size_t)w * A.Kcap, *pof = A.pair_of + (size_t)w * A.Kcap;
    for (int q = tid; q < WIN_MAXQ; q += WIN_THREADS) chunk_max[q] = 0;
    // ---- A. observations per selected point, points with >= 2 keep their order, edges sorted by (point, keyframe)
    int P = 0, E = 0;
    for (int base = 0; base < n_sel; base += WIN_THREADS) {
        const int s = base + tid;
        int cnt = 0;
        if (s < n_sel) {
            cnt = 1;                                                           // the anchor's own observation
            for (int j = 1; j < nk; ++j) {
                const int p = pof[j];
                if (p < 0) continue;
                if (A.direct ? A.o_ok[(size_t)p * A.m_stride + s] != 0
                             : (A.m_idx[(size_t)p * A.m_stride + s] >= 0 && A.m_good[(size_t)p * A.m_stride + s])) ++cnt;
            }
        }
        const bool keep = cnt >= 2;                                            // seen only by the constant anchor: constrains nothing
        int tp, te;
        const int l = P + win_block_excl_scan(keep ? 1 : 0, red, &tp);
        const int e0 = E + win_block_excl_scan(keep ? cnt : 0, red, &te);
        if (s < n_sel) { sc_new[s] = keep ? l : -1; sc_cnt[s] = cnt; sc_e0[s] = e0; }
        if (keep) atomicMax(&chunk_max[l >> 6], cnt);
        P += tp; E += te;
    }
    __syncthreads();
    // ---- B. rows: the c-th edge of every point of a 64-point chunk; a chunk has as many rows as its longest point
    const int Q = (P + 63) >> 6;
    if (tid == 0) { int o = 0; for (int q = 0; q < Q; ++q) { s_slot[q] = o; o += chunk_max[q]; } s_slot[Q] = o; }
    __syncthreads();
    const int R = s_slot[Q];
    int32_t *slot_off = const_cast<int32_t *>(B.slot_off), *pose_c = const_cast<int32_t *>(B.pose_c), *edge_rl = const_cast<int32_t *>(B.edge_rl);
    int16_t *ppc = const_cast<int16_t *>(B.ppc), *dupn = const_cast<int16_t *>(B.dupn);
    uint8_t *enable_c = const_cast<uint8_t *>(B.enable_c), *fixed = const_cast<uint8_t *>(B.fixed), *point_fixed = const_cast<uint8_t *>(B.point_fixed);
    double *obs_c = const_cast<double *>(B.obs_c), *huber_c = const_cast<double *>(B.huber_c), *points = const_cast<double *>(B.points);
    double *poses = const_cast<double *>(B.poses);
    for (int q = tid; q <= Q; q += WIN_THREADS) slot_off[q] = s_slot[q];
    for (size_t t = tid; t < (size_t)R * 64; t += WIN_THREADS) { pose_c[t] = -1; dupn[t] = -1; enable_c[t] = 0; }
    for (size_t t = tid; t < (size_t)P * (Kf > 0 ? Kf : 1); t += WIN_THREADS) ppc[t] = -1;
    __syncthreads();
    // ---- C. fill: edge 0 = the anchor (pose 0), then the good matches in keyframe order
    for (int s = tid; s < n_sel; s += WIN_THREADS) {
        const int l = sc_new[s];
        if (l < 0) continue;
        const int lane = l & 63, row0 = s_slot[l >> 6], e0 = sc_e0[s];
        const double x = spx[2 * s], y = spx[2 * s + 1], z = sdp[s];
        points[3 * (size_t)l] = (x - B.cx) * z / B.fx; points[3 * (size_t)l + 1] = (y - B.cy) * z / B.fy; points[3 * (size_t)l + 2] = z;   // Pixel2Camera
        point_fixed[l] = 0;
        int c = 0;
        {
            const size_t r = (size_t)row0;
            pose_c[r * 64 + lane] = 0; enable_c[r * 64 + lane] = 1; huber_c[r * 64 + lane] = B.huber;
            BA_EC(obs_c, r, 2, 0, lane) = x; BA_EC(obs_c, r, 2, 1, lane) = y;
            edge_rl[e0] = (int32_t)(r * 64 + lane);
            c = 1;
        }
        for (int j = 1; j < nk; ++j) {
            const int p = pof[j];
            if (p < 0) continue;
            double ox, oy;
            if (A.direct) {
                if (!A.o_ok[(size_t)p * A.m_stride + s]) continue;
                ox = A.o_px[2 * ((size_t)p * A.m_stride + s)]; oy = A.o_px[2 * ((size_t)p * A.m_stride + s) + 1];
            } else {
                const int t = A.m_idx[(size_t)p * A.m_stride + s];
                if (t < 0 || !A.m_good[(size_t)p * A.m_stride + s]) continue;
                const double *kpx = kfv_px(A.V, kfi[j]);
                ox = kpx[2 * (size_t)t]; oy = kpx[2 * (size_t)t + 1];
            }
            const size_t r = (size_t)row0 + c;
            pose_c[r * 64 + lane] = j; enable_c[r * 64 + lane] = 1; huber_c[r * 64 + lane] = B.huber;
            BA_EC(obs_c, r, 2, 0, lane) = ox; BA_EC(obs_c, r, 2, 1, lane) = oy;
            edge_rl[e0 + c] = (int32_t)(r * 64 + lane);
            ppc[(size_t)l * Kf + (j - 1)] = (int16_t)c;
            ++c;
        }
    }
    // ---- D. vertices: keyframe j's pose relative to the anchor, chained from the frames' relative poses
    if (tid < A.Kcap) {
        const int j = tid;
        int32_t *free_idx = const_cast<int32_t *>(B.free_idx), *free_pose = const_cast<int32_t *>(B.free_pose);
        fixed[j] = (uint8_t)(j == 0);
        free_idx[j] = j < nk ? j - 1 : -1;
        if (j >= 1) free_pose[j - 1] = j;
        double est[6] = { 0, 0, 0, 0, 0, 0 };
        if (j >= 1 && j < nk) {
            Se3 T;
            win_chain(A, w, j, T);
            double lg[6];
            se3_log_d(&T, lg);                                              // [upsilon; omega]
            est[0] = lg[3]; est[1] = lg[4]; est[2] = lg[5]; est[3] = lg[0]; est[4] = lg[1]; est[5] = lg[2];   // VertexSE3Sophus: [omega; upsilon]
        }
        for (int k = 0; k < 6; ++k) poses[6 * (size_t)j + k] = est[k];
    }
    if (tid == 0) {
        BaDev *o = A.wins + w;
        o->K = nk > 0 ? nk : 1; o->Kf = Kf; o->P = P; o->E = E; o->R = R; o->Q = Q;
    }
}

// [poses 6 Kcap | points 3 Pcap | K P E iterations trials chi2_initial chi2_final lambda | edges tested, outliers, chi2 of the tested edges, chi2 of the inliers] per window (unused entries 0)
__global__ __launch_bounds__(256) void k_ba_pack(const BaDev *__restrict__ wins, double *__restrict__ out, size_t row_doubles, int Kcap, int Pcap)
{
    const BaDev B = wins[blockIdx.x];
    double *o = out + (size_t)blockIdx.x * row_doubles;
    const int nk = 6 * B.K, np = 3 * B.P;
    for (int t = threadIdx.x; t < 6 * Kcap; t += 256) o[t] = t < nk ? B.poses[t] : 0.0;
    for (int t = threadIdx.x; t < 3 * Pcap; t += 256) o[6 * Kcap + t] = t < np ? B.points[t] : 0.0;
    if (threadIdx.x == 0) {
        const ygz_ba_stats st = *reinterpret_cast<const ygz_ba_stats *>(B.lm_out);
        double *d = o + 6 * (size_t)Kcap + 3 * (size_t)Pcap;
        d[0] = B.K; d[1] = B.P; d[2] = B.E; d[3] = st.iterations; d[4] = st.lm_trials; d[5] = st.chi2_initial; d[6] = st.chi2_final; d[7] = st.lambda_final;
        for (int i = 0; i < 4; ++i) d[8 + i] = B.lm_out[4 + i];           // edges tested, outliers, chi2 of all tested edges, chi2 of the inliers (ygz_hip_ba_mark_outliers; -1: not computed)
    }
}

void ygz_kf_store_free(ygz_hip_ctx *ctx)
{
    auto *S = ctx->kfs;
    if (!S) return;
    if (S->own_rows && S->rows) (void)hipFree(S->rows);
    if (S->trel) (void)hipFree(S->trel);
    if (S->counts) (void)hipFree(S->counts);
    delete S;
    ctx->kfs = nullptr;
}

extern "C" {

// host helper of the offline run: the trajectory T[0] = identity, T[i] = T_rel[i] * T[i - 1] (Sophus SE3 product, the formulas the device
// uses for the windows' chains): a thousand products are microseconds here and a millisecond in an interpreter loop
int ygz_hip_se3_chain(const double *T_rel, int n, double *T_out)
{
    if (!T_rel || !T_out || n < 0) return YGZ_E_INVALID;
    Se3 acc; acc.q[0] = acc.q[1] = acc.q[2] = 0; acc.q[3] = 1; acc.t[0] = acc.t[1] = acc.t[2] = 0;
    for (int i = 0; i < n; ++i) {
        if (i > 0) {
            Se3 r, c;
            for (int k = 0; k < 4; ++k) r.q[k] = T_rel[7 * (size_t)i + k];
            for (int k = 0; k < 3; ++k) r.t[k] = T_rel[7 * (size_t)i + 4 + k];
            se3_mul_d(&r, &acc, &c);
            acc = c;
        }
        for (int k = 0; k < 4; ++k) T_out[7 * (size_t)i + k] = acc.q[k];
        for (int k = 0; k < 3; ++k) T_out[7 * (size_t)i + 4 + k] = acc.t[k];
    }
    return YGZ_OK;
}

size_t ygz_hip_kf_row_bytes(const ygz_hip_ctx *ctx, int with_images)
{
    if (!ctx) return 0;
    size_t off[5], oi[YGZ_MAX_LEVELS], rb;
    kf_layout(ctx, with_images != 0, off, oi, &rb);
    return rb;
}

int ygz_hip_kf_store_create(ygz_hip_ctx *ctx, int n_keyframes, int n_frames, int max_windows, void *rows_mem, size_t rows_mem_bytes, int with_images)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || n_keyframes < 1 || n_frames < 1 || max_windows < 1) return YGZ_E_INVALID;
    if (ctx->kfs) { YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); ygz_kf_store_free(ctx); }
    auto *S = new ygz_hip_ctx::KfStore();
    size_t off[5];
    S->with_images = with_images != 0;
    kf_layout(ctx, S->with_images, off, S->off_img, &S->row_bytes);
    S->off_px = off[0]; S->off_depth = off[1]; S->off_level = off[2]; S->off_desc = off[3]; S->off_count = off[4];
    S->n_kf = n_keyframes; S->n_frames = n_frames; S->max_windows = max_windows;
    const size_t n_rows = (size_t)n_keyframes + max_windows, need = n_rows * S->row_bytes;
    ctx->kfs = S;
    if (rows_mem) {
        if (rows_mem_bytes < need || ((uintptr_t)rows_mem & 15)) { ygz_kf_store_free(ctx); return YGZ_E_INVALID; }
        S->rows = (uint8_t *)rows_mem; S->own_rows = false;
    } else {
        hipError_t e = hipMalloc((void **)&S->rows, need + 64);
        if (e != hipSuccess) { ctx->last_hip_error = (int)e; ygz_kf_store_free(ctx); return YGZ_E_HIP; }
        S->own_rows = true;
    }
    hipError_t e = hipMalloc((void **)&S->trel, (size_t)n_frames * 56);
    if (e == hipSuccess) e = hipMalloc((void **)&S->counts, n_rows * 4);
    if (e != hipSuccess) { ctx->last_hip_error = (int)e; ygz_kf_store_free(ctx); return YGZ_E_HIP; }
    YGZ_HIPCHK(ctx, hipMemsetAsync(S->rows, 0, need, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemsetAsync(S->trel, 0, (size_t)n_frames * 56, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemsetAsync(S->counts, 0, n_rows * 4, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_kf_store_info(ygz_hip_ctx *ctx, void **rows, size_t *row_bytes, void **trel, int *n_keyframes, int *n_frames)
{
    if (!ctx || !ctx->kfs) return YGZ_E_STATE;
    if (rows) *rows = ctx->kfs->rows;
    if (row_bytes) *row_bytes = ctx->kfs->row_bytes;
    if (trel) *trel = ctx->kfs->trel;
    if (n_keyframes) *n_keyframes = ctx->kfs->n_kf;
    if (n_frames) *n_frames = ctx->kfs->n_frames;
    return YGZ_OK;
}

// rows kf_index[i] <- the keypoints (pixels, levels, descriptors, depths) of slot src_slot[i] of `src`, device to device, enqueued on
// src's stream (behind the extraction that fills the slots); order the store's context behind it with ygz_hip_stream_wait
int ygz_hip_kf_store_put(ygz_hip_ctx *store, ygz_hip_ctx *src, int n, const int32_t *src_slot, const int32_t *kf_index)
{
    if (!store || !src || !store->kfs || n < 0 || (n > 0 && (!src_slot || !kf_index))) return YGZ_E_INVALID;
    if (store->device != src->device || store->cells != src->cells || !src->trk_alloc) return YGZ_E_STATE;
    YgzDeviceGuard dg_(src);
    { int rj = ygz_join(src); if (rj != YGZ_OK) return rj; }
    for (int i = 0; i < n; ++i)
        if (src_slot[i] < 0 || src_slot[i] >= src->prm.max_frames || kf_index[i] < 0 || kf_index[i] >= store->kfs->n_kf) return YGZ_E_INVALID;
    if (store->kfs->with_images) {
        // every precondition of the image copy is checked BEFORE the first launch: the call either fails cleanly or enqueues all n rows
        if (src->prm.pyramid_levels != store->prm.pyramid_levels || src->lw[0] != store->lw[0] || src->lh[0] != store->lh[0]) return YGZ_E_STATE;
        for (int i = 0; i < n; ++i) if (!src->pyr_valid[src_slot[i]]) return YGZ_E_STATE;
    }
    for (int b = 0; b < n; b += KF_PUT_MAX) {
        KfPutArgs A;
        A.V = kf_view(store); A.n = n - b < KF_PUT_MAX ? n - b : KF_PUT_MAX;
        A.n_kp = src->n_kp; A.kp_px = src->kp_px; A.kp_depth = src->kp_depth; A.kp_level = src->kp_level; A.kp_desc = src->kp_desc;
        for (int i = 0; i < KF_PUT_MAX; ++i) { A.slot[i] = i < A.n ? src_slot[b + i] : 0; A.kf[i] = i < A.n ? kf_index[b + i] : 0; }
        YGZ_LAUNCH(src, KID_WINDOW, k_kf_put, dim3(ygz_div_up(src->cells, 256), A.n), dim3(256), A);
        if (store->kfs->with_images) {
            KfImgArgs I;
            size_t mx = 0;
            for (int L = 0; L < YGZ_MAX_LEVELS; ++L) {
                const bool in = L < store->prm.pyramid_levels;
                I.lvl[L] = in ? src->lvl[L] : nullptr; I.off_img[L] = store->kfs->off_img[L]; I.bytes[L] = in ? (size_t)src->lw[L] * src->lh[L] : 0;
                if (I.bytes[L] > mx) mx = I.bytes[L];
            }
            I.rows = store->kfs->rows; I.row_bytes = store->kfs->row_bytes;
            for (int i = 0; i < KF_PUT_MAX; ++i) { I.slot[i] = A.slot[i]; I.kf[i] = A.kf[i]; }
            YGZ_LAUNCH(src, KID_WINDOW, k_kf_put_img, dim3((unsigned)((mx / 16 + 255) / 256 + 1), A.n, store->prm.pyramid_levels), dim3(256), I);
        }
    }
    YGZ_HIPCHK(src, hipGetLastError());
    return YGZ_OK;
}

// T_rel[first_frame + i] <- the pose-only pose of pair first_pair + i of `src` (ygz_hip_track_pose_only), i < n_pairs; on src's stream
int ygz_hip_kf_store_put_trel(ygz_hip_ctx *store, ygz_hip_ctx *src, int first_pair, int n_pairs, int first_frame)
{
    if (!store || !src || !store->kfs || first_pair < 0 || n_pairs < 1 || first_frame < 0) return YGZ_E_INVALID;
    if (store->device != src->device || !src->trk_alloc) return YGZ_E_STATE;
    if (first_pair + n_pairs > src->n_pairs || first_frame + n_pairs > store->kfs->n_frames) return YGZ_E_INVALID;
    YgzDeviceGuard dg_(src);
    { int rj = ygz_join(src); if (rj != YGZ_OK) return rj; }
    YGZ_LAUNCH(src, KID_WINDOW, k_trel_put, dim3(ygz_div_up(7 * n_pairs, 256)), dim3(256), store->kfs->trel + 7 * (size_t)first_frame,
               src->po_T + 7 * (size_t)first_pair, n_pairs);
    YGZ_HIPCHK(src, hipGetLastError());
    return YGZ_OK;
}

// the same rows from the host (frames tracked by other ranks, after the trajectory all-gather); asynchronous on the store's stream
int ygz_hip_kf_store_set_trel(ygz_hip_ctx *ctx, int first_frame, int n, const double *T_rel)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !ctx->kfs || !T_rel || first_frame < 0 || n < 1 || first_frame + n > ctx->kfs->n_frames) return YGZ_E_INVALID;
    void *st = ygz_stage(ctx, (size_t)n * 56);
    if (!st) return YGZ_E_HIP;
    memcpy(st, T_rel, (size_t)n * 56);
    YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->kfs->trel + 7 * (size_t)first_frame, st, (size_t)n * 56, hipMemcpyHostToDevice, ctx->stream));
    return YGZ_OK;
}

// after a collective wrote rows of other ranks into the store's memory: the per-row counts into the contiguous array the matcher reads
int ygz_hip_kf_store_refresh(ygz_hip_ctx *ctx)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !ctx->kfs) return YGZ_E_STATE;
    YGZ_LAUNCH(ctx, KID_WINDOW, k_kf_counts, dim3(ygz_div_up(ctx->kfs->n_kf, 256)), dim3(256), kf_view(ctx));
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

int ygz_hip_ba_reserve_windows(ygz_hip_ctx *ctx, int window_begin, int n_windows, int K, int max_points, double huber_delta)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || window_begin < 0 || n_windows < 1 || max_points < 1 || max_points > 64 * WIN_MAXQ) return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    for (int i = 0; i < n_windows; ++i) {
        const int rc = ygz_ba_reserve_window(ctx, window_begin + i, K, max_points, huber_delta);
        if (rc != YGZ_OK) return rc;
    }
    return YGZ_OK;
}

// The graphs of windows window_begin .. +n_windows-1 (reserved with ygz_hip_ba_reserve_windows) from the keyframe store, entirely on
// the device and asynchronous: window i consists of the n_kfs[i] keyframes in store rows kf_index[i][0 .. K) which are the frames
// kf_frame[i][.] of the sequence (ascending); see the head of this file for what is built.
int ygz_hip_ba_build_windows(ygz_hip_ctx *ctx, int window_begin, int n_windows, const int32_t *kf_index, const int32_t *kf_frame,
                             const int32_t *n_kfs, int obs_mode)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !ctx->kfs || window_begin < 0 || n_windows < 1 || window_begin + n_windows > (int)ctx->ba.size() || !kf_index || !kf_frame || !n_kfs)
        return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    auto *S = ctx->kfs;
    if (n_windows > S->max_windows) return YGZ_E_CAPACITY;
    if (obs_mode != 0 && obs_mode != 1) return YGZ_E_INVALID;
    if (obs_mode == 1 && !S->with_images) return YGZ_E_STATE;                // direct projection reads the keyframes' pyramids from the rows
    const auto *w0 = ctx->ba[window_begin];
    if (!w0 || !w0->device_built) return YGZ_E_STATE;
    const int K = w0->cap_K, P = w0->cap_P;
    if (K > 64) return YGZ_E_CAPACITY;                                       // k_win_poses: one lane per keyframe of a window
    int n_pairs = 0;
    for (int i = 0; i < n_windows; ++i) {
        const auto *w = ctx->ba[window_begin + i];
        if (!w || !w->device_built || w->cap_K != K || w->cap_P != P) return YGZ_E_STATE;
        if (n_kfs[i] < 2 || n_kfs[i] > K) return YGZ_E_INVALID;
        for (int j = 0; j < n_kfs[i]; ++j) {
            const int r = kf_index[(size_t)i * K + j], f = kf_frame[(size_t)i * K + j];
            if (r < 0 || r >= S->n_kf || f < 0 || f >= S->n_frames || (j > 0 && f <= kf_frame[(size_t)i * K + j - 1])) return YGZ_E_INVALID;
        }
        n_pairs += n_kfs[i] - 1;
    }
    if (obs_mode == 0 && n_pairs > ctx->prm.max_frames) return YGZ_E_CAPACITY;   // the matcher's per-pair result rows
    int rc = YGZ_OK;
    const BaDev *table = ygz_ba_table(ctx, &rc);                            // (uploads the entries of freshly reserved windows)
    if (!table) return rc;
    if ((rc = ygz_pf_ensure(ctx)) != YGZ_OK) return rc;
    // host tables -> page-locked stage -> device scratch
    // int tables: kf_index, kf_frame, pair_of [n_win][K] | n_kfs [n_win] | per pair: (set row, keyframe row) for the matcher, (window, j) for the projection
    const size_t nK = (size_t)n_windows * K, n_int = 3 * nK + n_windows + 4 * (size_t)n_pairs;
    int32_t *h = (int32_t *)ygz_stage(ctx, n_int * 4);
    if (!h) return YGZ_E_HIP;
    int32_t *h_kfi = h, *h_kff = h + nK, *h_pof = h + 2 * nK, *h_nk = h + 3 * nK, *h_pq = h_nk + n_windows, *h_pt = h_pq + n_pairs, *h_pw = h_pt + n_pairs,
            *h_pj = h_pw + n_pairs;
    int p = 0;
    for (int i = 0; i < n_windows; ++i) {
        h_nk[i] = n_kfs[i];
        for (int j = 0; j < K; ++j) {
            const bool in = j < n_kfs[i];
            h_kfi[(size_t)i * K + j] = in ? kf_index[(size_t)i * K + j] : 0;
            h_kff[(size_t)i * K + j] = in ? kf_frame[(size_t)i * K + j] : 0;
            h_pof[(size_t)i * K + j] = -1;
            if (in && j >= 1) { h_pof[(size_t)i * K + j] = p; h_pq[p] = S->n_kf + i; h_pt[p] = kf_index[(size_t)i * K + j]; h_pw[p] = i; h_pj[p] = j; ++p; }
        }
    }
    // scratch: the int tables | 4 x [n_win][P] ints | Tj [n_win][K][7] doubles | (direct) obs_px [pairs][P][2] doubles, obs_ok [pairs][P]
    const size_t n_scr_int = ((n_int + 4 * (size_t)n_windows * P + 1) & ~(size_t)1);
    const size_t tj_bytes = nK * 56, opx_bytes = obs_mode == 1 ? (size_t)n_pairs * P * 16 : 0, ook_bytes = obs_mode == 1 ? (size_t)n_pairs * P : 0;
    int32_t *d = nullptr;
    if ((rc = ygz_scratch(ctx, SCR_WIN, n_scr_int * 4 + tj_bytes + opx_bytes + ook_bytes + 64, (void **)&d)) != YGZ_OK) return rc;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d, h, n_int * 4, hipMemcpyHostToDevice, ctx->stream));
    WinArgs A;
    A.V = kf_view(ctx);
    A.kf_index = d; A.kf_frame = d + nK; A.pair_of = d + 2 * nK; A.n_kfs = d + 3 * nK;
    const int32_t *d_pq = d + 3 * nK + n_windows, *d_pt = d_pq + n_pairs;
    A.Kcap = K; A.Pcap = P; A.set_base = S->n_kf;
    A.wins = const_cast<BaDev *>(table) + window_begin;
    A.sel_idx = d + n_int; A.sc_new = A.sel_idx + (size_t)n_windows * P; A.sc_cnt = A.sc_new + (size_t)n_windows * P; A.sc_e0 = A.sc_cnt + (size_t)n_windows * P;
    A.m_idx = ctx->m_idx; A.m_good = ctx->m_good; A.m_stride = (size_t)ctx->cells;
    A.Tj = reinterpret_cast<double *>(d + n_scr_int);
    double *d_opx = reinterpret_cast<double *>(reinterpret_cast<uint8_t *>(A.Tj) + tj_bytes);
    uint8_t *d_ook = reinterpret_cast<uint8_t *>(d_opx) + opx_bytes;
    A.direct = obs_mode; A.o_ok = d_ook; A.o_px = d_opx;
    if (obs_mode == 1) A.m_stride = (size_t)P;
    for (int i = 0; i < n_windows; ++i) {
        auto *w = ctx->ba[window_begin + i];
        YGZ_HIPCHK(ctx, hipMemsetAsync(w->Hpp, 0, ygz_ba_zero_bytes(w), ctx->stream));
    }
    YGZ_LAUNCH(ctx, KID_WINDOW, k_win_select, dim3(n_windows), dim3(WIN_THREADS), A);
    YGZ_LAUNCH(ctx, KID_WINDOW, k_win_poses, dim3(n_windows), dim3(64), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    if (obs_mode == 0) {
        rc = ygz_run_match(ctx, reinterpret_cast<const uint32_t *>(S->rows + S->off_desc), S->row_bytes / 4, S->counts, d_pq, d_pt, n_pairs, ctx->cells, 1, false);
        if (rc != YGZ_OK) return rc;
        if ((rc = ygz_launch_match_postfilter(ctx, S->counts, d_pq, n_pairs, 20.0, 50.0, 3.0)) != YGZ_OK) return rc;     // test_orb_match.cpp:97-104
        ctx->n_pairs = 0; ctx->pf_valid = false;                            // the per-pair buffers of the resident pair table were reused
    } else {
        // FindCandidates + ProjectMapPoints (LocalMapping.cpp:47-120): every selected anchor feature into every other keyframe of its window
        YGZ_HIPCHK(ctx, hipMemsetAsync(d_ook, 0, ook_bytes, ctx->stream));
        YgzWinProject W;
        W.rows = S->rows; W.row_bytes = S->row_bytes; W.off_px = S->off_px; W.off_depth = S->off_depth; W.off_level = S->off_level;
        W.n_levels = ctx->prm.pyramid_levels;
        for (int L = 0; L < YGZ_MAX_LEVELS; ++L) { W.off_img[L] = S->off_img[L]; W.w[L] = ctx->lw[L]; W.h[L] = ctx->lh[L]; }
        W.pair_w = d_pt + n_pairs; W.pair_j = W.pair_w + n_pairs; W.kf_index = A.kf_index; W.counts = S->counts;
        W.n_pairs = n_pairs; W.Kcap = K; W.Pcap = P; W.set_base = S->n_kf; W.Tj = A.Tj; W.obs_px = d_opx; W.obs_ok = d_ook; W.stride = (size_t)P;
        if ((rc = ygz_launch_win_project(ctx, W)) != YGZ_OK) return rc;
    }
    YGZ_LAUNCH(ctx, KID_WINDOW, k_win_edges, dim3(n_windows), dim3(WIN_THREADS), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

// state rows of the windows (layout at the head of this file), row_doubles >= 6 K + 3 max_points + 8 apart, into device memory (e.g.
// the buffer of the caller's all-gather) or, through a staging buffer, into host memory
int ygz_hip_ba_pack_states(ygz_hip_ctx *ctx, int window_begin, int n_windows, double *dst, size_t row_doubles, int dst_on_device, int wait)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !dst || window_begin < 0 || n_windows < 1 || window_begin + n_windows > (int)ctx->ba.size()) return YGZ_E_INVALID;
    const auto *w0 = ctx->ba[window_begin];
    if (!w0) return YGZ_E_INVALID;
    const int K = w0->device_built ? w0->cap_K : w0->K, P = w0->device_built ? w0->cap_P : w0->P;
    for (int i = 0; i < n_windows; ++i) {
        const auto *w = ctx->ba[window_begin + i];
        if (!w || (w->device_built ? w->cap_K : w->K) != K || (w->device_built ? w->cap_P : w->P) != P) return YGZ_E_INVALID;
    }
    if (row_doubles < (size_t)6 * K + (size_t)3 * P + 12) return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    int rc = YGZ_OK;
    const BaDev *table = ygz_ba_table(ctx, &rc);
    if (!table) return rc;
    double *out = dst;
    if (!dst_on_device && (rc = ygz_scratch(ctx, SCR_GEN_0 + 3, (size_t)n_windows * row_doubles * 8, (void **)&out)) != YGZ_OK) return rc;
    if (row_doubles > (size_t)6 * K + (size_t)3 * P + 12) YGZ_HIPCHK(ctx, hipMemsetAsync(out, 0, (size_t)n_windows * row_doubles * 8, ctx->stream));
    YGZ_LAUNCH(ctx, KID_WINDOW, k_ba_pack, dim3(n_windows), dim3(256), table + window_begin, out, row_doubles, K, P);
    YGZ_HIPCHK(ctx, hipGetLastError());
    if (!dst_on_device) YGZ_HIPCHK(ctx, hipMemcpyAsync(dst, out, (size_t)n_windows * row_doubles * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (wait) YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

}  // extern "C"
