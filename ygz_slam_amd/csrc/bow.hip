// M4 / M5 -- BoW-guided matching.
//   Frame::ComputeBoW (src/Basic/Frame.cpp:190-201) -> Vocabulary::transform (thirdparty/DBoW3/src/Vocabulary.cpp:706-835):
//     k_bow_transform, lane = feature: descend the vocabulary tree, at every level the FIRST child with the strictly smallest
//     Hamming distance wins; word id, weight and the node `levelsup` levels above the leaf (the FeatureVector key).
//   Matcher::SearchByBoW (src/Algorithm/Matcher.cpp:196-292) and Matcher::SearchForTriangulation (:86-193) with
//     CheckDistEpipolarLine (:338-354): k_bow_match, lane = feature of frame 1, frame 2 streamed through LDS in tiles of 256
//     (descriptor + node id [+ pixel]); only features that share the vocabulary node are compared -- the segmented Hamming
//     search the reference performs by walking two std::map<NodeId, vector<idx>> in lockstep.  Integer work: bit-exact.
// The vocabulary is DBoW3's binary file (Vocabulary::loadFromBinaryFile layout); the reference does not ship vocab/ORBvoc.bin.
#include "ygz_internal.h"
#include <vector>
#include <string.h>

struct ygz_hip_ctx::Vocab {
    int k = 0, L = 0, n_nodes = 0, n_words = 0;
    void *blob = nullptr;
    int32_t *child_off, *child, *word_id; uint32_t *desc; double *weight;
    int32_t *kp_word = nullptr, *kp_node = nullptr; double *kp_weight = nullptr;       // [F][cells] per resident keypoint
};

struct VocDev { int L; const int32_t *child_off, *child, *word_id; const uint32_t *desc; const double *weight; };

__device__ __forceinline__ int bow_dist(const uint32_t a[8], const uint32_t *__restrict__ b)
{
    const uint4 b0 = *reinterpret_cast<const uint4 *>(b), b1 = *reinterpret_cast<const uint4 *>(b + 4);
    return __popc(a[0] ^ b0.x) + __popc(a[1] ^ b0.y) + __popc(a[2] ^ b0.z) + __popc(a[3] ^ b0.w) +
           __popc(a[4] ^ b1.x) + __popc(a[5] ^ b1.y) + __popc(a[6] ^ b1.z) + __popc(a[7] ^ b1.w);
}

// desc_base + slot/set stride; n per set from n_arr (device) or n_fixed
__global__ __launch_bounds__(256) void k_bow_transform(VocDev V, const uint32_t *__restrict__ desc, const int32_t *__restrict__ n_arr, int n_fixed,
                                                       size_t set_stride, int levelsup, int32_t *__restrict__ word, double *__restrict__ weight,
                                                       int32_t *__restrict__ node)
{
    const int set = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_arr ? n_arr[set] : n_fixed;
    if (i >= n) return;
    const size_t g = (size_t)set * set_stride + i;
    uint32_t d[8];
    { const uint4 a = *reinterpret_cast<const uint4 *>(desc + 8 * g), b = *reinterpret_cast<const uint4 *>(desc + 8 * g + 4);
      d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w; }
    const int nid_level = V.L - levelsup;
    int nid = 0, final_id = 0, level = 0;
    for (;;) {
        ++level;
        const int c0 = V.child_off[final_id], c1 = V.child_off[final_id + 1];
        if (c0 == c1) break;
        final_id = V.child[c0];
        int best = bow_dist(d, V.desc + 8 * (size_t)final_id);
        for (int c = c0 + 1; c < c1; ++c) {
            const int id = V.child[c];
            const int dd = bow_dist(d, V.desc + 8 * (size_t)id);
            if (dd < best) { best = dd; final_id = id; }
        }
        if (level == nid_level) nid = final_id;
        if (V.child_off[final_id] == V.child_off[final_id + 1]) break;
    }
    const int w = V.word_id[final_id];
    const double wt = V.weight[final_id];
    word[g] = w; weight[g] = wt;
    node[g] = (wt > 0 && w >= 0) ? nid : -1;                 // stopped words stay out of the FeatureVector
}

struct BowPair { const uint32_t *d1, *d2; const int32_t *n1, *n2; const double *p1, *p2; int c1, c2; double E[9]; };

// MODE 0: SearchByBoW (best + second best, th_low, knn ratio).  MODE 1: SearchForTriangulation (last candidate with
// dist <= th_low, dist <= best and the epipolar constraint).
template <int MODE>
__global__ __launch_bounds__(256) void k_bow_match(const BowPair *__restrict__ pairs, int th_low, float knn_ratio, double eps_dsqr,
                                                   double fx, double fy, double cx, double cy, int32_t *__restrict__ match12, size_t match_stride,
                                                   int32_t *__restrict__ counts)
{
    __shared__ uint32_t s_d[256][9];          // descriptor + node (pitch 9: conflict-free broadcast reads)
    __shared__ double s_p[256][2];
    const BowPair P = pairs[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= P.c1) return;
    const bool live = i < P.c1;
    uint32_t d[8];
    int nd = -1;
    double pt1x = 0, pt1y = 0;
    if (live) {
        const uint4 a = *reinterpret_cast<const uint4 *>(P.d1 + 8 * (size_t)i), b = *reinterpret_cast<const uint4 *>(P.d1 + 8 * (size_t)i + 4);
        d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
        nd = P.n1[i];
        if (MODE == 1) { pt1x = (P.p1[2 * (size_t)i] - cx) * 1.0 / fx; pt1y = (P.p1[2 * (size_t)i + 1] - cy) * 1.0 / fy; }     // Pixel2Camera, Camera.h:56-62
    }
    float ea = 0.f, eb = 0.f, ec = 0.f;
    if (MODE == 1) {      // CheckDistEpipolarLine: a, b, c are floats of double expressions (Matcher.cpp:341-343)
        ea = (float)(pt1x * P.E[0] + pt1y * P.E[3] + P.E[6]);
        eb = (float)(pt1x * P.E[1] + pt1y * P.E[4] + P.E[7]);
        ec = (float)(pt1x * P.E[2] + pt1y * P.E[5] + P.E[8]);
    }
    int best1 = 256, best2 = 256, bidx = -1;
    for (int j0 = 0; j0 < P.c2; j0 += 256) {
        const int j = j0 + threadIdx.x;
        __syncthreads();
        if (j < P.c2) {
            const uint4 a = *reinterpret_cast<const uint4 *>(P.d2 + 8 * (size_t)j), b = *reinterpret_cast<const uint4 *>(P.d2 + 8 * (size_t)j + 4);
            uint32_t *r = s_d[threadIdx.x];
            r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w; r[8] = (uint32_t)P.n2[j];
            if (MODE == 1) { s_p[threadIdx.x][0] = P.p2[2 * (size_t)j]; s_p[threadIdx.x][1] = P.p2[2 * (size_t)j + 1]; }
        }
        __syncthreads();
        if (!live || nd < 0) continue;
        const int cnt = min(256, P.c2 - j0);
        for (int t = 0; t < cnt; ++t) {
            const uint32_t *r = s_d[t];
            if ((int)r[8] != nd) continue;
            const int dist = __popc(d[0] ^ r[0]) + __popc(d[1] ^ r[1]) + __popc(d[2] ^ r[2]) + __popc(d[3] ^ r[3]) +
                             __popc(d[4] ^ r[4]) + __popc(d[5] ^ r[5]) + __popc(d[6] ^ r[6]) + __popc(d[7] ^ r[7]);
            if (MODE == 0) {
                if (dist < best1) { best2 = best1; best1 = dist; bidx = j0 + t; }
                else if (dist < best2) best2 = dist;
            } else {
                if (dist > th_low || dist > best1) continue;
                const double p2x = (s_p[t][0] - cx) * 1.0 / fx, p2y = (s_p[t][1] - cy) * 1.0 / fy;
                const float num = (float)((double)ea * p2x + (double)eb * p2y + (double)ec);
                const float den = __fadd_rn(__fmul_rn(ea, ea), __fmul_rn(eb, eb));
                if ((double)den < 1e-6) continue;
                const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
                if ((double)fabsf(dsqr) < eps_dsqr) { bidx = j0 + t; best1 = dist; }
            }
        }
    }
    if (!live) return;
    int m = -1;
    if (MODE == 0) { if (bidx >= 0 && best1 < th_low && (float)best1 < __fmul_rn(knn_ratio, (float)best2)) m = bidx; }
    else m = bidx;
    match12[(size_t)blockIdx.y * match_stride + i] = m;
    if (m >= 0) atomicAdd(&counts[blockIdx.y], 1);
}

extern "C" void ygz_hip_vocab_free(ygz_hip_ctx *ctx)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx->vocab) return;
    if (ctx->vocab->blob) (void)hipFree(ctx->vocab->blob);
    if (ctx->vocab->kp_word) (void)hipFree(ctx->vocab->kp_word);
    if (ctx->vocab->kp_node) (void)hipFree(ctx->vocab->kp_node);
    if (ctx->vocab->kp_weight) (void)hipFree(ctx->vocab->kp_weight);
    delete ctx->vocab; ctx->vocab = nullptr;
}

static VocDev voc_dev(const ygz_hip_ctx::Vocab *v)
{
    VocDev V; V.L = v->L; V.child_off = v->child_off; V.child = v->child; V.word_id = v->word_id; V.desc = v->desc; V.weight = v->weight;
    return V;
}

extern "C" {

int ygz_hip_vocab_load(ygz_hip_ctx *ctx, const void *blob, size_t bytes)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !blob || bytes < 24) return YGZ_E_INVALID;
    const uint8_t *p = (const uint8_t *)blob;
    uint32_t nb_nodes, size_node; int k, L;
    memcpy(&nb_nodes, p, 4); memcpy(&size_node, p + 4, 4); memcpy(&k, p + 8, 4); memcpy(&L, p + 12, 4);
    if (size_node < 41 || bytes < 24 + (size_t)nb_nodes * size_node || L < 1 || L > 10 || k < 1) return YGZ_E_INVALID;
    const int n = (int)nb_nodes + 1;
    std::vector<int32_t> parent(n, 0), word_id(n, -1), cnt(n, 0), child_off(n + 1, 0), child(n, 0);
    std::vector<uint32_t> desc((size_t)n * 8, 0);
    std::vector<double> weight(n, 0.0);
    int words = 0;
    for (int nid = 1; nid < n; ++nid) {                     // exactly nb_nodes records (the reference reads one too many)
        const uint8_t *rec = p + 24 + (size_t)(nid - 1) * size_node;
        int32_t par; float w;
        memcpy(&par, rec, 4); memcpy(&desc[(size_t)nid * 8], rec + 4, 32); memcpy(&w, rec + 36, 4);
        if (par < 0 || par >= nid) return YGZ_E_INVALID;
        parent[nid] = par; weight[nid] = (double)w;
        if (rec[40]) word_id[nid] = words++;
        cnt[par]++;
    }
    for (int i = 0; i < n; ++i) child_off[i + 1] = child_off[i] + cnt[i];
    std::fill(cnt.begin(), cnt.end(), 0);
    for (int nid = 1; nid < n; ++nid) { const int par = parent[nid]; child[child_off[par] + cnt[par]++] = nid; }
    ygz_hip_vocab_free(ctx);
    auto *v = new ygz_hip_ctx::Vocab();
    v->k = k; v->L = L; v->n_nodes = n; v->n_words = words;
    const size_t b_desc = (size_t)n * 32, b_w = (size_t)n * 8, b_i = (size_t)n * 4;
    hipError_t e = hipMalloc(&v->blob, b_desc + b_w + 3 * b_i + 4 + 64);
    if (e != hipSuccess) { ctx->last_hip_error = (int)e; delete v; return YGZ_E_HIP; }
    uint8_t *d = (uint8_t *)v->blob;
    v->desc = (uint32_t *)d; d += b_desc; v->weight = (double *)d; d += b_w;
    v->child_off = (int32_t *)d; d += b_i + 4; v->child = (int32_t *)d; d += b_i; v->word_id = (int32_t *)d;
    ctx->vocab = v;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(v->desc, desc.data(), b_desc, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(v->weight, weight.data(), b_w, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(v->child_off, child_off.data(), b_i + 4, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(v->child, child.data(), b_i, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(v->word_id, word_id.data(), b_i, hipMemcpyHostToDevice, ctx->stream));
    const size_t F = (size_t)ctx->prm.max_frames, Cn = (size_t)ctx->cells;
    YGZ_HIPCHK(ctx, hipMalloc((void **)&v->kp_word, F * Cn * 4));
    YGZ_HIPCHK(ctx, hipMalloc((void **)&v->kp_node, F * Cn * 4));
    YGZ_HIPCHK(ctx, hipMalloc((void **)&v->kp_weight, F * Cn * 8));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_vocab_info(ygz_hip_ctx *ctx, int *k, int *L, int *n_nodes, int *n_words)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !ctx->vocab) return YGZ_E_STATE;
    if (k) *k = ctx->vocab->k; if (L) *L = ctx->vocab->L; if (n_nodes) *n_nodes = ctx->vocab->n_nodes; if (n_words) *n_words = ctx->vocab->n_words;
    return YGZ_OK;
}

// Frame::ComputeBoW for the resident keypoints of slots slot_begin .. +n_slots-1
int ygz_hip_compute_bow(ygz_hip_ctx *ctx, int slot_begin, int n_slots, int levelsup)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || slot_begin < 0 || n_slots < 1 || slot_begin + n_slots > ctx->prm.max_frames || levelsup < 0) return YGZ_E_INVALID;
    if (!ctx->vocab) return YGZ_E_STATE;
    auto *v = ctx->vocab;
    const size_t off = (size_t)slot_begin * ctx->cells;
    YGZ_LAUNCH(ctx, KID_BOW_TRANSFORM, k_bow_transform, dim3(ygz_div_up(ctx->cells, 256), n_slots), dim3(256), voc_dev(v),
               ctx->kp_desc + 8 * off, ctx->n_kp + slot_begin, 0, (size_t)ctx->cells, levelsup, v->kp_word + off, v->kp_weight + off, v->kp_node + off);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

int ygz_hip_get_bow(ygz_hip_ctx *ctx, int slot, int32_t *word, double *weight, int32_t *node, int capacity, int *n)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || slot < 0 || slot >= ctx->prm.max_frames || capacity < 0) return YGZ_E_INVALID;
    if (!ctx->vocab) return YGZ_E_STATE;
    int cnt = 0;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&cnt, ctx->n_kp + slot, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (n) *n = cnt;
    if (cnt > capacity) return YGZ_E_CAPACITY;
    const size_t off = (size_t)slot * ctx->cells;
    if (cnt > 0) {
        if (word) YGZ_HIPCHK(ctx, hipMemcpyAsync(word, ctx->vocab->kp_word + off, (size_t)cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (weight) YGZ_HIPCHK(ctx, hipMemcpyAsync(weight, ctx->vocab->kp_weight + off, (size_t)cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (node) YGZ_HIPCHK(ctx, hipMemcpyAsync(node, ctx->vocab->kp_node + off, (size_t)cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return YGZ_OK;
}

// host descriptors in, host BoW out (what the class surface uses: Feature::_desc lives on the host)
int ygz_hip_bow_transform(ygz_hip_ctx *ctx, const uint8_t *desc, int n, int levelsup, int32_t *word, double *weight, int32_t *node)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || n < 0 || levelsup < 0 || (n > 0 && !desc)) return YGZ_E_INVALID;
    if (!ctx->vocab) return YGZ_E_STATE;
    if (n == 0) return YGZ_OK;
    uint8_t *buf = nullptr;
    const size_t N = (size_t)n;
    int rc = ygz_scratch(ctx, SCR_GEN_0, N * (32 + 4 + 8 + 4) + 64, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    uint32_t *d_desc = (uint32_t *)buf; double *d_w = (double *)(buf + N * 32); int32_t *d_word = (int32_t *)(buf + N * 40), *d_node = d_word + N;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_desc, desc, N * 32, hipMemcpyHostToDevice, ctx->stream));
    YGZ_LAUNCH(ctx, KID_BOW_TRANSFORM, k_bow_transform, dim3(ygz_div_up(n, 256), 1), dim3(256), voc_dev(ctx->vocab), d_desc, (const int32_t *)nullptr, n,
               (size_t)0, levelsup, d_word, d_w, d_node);
    YGZ_HIPCHK(ctx, hipGetLastError());
    if (word) YGZ_HIPCHK(ctx, hipMemcpyAsync(word, d_word, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (weight) YGZ_HIPCHK(ctx, hipMemcpyAsync(weight, d_w, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (node) YGZ_HIPCHK(ctx, hipMemcpyAsync(node, d_node, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

static int bow_match_launch(ygz_hip_ctx *ctx, int mode, const std::vector<BowPair> &pairs, int max_c1, int th_low, float knn_ratio, double eps,
                            int32_t *d_match, size_t stride, int32_t *d_counts)
{
    void *d_pairs = nullptr;
    int rc = ygz_scratch(ctx, SCR_BOW, pairs.size() * sizeof(BowPair), &d_pairs);
    if (rc != YGZ_OK) return rc;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_pairs, pairs.data(), pairs.size() * sizeof(BowPair), hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemsetAsync(d_counts, 0, pairs.size() * 4, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemsetAsync(d_match, 0xff, pairs.size() * stride * 4, ctx->stream));
    const dim3 grid(ygz_div_up(max_c1 > 0 ? max_c1 : 1, 256), (unsigned)pairs.size());
    const double fx = (double)ctx->prm.fx, fy = (double)ctx->prm.fy, cx = (double)ctx->prm.cx, cy = (double)ctx->prm.cy;
    if (mode == 0) YGZ_LAUNCH(ctx, KID_BOW_MATCH, k_bow_match<0>, grid, dim3(256), (const BowPair *)d_pairs, th_low, knn_ratio, eps, fx, fy, cx, cy, d_match, stride, d_counts);
    else YGZ_LAUNCH(ctx, KID_BOW_MATCH, k_bow_match<1>, grid, dim3(256), (const BowPair *)d_pairs, th_low, knn_ratio, eps, fx, fy, cx, cy, d_match, stride, d_counts);
    YGZ_HIPCHK(ctx, hipGetLastError());
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));        // `pairs` (host) may go out of scope
    return YGZ_OK;
}

// resident form: pairs of slots whose BoW was computed with ygz_hip_compute_bow.  mode 0 = SearchByBoW, 1 = SearchForTriangulation
// (E12 [n_pairs][9] row-major).  match12 [n_pairs][max_keypoints] (index in slot2 or -1), counts [n_pairs].
int ygz_hip_search_by_bow_slots(ygz_hip_ctx *ctx, int mode, int n_pairs, const int32_t *slot1, const int32_t *slot2, const double *E12,
                                int th_low, float knn_ratio, double epipolar_dsqr, int32_t *match12, int32_t *counts)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || n_pairs < 1 || !slot1 || !slot2 || (mode != 0 && mode != 1) || (mode == 1 && !E12)) return YGZ_E_INVALID;
    if (!ctx->vocab) return YGZ_E_STATE;
    std::vector<int32_t> nk(ctx->prm.max_frames);
    YGZ_HIPCHK(ctx, hipMemcpyAsync(nk.data(), ctx->n_kp, nk.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<BowPair> pairs(n_pairs);
    int max_c1 = 0;
    for (int p = 0; p < n_pairs; ++p) {
        const int a = slot1[p], b = slot2[p];
        if (a < 0 || a >= ctx->prm.max_frames || b < 0 || b >= ctx->prm.max_frames) return YGZ_E_INVALID;
        BowPair &P = pairs[p];
        const size_t oa = (size_t)a * ctx->cells, ob = (size_t)b * ctx->cells;
        P.d1 = ctx->kp_desc + 8 * oa; P.d2 = ctx->kp_desc + 8 * ob; P.n1 = ctx->vocab->kp_node + oa; P.n2 = ctx->vocab->kp_node + ob;
        P.p1 = ctx->kp_px + 2 * oa; P.p2 = ctx->kp_px + 2 * ob; P.c1 = nk[a]; P.c2 = nk[b];
        for (int i = 0; i < 9; ++i) P.E[i] = E12 ? E12[9 * (size_t)p + i] : 0.0;
        if (P.c1 > max_c1) max_c1 = P.c1;
    }
    const size_t stride = (size_t)ctx->cells;
    int32_t *d_match = nullptr;
    int rc = ygz_scratch(ctx, SCR_GEN_0, (size_t)n_pairs * (stride + 1) * 4 + 64, (void **)&d_match);
    if (rc != YGZ_OK) return rc;
    int32_t *d_counts = d_match + (size_t)n_pairs * stride;
    if ((rc = bow_match_launch(ctx, mode, pairs, max_c1, th_low, knn_ratio, epipolar_dsqr, d_match, stride, d_counts)) != YGZ_OK) return rc;
    if (match12) YGZ_HIPCHK(ctx, hipMemcpyAsync(match12, d_match, (size_t)n_pairs * stride * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (counts) YGZ_HIPCHK(ctx, hipMemcpyAsync(counts, d_counts, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

// ---- the checkOrientation part of Matcher::SearchByBoW (Matcher.cpp:247-256, 271-289): one workgroup per pair builds the 30-bin histogram of
// rot = angle1 - angle2 over the pair's matches, thread 0 takes the three maxima (ComputeThreeMaxima, :293-336) and the count that
// is left.  The reference leaves the matches themselves alone (its TODO at :284): only cnt_matches changes.
struct BowRotArgs {
    const double *a1_d, *a2_d; const float *a1_f, *a2_f;      // angles as doubles (host form) or as the extractor's floats (slot form)
    const int32_t *off1, *off2, *cnt1;                          // [pairs] first element of the pair's angle arrays, features of frame 1
    const int32_t *match12; size_t stride;                      // [pairs][stride]
    int32_t *kept, *hist, *ind;                                 // [pairs], [pairs][30], [pairs][3]
};
__global__ __launch_bounds__(256) void k_bow_orientation(BowRotArgs A)
{
    __shared__ int h[30];
    const int p = blockIdx.x, tid = threadIdx.x;
    if (tid < 30) h[tid] = 0;
    __syncthreads();
    const int n1 = A.cnt1[p];
    const int32_t *m = A.match12 + (size_t)p * A.stride;
    const float factor = 1.0f / 30;
    for (int i = tid; i < n1; i += 256) {
        const int j = m[i];
        if (j < 0) continue;
        const double x1 = A.a1_d ? A.a1_d[A.off1[p] + i] : (double)A.a1_f[A.off1[p] + i], x2 = A.a2_d ? A.a2_d[A.off2[p] + j] : (double)A.a2_f[A.off2[p] + j];
        float rot = (float)(x1 - x2);
        if (rot < 0) rot = __fadd_rn(rot, 360.f);
        int bin = (int)round((double)__fmul_rn(rot, factor));
        if (bin == 30) bin = 0;
        if (bin >= 0 && bin < 30) atomicAdd(&h[bin], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1, cnt = 0;
        for (int i = 0; i < 30; ++i) {
            const int s = h[i];
            cnt += s;
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
            else if (s > max3) { max3 = s; i3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) i3 = -1;
        for (int i = 0; i < 30; ++i) if (i != i1 && i != i2 && i != i3) cnt -= h[i];
        A.kept[p] = cnt; A.ind[3 * p] = i1; A.ind[3 * p + 1] = i2; A.ind[3 * p + 2] = i3;
    }
    if (tid < 30) A.hist[30 * p + tid] = h[tid];
}

// host arrays, one pair (the class surface: Feature::_angle is a double): angle1 [n1], angle2 [n2], match12 [n1] as ygz_hip_search_by_bow
// returned it.  kept = the count Matcher::SearchByBoW returns with checkOrientation on; hist [30] / maxima [3] may be NULL.
int ygz_hip_bow_orientation(ygz_hip_ctx *ctx, const double *angle1, int n1, const double *angle2, int n2, const int32_t *match12, int *kept,
                            int32_t *hist, int32_t *maxima)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || n1 < 0 || n2 < 0 || !kept) return YGZ_E_INVALID;
    *kept = 0;
    if (hist) for (int i = 0; i < 30; ++i) hist[i] = 0;
    if (maxima) maxima[0] = maxima[1] = maxima[2] = -1;
    if (n1 == 0) return YGZ_OK;
    if (!angle1 || !match12 || (n2 > 0 && !angle2)) return YGZ_E_INVALID;
    for (int i = 0; i < n1; ++i) if (match12[i] >= n2) return YGZ_E_INVALID;
    const size_t N1 = (size_t)n1, N2 = (size_t)(n2 > 0 ? n2 : 1);
    uint8_t *buf = nullptr;
    int rc = ygz_scratch(ctx, SCR_GEN_0, (N1 + N2) * 8 + N1 * 4 + 64 * 4, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    double *d_a1 = (double *)buf, *d_a2 = d_a1 + N1;
    int32_t *d_m = (int32_t *)(d_a2 + N2), *d_out = d_m + N1;      // off1, off2, cnt1, kept, ind[3], hist[30]
    const int32_t head[3] = { 0, 0, n1 };
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_a1, angle1, N1 * 8, hipMemcpyHostToDevice, ctx->stream));
    if (n2 > 0) YGZ_HIPCHK(ctx, hipMemcpyAsync(d_a2, angle2, (size_t)n2 * 8, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_m, match12, N1 * 4, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_out, head, sizeof(head), hipMemcpyHostToDevice, ctx->stream));
    BowRotArgs A;
    A.a1_d = d_a1; A.a2_d = d_a2; A.a1_f = A.a2_f = nullptr; A.off1 = d_out; A.off2 = d_out + 1; A.cnt1 = d_out + 2;
    A.match12 = d_m; A.stride = N1; A.kept = d_out + 3; A.ind = d_out + 4; A.hist = d_out + 8;
    YGZ_LAUNCH(ctx, KID_BOW_MATCH, k_bow_orientation, dim3(1), dim3(256), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    int32_t h[40];
    YGZ_HIPCHK(ctx, hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *kept = h[3];
    if (maxima) for (int i = 0; i < 3; ++i) maxima[i] = h[4 + i];
    if (hist) for (int i = 0; i < 30; ++i) hist[i] = h[8 + i];
    return YGZ_OK;
}

// slot form: the pairs and match12 [n_pairs][max_keypoints] of ygz_hip_search_by_bow_slots (mode 0), angles = the resident keypoints' (floats)
int ygz_hip_bow_orientation_slots(ygz_hip_ctx *ctx, int n_pairs, const int32_t *slot1, const int32_t *slot2, const int32_t *match12, int32_t *kept,
                                  int32_t *hist, int32_t *maxima)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || n_pairs < 1 || !slot1 || !slot2 || !match12 || !kept) return YGZ_E_INVALID;
    const size_t stride = (size_t)ctx->cells, NP = (size_t)n_pairs;
    std::vector<int32_t> nk(ctx->prm.max_frames), tab(3 * NP);
    YGZ_HIPCHK(ctx, hipMemcpyAsync(nk.data(), ctx->n_kp, nk.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int p = 0; p < n_pairs; ++p) {
        const int a = slot1[p], b = slot2[p];
        if (a < 0 || a >= ctx->prm.max_frames || b < 0 || b >= ctx->prm.max_frames) return YGZ_E_INVALID;
        tab[p] = (int32_t)((size_t)a * ctx->cells); tab[NP + p] = (int32_t)((size_t)b * ctx->cells); tab[2 * NP + p] = nk[a];
        for (int i = 0; i < nk[a]; ++i) if (match12[(size_t)p * stride + i] >= nk[b]) return YGZ_E_INVALID;
    }
    int32_t *d = nullptr;
    int rc = ygz_scratch(ctx, SCR_GEN_0, (NP * stride + NP * (3 + 1 + 3 + 30)) * 4 + 64, (void **)&d);
    if (rc != YGZ_OK) return rc;
    int32_t *d_tab = d + NP * stride, *d_kept = d_tab + 3 * NP, *d_ind = d_kept + NP, *d_hist = d_ind + 3 * NP;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d, match12, NP * stride * 4, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    BowRotArgs A;
    A.a1_d = A.a2_d = nullptr; A.a1_f = ctx->kp_angle; A.a2_f = ctx->kp_angle; A.off1 = d_tab; A.off2 = d_tab + NP; A.cnt1 = d_tab + 2 * NP;
    A.match12 = d; A.stride = stride; A.kept = d_kept; A.ind = d_ind; A.hist = d_hist;
    YGZ_LAUNCH(ctx, KID_BOW_MATCH, k_bow_orientation, dim3(n_pairs), dim3(256), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    YGZ_HIPCHK(ctx, hipMemcpyAsync(kept, d_kept, NP * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (maxima) YGZ_HIPCHK(ctx, hipMemcpyAsync(maxima, d_ind, NP * 12, hipMemcpyDeviceToHost, ctx->stream));
    if (hist) YGZ_HIPCHK(ctx, hipMemcpyAsync(hist, d_hist, NP * 120, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

// host-array form of one pair (the class surface): descriptors, FeatureVector nodes (and pixels for mode 1) of both frames
int ygz_hip_search_by_bow(ygz_hip_ctx *ctx, int mode, const uint8_t *desc1, const int32_t *node1, const double *px1, int n1,
                          const uint8_t *desc2, const int32_t *node2, const double *px2, int n2, const double *E12,
                          int th_low, float knn_ratio, double epipolar_dsqr, int32_t *match12, int *count)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || n1 < 0 || n2 < 0 || (mode != 0 && mode != 1) || (mode == 1 && (!E12 || (n1 > 0 && !px1) || (n2 > 0 && !px2)))) return YGZ_E_INVALID;
    if (count) *count = 0;
    if (n1 == 0) return YGZ_OK;
    if (!desc1 || !node1 || (n2 > 0 && (!desc2 || !node2))) return YGZ_E_INVALID;
    const size_t N1 = (size_t)n1, N2 = (size_t)(n2 > 0 ? n2 : 1);
    uint8_t *buf = nullptr;
    int rc = ygz_scratch(ctx, SCR_GEN_0, (N1 + N2) * (32 + 4 + 16) + N1 * 4 + 64 + 16, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    BowPair P;
    uint8_t *d = buf;
    P.p1 = (const double *)d; d += N1 * 16; P.p2 = (const double *)d; d += N2 * 16;
    P.d1 = (const uint32_t *)d; d += N1 * 32; P.d2 = (const uint32_t *)d; d += N2 * 32;
    P.n1 = (const int32_t *)d; d += N1 * 4; P.n2 = (const int32_t *)d; d += N2 * 4;
    int32_t *d_match = (int32_t *)d; d += N1 * 4; int32_t *d_counts = (int32_t *)d;
    P.c1 = n1; P.c2 = n2;
    for (int i = 0; i < 9; ++i) P.E[i] = E12 ? E12[i] : 0.0;
    YGZ_HIPCHK(ctx, hipMemcpyAsync((void *)P.d1, desc1, N1 * 32, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync((void *)P.n1, node1, N1 * 4, hipMemcpyHostToDevice, ctx->stream));
    if (px1) YGZ_HIPCHK(ctx, hipMemcpyAsync((void *)P.p1, px1, N1 * 16, hipMemcpyHostToDevice, ctx->stream));
    if (n2 > 0) {
        YGZ_HIPCHK(ctx, hipMemcpyAsync((void *)P.d2, desc2, (size_t)n2 * 32, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync((void *)P.n2, node2, (size_t)n2 * 4, hipMemcpyHostToDevice, ctx->stream));
        if (px2) YGZ_HIPCHK(ctx, hipMemcpyAsync((void *)P.p2, px2, (size_t)n2 * 16, hipMemcpyHostToDevice, ctx->stream));
    }
    std::vector<BowPair> pairs(1, P);
    if ((rc = bow_match_launch(ctx, mode, pairs, n1, th_low, knn_ratio, epipolar_dsqr, d_match, N1, d_counts)) != YGZ_OK) return rc;
    int c = 0;
    if (match12) YGZ_HIPCHK(ctx, hipMemcpyAsync(match12, d_match, N1 * 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&c, d_counts, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (count) *count = c;
    return YGZ_OK;
}

}  // extern "C"
