// The batched offline run of BASELINE configs[4] (include/ygz_offline.h) -- host code of libygz_host.so: plain C++ over the C ABI of
// libygz_hip.so, collectives through RCCL (librccl.so.1, bound at run time so that single-GPU users need no RCCL), no Python, no torch.
//
// It plays the part of the reference's callers for a whole sequence at once: the per-frame loop test/test_vo_track.cpp:100-113 ->
// VisualOdometry::AddFrame (src/Module/VisualOdometry.cpp:38-107) and the local-BA round LocalMapping::LocalBA
// (src/Module/LocalMapping.cpp:149-208, 301-336 -> ba::LocalBAG2O, src/Algorithm/BA.cpp:386-543).  What runs per frame pair and per window
// is listed at the head of include/ygz_offline.h; this file holds the schedule: shard, chunk plan, lanes, window readiness, LM launches,
// the two exchanges.  No arithmetic of the path lives here -- every number comes out of a kernel -- so any schedule gives the same bits.
#include "ygz_offline.h"
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// RCCL development headers are not installed: the few declarations of nccl.h the driver uses (the library itself is bound at run time with
// dlopen, so the single-GPU class surface builds and runs without RCCL).  Values as in RCCL 2.x's nccl.h.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0 } ncclDataType_t;
}
#endif
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

typedef std::pair<int, int> Range;                  // frames [first, second)
typedef std::vector<Range> ChunkRanges;             // a chunk: one range, or several short ones (deferred gaps)
static const double kI7[7] = { 0, 0, 0, 1, 0, 0, 0 };

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- partition (frames are independent units; pair (f, f - 1) belongs to the rank that owns f) -------------------------------
void shard_frames(int n_total, int rank, int world, int &start, int &count, int &halo)
{
    const int base = n_total / world, rem = n_total % world;
    count = base + (rank < rem ? 1 : 0);
    start = rank * base + std::min(rank, rem);
    halo = (start > 0 && count > 0) ? 1 : 0;
}
int frame_owner(int f, int n_total, int world)
{
    for (int r = 0; r < world; ++r) { int s, c, h; shard_frames(n_total, r, world, s, c, h); if (f >= s && f < s + c) return r; }
    return -1;
}
// non-overlapping windows of window_kfs consecutive keyframes; a trailing window needs two
std::vector<std::vector<int>> ba_windows(int n_total, int kf_stride, int window_kfs)
{
    std::vector<int> kfs;
    for (int f = 0; f < n_total; f += kf_stride) kfs.push_back(f);
    std::vector<std::vector<int>> out;
    for (size_t a = 0; a < kfs.size(); a += (size_t)window_kfs) {
        std::vector<int> w(kfs.begin() + a, kfs.begin() + std::min(kfs.size(), a + (size_t)window_kfs));
        if (w.size() >= 2) out.push_back(w);
    }
    return out;
}

// ---- the plan of a shard's chunks.  Every pair is solved from the identity, so the ORDER of the chunks is free; it decides when a BA
// window is complete and therefore where its resident-LM launch (a latency chain that uses a fraction of the GPU) falls. ----------------
// [first, last) in chunks of `chunk` frames with short chunks at both ends (nothing overlaps the upload of the first chunk or the kernels
// of the last); kf_stride > 0: the frames behind the shard's last keyframe complete no window and form a chunk of their own at the end
std::vector<Range> chunk_schedule(int first, int last, int chunk, bool ramp, int kf_stride, int ramp_from)
{
    const int n = last - first;
    std::vector<int> sizes;
    if (n > 0) {
        if (ramp && chunk >= 64 && n >= ramp_from * chunk) {
            const int head[2] = { chunk / 4, chunk / 2 }, tail[2] = { chunk / 2, chunk / 4 };
            const int body = n - head[0] - head[1] - tail[0] - tail[1];
            sizes.push_back(head[0]); sizes.push_back(head[1]);
            for (int i = 0; i < body / chunk; ++i) sizes.push_back(chunk);
            if (body % chunk) sizes.push_back(body % chunk);
            sizes.push_back(tail[0]); sizes.push_back(tail[1]);
        } else {
            for (int i = 0; i < n / chunk; ++i) sizes.push_back(chunk);
            if (n % chunk) sizes.push_back(n % chunk);
        }
    }
    std::vector<Range> out;
    int c0 = first;
    for (int s : sizes) { out.push_back(Range(c0, c0 + s)); c0 += s; }
    if (kf_stride > 0 && !out.empty()) {
        const int a = out.back().first, b = out.back().second;
        const int k_last = ((b - 1) / kf_stride) * kf_stride;             // the last keyframe of the shard
        if (a <= k_last && k_last + 1 < b && k_last + 1 > a) { out.back() = Range(a, k_last + 1); out.push_back(Range(k_last + 1, b)); }
    }
    return out;
}
// For the last `defer` windows that end inside the shard, the kf_stride - 1 frames between a window's last keyframe and the next anchor are
// taken out of the main pass and processed at the very end, about `group` frames per chunk (several short ranges per chunk): the last LM
// launch then runs beside their uploads and kernels.  Costs two more halo frames per gap, which is why short shards keep the plain plan.
std::vector<ChunkRanges> chunk_plan(int first, int last, int chunk, bool ramp, int kf_stride, const std::vector<std::vector<int>> &windows, int defer,
                                    int group = 45)
{
    std::vector<ChunkRanges> plain;
    for (const Range &r : chunk_schedule(first, last, chunk, ramp, kf_stride, 4)) plain.push_back(ChunkRanges(1, r));
    if (defer <= 0) return plain;
    std::vector<const std::vector<int> *> inside;
    std::vector<int> anchors;
    for (const auto &w : windows) { anchors.push_back(w[0]); if (w[0] >= first && w.back() < last) inside.push_back(&w); }
    std::sort(anchors.begin(), anchors.end());
    std::vector<Range> gaps;
    for (size_t i = inside.size() > (size_t)defer ? inside.size() - defer : 0; i < inside.size(); ++i) {
        const std::vector<int> &w = *inside[i];
        int nxt = last;
        for (int a : anchors) if (a > w.back()) { nxt = a; break; }
        const int g0 = w.back() + 1, g1 = std::min(last, nxt);
        if (g1 > g0 && g0 > first) gaps.push_back(Range(g0, g1));
    }
    if (gaps.empty()) return plain;
    std::vector<Range> main;
    int a = first;
    for (const Range &g : gaps) { if (g.first > a) main.push_back(Range(a, g.first)); a = g.second; }
    if (a < last) main.push_back(Range(a, last));
    int n_main = 0;
    for (const Range &r : main) n_main += r.second - r.first;
    std::vector<ChunkRanges> out;
    for (const Range &v : chunk_schedule(0, n_main, chunk, ramp, 0, 3)) {      // the schedule of a shard of n_main frames mapped back onto what is left
        ChunkRanges rs;
        int pos = 0;
        for (const Range &r : main) {
            const int lo = std::max(v.first, pos), hi = std::min(v.second, pos + (r.second - r.first));
            if (hi > lo) rs.push_back(Range(r.first + lo - pos, r.first + hi - pos));
            pos += r.second - r.first;
        }
        out.push_back(rs);
    }
    int tot = 0;
    for (const Range &g : gaps) tot += g.second - g.first;
    const int n_groups = std::max(1, (int)std::nearbyint(tot / (double)std::max(1, group)));     // (round half to even, as the Python plan did)
    const int per = (int)((gaps.size() + n_groups - 1) / n_groups);
    for (size_t k = 0; k < gaps.size(); k += per) {
        ChunkRanges ch;
        for (size_t i = k; i < std::min(gaps.size(), k + per); ++i) {
            int g0 = gaps[i].first; const int g1 = gaps[i].second;
            while (g1 - g0 > chunk) { out.push_back(ChunkRanges(1, Range(g0, g0 + chunk))); g0 += chunk; }     // (a gap longer than a chunk)
            ch.push_back(Range(g0, g1));
        }
        auto total = [&] { int t = 0; for (const Range &r : ch) t += r.second - r.first; return t; };
        while (total() > chunk) { out.push_back(ChunkRanges(1, ch.front())); ch.erase(ch.begin()); }               // never more than `chunk` frames per chunk
        out.push_back(ch);
    }
    return out;
}

struct Plan {
    int start = 0, count = 0, halo = 0;
    std::vector<std::vector<int>> wins;
    std::vector<ChunkRanges> chunks;
    int defer = 0;
};
int make_plan(const ygz_offline_params &p, Plan &P)
{
    if (p.n_frames < 1 || p.world < 1 || p.rank < 0 || p.rank >= p.world || p.chunk < 1 || p.kf_stride < 1 || p.window_kfs < 2) return YGZ_E_INVALID;
    shard_frames(p.n_frames, p.rank, p.world, P.start, P.count, P.halo);
    P.wins = ba_windows(p.n_frames, p.kf_stride, p.window_kfs);
    if (p.defer_gaps >= 0) P.defer = p.defer_gaps;
    else {
        // measured (tools/gpu_r05.sh k): a long shard hides its last LM launch beside 13 deferred gaps (59.2 against 60.6 ms per
        // 1024 frames); a SHORT shard (one rank of an 8-rank job: 128 frames, two windows) defers every gap -- its one LM launch, spread over the
        // XCDs, then runs beside the gap chunk (11.6 against 12.3 ms); in between the two extra halo frames per gap cost more than they hide
        int inside = 0;
        for (const auto &w : P.wins) if (w[0] >= P.start && w.back() < P.start + P.count) ++inside;
        P.defer = P.count >= 768 ? 13 : (P.count <= 160 ? inside : 0);
    }
    P.chunks = chunk_plan(P.start, P.start + P.count, p.chunk, p.ramp != 0, p.kf_tail ? p.kf_stride : 0, P.wins, p.pipeline_ba ? P.defer : 0);
    return YGZ_OK;
}

// ---- RCCL, bound at run time ------------------------------------------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl &rccl()
{
    static Rccl R;
    if (R.lib || R.ok) return R;
    // librccl.so.1 by soname: a process that already mapped one (torch ships its own copy) gets that one, bound to the HIP runtime in use
    for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { R.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (R.lib) break; }
    if (!R.lib) return R;
#define YGZ_SYM(field, sym) *(void **)(&R.field) = dlsym(R.lib, sym)
    YGZ_SYM(GetUniqueId, "ncclGetUniqueId"); YGZ_SYM(CommInitRank, "ncclCommInitRank"); YGZ_SYM(CommDestroy, "ncclCommDestroy");
    YGZ_SYM(AllGather, "ncclAllGather"); YGZ_SYM(Send, "ncclSend"); YGZ_SYM(Recv, "ncclRecv"); YGZ_SYM(GroupStart, "ncclGroupStart");
    YGZ_SYM(GroupEnd, "ncclGroupEnd"); YGZ_SYM(GetErrorString, "ncclGetErrorString");
#undef YGZ_SYM
    R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.AllGather && R.Send && R.Recv && R.GroupStart && R.GroupEnd;
    return R;
}

struct Chunk {
    ChunkRanges ranges;
    std::vector<int32_t> frames;                 // slot k holds frame frames[k] (every range with its halo frame in front)
    std::vector<std::pair<int, int>> spans;      // (first slot, frames) per range
    std::vector<int32_t> pairs;                  // [n_pairs][2] (cur, ref) frames
    std::vector<int32_t> q, t;                   // their slots
    std::vector<int32_t> kf_slot, kf_row;        // keyframes of the chunk: slot, store row
    std::vector<std::pair<int, int>> trel_runs;  // (first pair, pairs) of consecutive frames
    double *pin_sum = nullptr; int32_t *pin_cnt = nullptr;     // page-locked result rows
};

}  // namespace

struct ygz_offline {
    ygz_offline_params p;
    Plan plan;
    std::vector<Chunk> chunks;
    std::vector<ygz_hip_ctx *> lanes;
    ygz_hip_ctx *ba = nullptr;
    std::vector<int> owner, mine, local;         // per window: owning rank; this rank's windows; those whose keyframes this rank tracks itself
    bool any_cross = false;
    int n_kf = 0, build_group = 1, lm_group = 1, S = 0;
    std::vector<double> ident;                   // identity poses for track_begin
    // exchange
    int backend = 0;
    ygz_offline_exchange hook = { nullptr, nullptr, nullptr };
    ncclComm_t comm = nullptr;
    void *ba_stream = nullptr;
    void *d_send = nullptr, *d_recv = nullptr; size_t x_bytes = 0;     // device exchange buffers (RCCL)
    uint8_t *h_send = nullptr, *h_recv = nullptr;                       // page-locked host exchange buffers
    // run state
    const uint8_t *frames = nullptr; const uint8_t *depth = nullptr; int first_in_buffer = 0;
    std::vector<uint8_t> tracked, ba_done;       // per frame; per window
    std::vector<int> ba_built;
    ygz_hip_ctx *last_upload = nullptr;
    // results
    std::vector<double> T_rel, traj, summary, wstate;
    std::vector<int32_t> n_kp, wowner, wkfs;
    int lm_launches = 0, lm_retries = 0, n_degenerate = 0;
    double ms_track = 0, ms_gather = 0, ms_ba_tail = 0, ms_exchange = 0;
    ygz_offline_chunk_fn cb = nullptr; void *cb_user = nullptr;
    bool trace = false; double t_run0 = 0;         // YGZ_OFFLINE_TRACE=1: where the host thread spends a run (stderr)
    std::string err;
};

namespace {

#define OCHK(o, call, what) do { const int rc_ = (call); if (rc_ != YGZ_OK) { (o)->err = std::string(what) + ": " + ygz_hip_error_string(rc_); return rc_; } } while (0)
#define NCHK(o, call, what) do { const ncclResult_t r_ = (call); if (r_ != ncclSuccess) { (o)->err = std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r_) : "RCCL error"); return YGZ_E_HIP; } } while (0)

size_t frame_bytes(const ygz_offline_params &p) { return (size_t)p.width * p.height * (p.frame_channels == 1 ? 1 : 3); }
size_t depth_bytes(const ygz_offline_params &p) { return (size_t)p.depth_w * p.depth_h * (p.depth_kind == 0 ? 4 : p.depth_kind == 1 ? 2 : 8); }

int create_ctx(ygz_offline *o, int max_frames, ygz_hip_ctx **out)
{
    ygz_hip_params prm;
    ygz_hip_default_params(&prm);
    prm.image_width = o->p.width; prm.image_height = o->p.height; prm.pyramid_levels = o->p.levels; prm.max_frames = max_frames;
    OCHK(o, ygz_hip_create(out, o->p.device, &prm, nullptr), "ygz_hip_create");
    return YGZ_OK;
}

// ---- the exchanges ----------------------------------------------------------------------------------------------------------------------
// Ragged all-gather of row blocks in rank order: rank r contributes counts[r] rows of row_bytes; `full` receives the concatenation
// (sum(counts) rows).  ONE collective of fixed-shape blocks (every block padded to max(counts) rows).  Host form, through the hook:
int ragged_gather_host(const ygz_offline_exchange &hook, int rank, int world, const int32_t *counts, size_t row_bytes, const void *local, void *full,
                       std::vector<uint8_t> &send, std::vector<uint8_t> &recv)
{
    int mx = 1;
    for (int r = 0; r < world; ++r) mx = std::max(mx, (int)counts[r]);
    const size_t per = (size_t)mx * row_bytes;
    send.assign(per, 0); recv.resize(per * world);
    if (counts[rank]) memcpy(send.data(), local, (size_t)counts[rank] * row_bytes);
    if (hook.all_gather(hook.user, send.data(), recv.data(), per) != 0) return YGZ_E_HIP;
    uint8_t *dst = (uint8_t *)full;
    for (int r = 0; r < world; ++r) { memcpy(dst, recv.data() + (size_t)r * per, (size_t)counts[r] * row_bytes); dst += (size_t)counts[r] * row_bytes; }
    return YGZ_OK;
}
// The driver's form: `local` in page-locked host memory (h_send) or, with src_on_device, already in d_send (k_ba_pack wrote it there: RCCL
// path only).  RCCL: ncclAllGather of the padded blocks on the BA context's stream, one copy back, unpacked on the host.
int ragged_gather(ygz_offline *o, const std::vector<int32_t> &counts, size_t row_bytes, bool src_on_device, void *full)
{
    const int W = o->p.world;
    int mx = 1;
    for (int r = 0; r < W; ++r) mx = std::max(mx, (int)counts[r]);
    const size_t per = (size_t)mx * row_bytes;
    if (per * W > o->x_bytes) { o->err = "exchange buffer too small"; return YGZ_E_CAPACITY; }
    if (o->backend == 2) {
        OCHK(o, ygz_hip_synchronize(o->ba), "synchronize");
        std::vector<uint8_t> send, recv;
        if (ragged_gather_host(o->hook, o->p.rank, W, counts.data(), row_bytes, o->h_send, full, send, recv) != YGZ_OK) { o->err = "exchange hook: all_gather failed"; return YGZ_E_HIP; }
        return YGZ_OK;
    }
    if (!src_on_device) OCHK(o, ygz_hip_copy(o->ba, o->d_send, o->h_send, (size_t)counts[o->p.rank] * row_bytes, 0, 0), "copy");
    OCHK(o, ygz_hip_make_current(o->ba), "make_current");
    NCHK(o, rccl().AllGather(o->d_send, o->d_recv, per, ncclChar, o->comm, (hipStream_t)o->ba_stream), "ncclAllGather");
    OCHK(o, ygz_hip_copy(o->ba, o->h_recv, o->d_recv, per * W, 1, 1), "copy");
    uint8_t *dst = (uint8_t *)full;
    for (int r = 0; r < W; ++r) { memcpy(dst, o->h_recv + (size_t)r * per, (size_t)counts[r] * row_bytes); dst += (size_t)counts[r] * row_bytes; }
    return YGZ_OK;
}

// keyframe rows of windows that straddle a shard boundary: row of frame f from its owner to the owner of the window's anchor, point to
// point (ncclSend / ncclRecv on the store's own memory in one group; through the hook: device -> host -> hook -> host -> device)
int exchange_rows(ygz_offline *o)
{
    if (o->backend == 0 || !o->any_cross) return YGZ_OK;
    void *rows = nullptr; size_t rb = 0;
    OCHK(o, ygz_hip_kf_store_info(o->ba, &rows, &rb, nullptr, nullptr, nullptr), "kf_store_info");
    struct X { int row, src, dst; };
    std::vector<X> xs;
    for (size_t wi = 0; wi < o->plan.wins.size(); ++wi)
        for (int f : o->plan.wins[wi]) {
            const int r = frame_owner(f, o->p.n_frames, o->p.world);
            if (r != o->owner[wi]) xs.push_back({ f / o->p.kf_stride, r, o->owner[wi] });
        }
    const int me = o->p.rank;
    OCHK(o, ygz_hip_synchronize(o->ba), "synchronize");           // (the lanes were synchronised when their chunks were collected: the rows are complete)
    if (o->backend == 2) {
        std::vector<uint8_t> tmp(rb);
        for (const X &x : xs) {
            if (x.src != me && x.dst != me) continue;
            uint8_t *dev = (uint8_t *)rows + (size_t)x.row * rb;
            if (x.src == me) OCHK(o, ygz_hip_copy(o->ba, tmp.data(), dev, rb, 1, 1), "copy");
            if (o->hook.send_recv(o->hook.user, tmp.data(), rb, x.src, x.dst) != 0) { o->err = "exchange hook: send_recv failed"; return YGZ_E_HIP; }
            if (x.dst == me) OCHK(o, ygz_hip_copy(o->ba, dev, tmp.data(), rb, 0, 1), "copy");
        }
    } else {
        OCHK(o, ygz_hip_make_current(o->ba), "make_current");
        NCHK(o, rccl().GroupStart(), "ncclGroupStart");
        for (const X &x : xs) {
            uint8_t *dev = (uint8_t *)rows + (size_t)x.row * rb;
            if (x.src == me) NCHK(o, rccl().Send(dev, rb, ncclChar, x.dst, o->comm, (hipStream_t)o->ba_stream), "ncclSend");
            if (x.dst == me) NCHK(o, rccl().Recv(dev, rb, ncclChar, x.src, o->comm, (hipStream_t)o->ba_stream), "ncclRecv");
        }
        NCHK(o, rccl().GroupEnd(), "ncclGroupEnd");
    }
    OCHK(o, ygz_hip_kf_store_refresh(o->ba), "kf_store_refresh");
    return YGZ_OK;
}

// ---- one chunk: uploads, the batched kernels, keyframe rows and relative poses into the store, result rows -- all asynchronous ------------
void trace_line(const ygz_offline *o, const char *what, int ci, double t0)
{
    if (o->trace) fprintf(stderr, "[offline host] %8.3f  %-10s chunk %2d  %6.3f ms\n", t0 - o->t_run0, what, ci, now_ms() - t0);
}

int enqueue_chunk(ygz_offline *o, int ci)
{
    const double t_enq = now_ms();
    Chunk &c = o->chunks[ci];
    ygz_hip_ctx *L = o->lanes[ci % o->lanes.size()];
    const ygz_offline_params &p = o->p;
    const int n = (int)c.frames.size();
    if (o->last_upload) OCHK(o, ygz_hip_wait_mark(L, o->last_upload), "wait_mark");       // uploads cross PCIe one after the other, each at the full rate
    const size_t fb = frame_bytes(p), db = depth_bytes(p);
    for (const auto &sp : c.spans) {
        const int f0 = c.frames[sp.first];
        const size_t k = (size_t)(f0 - o->first_in_buffer);
        if (p.frame_channels == 1) OCHK(o, ygz_hip_upload_gray_batch(L, sp.first, sp.second, o->frames + k * fb, 0), "upload_gray_batch");
        else OCHK(o, ygz_hip_upload_bgr_batch(L, sp.first, sp.second, o->frames + k * fb, 0), "upload_bgr_batch");
        OCHK(o, ygz_hip_upload_depth_batch(L, sp.first, sp.second, o->depth + k * db, p.depth_w, p.depth_h, p.depth_kind, p.depth_scale, 0), "upload_depth_batch");
    }
    OCHK(o, ygz_hip_mark(L), "mark");
    o->last_upload = L;
    OCHK(o, ygz_hip_build_pyramid(L, 0, n, p.frame_channels == 1 ? 0 : 1), "build_pyramid");          // Frame::InitFrame
    OCHK(o, ygz_hip_detect(L, 0, n, nullptr), "detect");                                                // FeatureDetector::Detect
    OCHK(o, ygz_hip_keypoint_depths_from_image(L, 0, n), "keypoint_depths_from_image");                 // Feature::_depth / _mappoint of the fresh keypoints
    const int np = (int)c.q.size();
    if (np) {
        OCHK(o, ygz_hip_match_slots(L, c.q.data(), c.t.data(), np, 1), "match_slots");                  // cv::BFMatcher(crossCheck)
        OCHK(o, ygz_hip_match_postfilter(L, 20.0, 50.0, 3.0), "match_postfilter");                      // test_orb_match.cpp:95-104
        OCHK(o, ygz_hip_track_begin(L, c.q.data(), c.t.data(), o->ident.data(), o->ident.data(), np, 0), "track_begin");
        OCHK(o, ygz_hip_track_sparse_align(L, 2, 0, 30), "track_sparse_align");                         // TrackRefFrame (Matcher.cpp:18)
        ygz_klt_params kp;
        ygz_hip_default_klt_params(&kp);
        OCHK(o, ygz_hip_track_klt(L, &kp), "track_klt");                                                // Tracker::TrackKLT
        OCHK(o, ygz_hip_track_adopt_pose(L), "track_adopt_pose");                                       // TrackRefFrame -> TrackLocalMap
        OCHK(o, ygz_hip_track_direct(L), "track_direct");                                               // ProjectMapPoints
        OCHK(o, ygz_hip_track_pose_only(L), "track_pose_only");                                         // OptimizeCurrentPoseOnly
        int got = 0;
        OCHK(o, ygz_hip_track_get_summary(L, c.pin_sum, std::max(2, n), &got, 0), "track_get_summary");
        for (const auto &run : c.trel_runs) OCHK(o, ygz_hip_kf_store_put_trel(o->ba, L, run.first, run.second, c.pairs[2 * (size_t)run.first]), "kf_store_put_trel");
    }
    OCHK(o, ygz_hip_get_keypoint_counts(L, 0, n, c.pin_cnt, 0), "get_keypoint_counts");
    if (!c.kf_slot.empty()) OCHK(o, ygz_hip_kf_store_put(o->ba, L, (int)c.kf_slot.size(), c.kf_slot.data(), c.kf_row.data()), "kf_store_put");
    trace_line(o, "enqueue", ci, t_enq);
    return YGZ_OK;
}

// wait for the chunk's lane, then take its result rows
int collect_chunk(ygz_offline *o, int ci)
{
    Chunk &c = o->chunks[ci];
    ygz_hip_ctx *L = o->lanes[ci % o->lanes.size()];
    const double t_col = now_ms();
    OCHK(o, ygz_hip_synchronize(L), "synchronize");
    trace_line(o, "wait", ci, t_col);
    for (size_t si = 0; si < c.spans.size(); ++si) {                          // the frames of the chunk proper (a halo frame belongs to another chunk)
        const int k0 = c.spans[si].first + (c.ranges[si].first > 0 ? 1 : 0);
        for (int k = k0; k < c.spans[si].first + c.spans[si].second; ++k) o->n_kp[c.frames[k]] = c.pin_cnt[k];
    }
    for (size_t pi = 0; pi < c.q.size(); ++pi)
        memcpy(&o->summary[(size_t)c.pairs[2 * pi] * YGZ_OFFLINE_SUMMARY_FIELDS], c.pin_sum + pi * YGZ_OFFLINE_SUMMARY_FIELDS, YGZ_OFFLINE_SUMMARY_FIELDS * 8);
    if (o->cb) o->cb(o->cb_user, L, ci, (int)c.frames.size(), c.frames.data(), (int)c.q.size(), c.pairs.data());
    return YGZ_OK;
}

// optimize(20) + the inlier test of BA.cpp:503-515 (+ a second optimisation without the outliers), asynchronous
int lm(ygz_offline *o, int slot0, int n)
{
    OCHK(o, ygz_hip_ba_optimize_resident(o->ba, slot0, n, o->p.ba_iterations, nullptr), "ba_optimize_resident");
    OCHK(o, ygz_hip_ba_mark_outliers(o->ba, slot0, n, o->p.outlier_chi2, o->p.ba_rounds == 2), "ba_mark_outliers");
    if (o->p.ba_rounds == 2) {
        OCHK(o, ygz_hip_ba_optimize_resident(o->ba, slot0, n, o->p.ba_iterations, nullptr), "ba_optimize_resident");
        OCHK(o, ygz_hip_ba_mark_outliers(o->ba, slot0, n, o->p.outlier_chi2, 0), "ba_mark_outliers");
    }
    return YGZ_OK;
}

// build (+ optimise) owned windows [slot0, slot0 + n) on the BA context, behind everything the lanes have enqueued so far
int ba_launch(ygz_offline *o, int slot0, int n, bool optimize)
{
    if (n <= 0) return YGZ_OK;
    for (ygz_hip_ctx *L : o->lanes) OCHK(o, ygz_hip_stream_wait(o->ba, L), "stream_wait");
    const int K = o->p.window_kfs;
    std::vector<int32_t> kfi, kff, nk;
    for (int g0 = 0; g0 < n; g0 += o->build_group) {
        const int g = std::min(o->build_group, n - g0);
        kfi.assign((size_t)g * K, 0); kff.assign((size_t)g * K, 0); nk.assign(g, 0);
        for (int a = 0; a < g; ++a) {
            const std::vector<int> &w = o->plan.wins[o->mine[slot0 + g0 + a]];
            nk[a] = (int)w.size();
            for (size_t j = 0; j < w.size(); ++j) { kff[(size_t)a * K + j] = w[j]; kfi[(size_t)a * K + j] = w[j] / o->p.kf_stride; }
        }
        OCHK(o, ygz_hip_ba_build_windows(o->ba, slot0 + g0, g, kfi.data(), kff.data(), nk.data(), o->p.obs_mode ? 1 : 0), "ba_build_windows");
        if (optimize) { const int rc = lm(o, slot0 + g0, g); if (rc != YGZ_OK) return rc; }
    }
    if (optimize) for (int k = slot0; k < slot0 + n; ++k) o->ba_done[k] = 1;
    return YGZ_OK;
}

int retry_windows(ygz_offline *o, const int32_t *slots, int n)
{
    if (n <= 0) return YGZ_OK;
    o->lm_retries += n;
    OCHK(o, ygz_hip_ba_set_team_budget(o->ba, 1), "ba_set_team_budget");        // one workgroup per window: needs no co-residency
    for (int i = 0; i < n; ++i) { const int rc = ba_launch(o, slots[i], 1, true); if (rc != YGZ_OK) return rc; }
    OCHK(o, ygz_hip_ba_set_team_budget(o->ba, 0), "ba_set_team_budget");
    OCHK(o, ygz_hip_synchronize(o->ba), "synchronize");
    std::vector<ygz_ba_stats> st(1);
    for (int i = 0; i < n; ++i) {
        const int rc = ygz_hip_ba_get_stats(o->ba, slots[i], 1, st.data(), nullptr);
        if ((rc != YGZ_OK && rc != YGZ_E_HIP && rc != YGZ_E_STATE) || st[0].iterations < 0) {
            o->err = "BA window " + std::to_string(o->mine[slots[i]]) + ": the resident LM did not finish even with one workgroup";
            return YGZ_E_HIP;
        }
    }
    return YGZ_OK;
}

// A resident-LM team whose members were not co-resident within the spin bound of a team barrier (possible beside the tracking kernels of
// several lanes) leaves without a result: its window is rebuilt (the loop updates the points in place) and solved once more by ONE workgroup
int retry_timed_out(ygz_offline *o)
{
    const int n = (int)o->mine.size();
    if (!n) return YGZ_OK;
    std::vector<ygz_ba_stats> st(n);
    const int rc = ygz_hip_ba_get_stats(o->ba, 0, n, st.data(), nullptr);
    if (rc != YGZ_OK && rc != YGZ_E_HIP && rc != YGZ_E_STATE) OCHK(o, rc, "ba_get_stats");
    std::vector<int32_t> bad;
    for (int k = 0; k < n; ++k) if (st[k].iterations < 0 && o->ba_done[k]) bad.push_back(k);
    return retry_windows(o, bad.data(), (int)bad.size());
}

int lm_next(const ygz_offline *o) { return o->lm_group; }

}  // namespace

extern "C" {

void ygz_offline_default_params(ygz_offline_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->width = 1280; p->height = 720; p->levels = 3; p->n_frames = 1024; p->rank = 0; p->world = 1; p->device = 0;
    p->chunk = 128; p->kf_stride = 8; p->window_kfs = 8; p->max_points = 2000; p->ba_iterations = 20;
    p->lanes = 3; p->lm_group = 0; p->obs_mode = 1; p->ba_rounds = 1; p->outlier_chi2 = 5.991;
    p->frame_channels = 3; p->depth_w = 320; p->depth_h = 180; p->depth_kind = 1; p->depth_scale = 1.0 / 5000.0;
    p->pipeline_ba = 1; p->defer_gaps = -1; p->ramp = 1; p->kf_tail = 1; p->stage_overlap = 0; p->bg_team_budget = 0; p->bg_team_spread = 1;
}

int ygz_offline_shard(int n_frames, int rank, int world, int *first, int *count, int *halo)
{
    if (n_frames < 0 || world < 1 || rank < 0 || rank >= world) return YGZ_E_INVALID;
    int s, c, h;
    shard_frames(n_frames, rank, world, s, c, h);
    if (first) *first = s;
    if (count) *count = c;
    if (halo) *halo = h;
    return YGZ_OK;
}

int ygz_offline_plan(const ygz_offline_params *p, int32_t *ranges, int capacity, int *n)
{
    if (!p || !n) return YGZ_E_INVALID;
    Plan P;
    const int rc = make_plan(*p, P);
    if (rc != YGZ_OK) return rc;
    int k = 0;
    for (size_t ci = 0; ci < P.chunks.size(); ++ci)
        for (const Range &r : P.chunks[ci]) {
            if (ranges && k < capacity) { ranges[3 * k] = (int32_t)ci; ranges[3 * k + 1] = r.first; ranges[3 * k + 2] = r.second; }
            ++k;
        }
    *n = k;
    return k > capacity && ranges ? YGZ_E_CAPACITY : YGZ_OK;
}

int ygz_offline_rccl_unique_id(void *id128)
{
    if (!id128) return YGZ_E_INVALID;
    if (!rccl().ok) return YGZ_E_STATE;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return YGZ_E_HIP;
    static_assert(sizeof(ncclUniqueId) == YGZ_OFFLINE_RCCL_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return YGZ_OK;
}

int ygz_offline_plan_range(int first, int last, int chunk, int ramp, int kf_stride, const int32_t *win_first_last, int n_windows, int defer,
                           int32_t *ranges, int capacity, int *n)
{
    if (!n || first < 0 || last < first || chunk < 1 || kf_stride < 0 || n_windows < 0 || (n_windows > 0 && !win_first_last)) return YGZ_E_INVALID;
    std::vector<std::vector<int>> wins;
    for (int i = 0; i < n_windows; ++i) wins.push_back(std::vector<int>{ win_first_last[2 * i], win_first_last[2 * i + 1] });
    const std::vector<ChunkRanges> plan = chunk_plan(first, last, chunk, ramp != 0, kf_stride, wins, defer);
    int k = 0;
    for (size_t ci = 0; ci < plan.size(); ++ci)
        for (const Range &r : plan[ci]) {
            if (ranges && k < capacity) { ranges[3 * k] = (int32_t)ci; ranges[3 * k + 1] = r.first; ranges[3 * k + 2] = r.second; }
            ++k;
        }
    *n = k;
    return k > capacity && ranges ? YGZ_E_CAPACITY : YGZ_OK;
}

int ygz_offline_ragged_all_gather(const ygz_offline_exchange *hook, int rank, int world, const int32_t *counts, size_t row_bytes, const void *local, void *full)
{
    if (!hook || !hook->all_gather || !counts || !full || world < 1 || rank < 0 || rank >= world || row_bytes == 0 || (counts[rank] > 0 && !local)) return YGZ_E_INVALID;
    std::vector<uint8_t> send, recv;
    return ragged_gather_host(*hook, rank, world, counts, row_bytes, local, full, send, recv);
}

const char *ygz_offline_last_error(const ygz_offline *o) { return o ? o->err.c_str() : "null handle"; }

void ygz_offline_destroy(ygz_offline *o)
{
    if (!o) return;
    for (ygz_hip_ctx *L : o->lanes) if (L) (void)ygz_hip_synchronize(L);
    if (o->ba) (void)ygz_hip_synchronize(o->ba);
    if (o->comm && rccl().ok) (void)rccl().CommDestroy(o->comm);
    if (o->ba) { (void)ygz_hip_device_free(o->ba, o->d_send); (void)ygz_hip_device_free(o->ba, o->d_recv); }
    (void)ygz_hip_pinned_free(o->h_send); (void)ygz_hip_pinned_free(o->h_recv);
    for (Chunk &c : o->chunks) { (void)ygz_hip_pinned_free(c.pin_sum); (void)ygz_hip_pinned_free(c.pin_cnt); }
    for (ygz_hip_ctx *L : o->lanes) if (L) ygz_hip_destroy(L);
    if (o->ba) ygz_hip_destroy(o->ba);
    delete o;
}

int ygz_offline_create(ygz_offline **out, const ygz_offline_params *p, const void *rccl_id, const ygz_offline_exchange *hook)
{
    if (!out || !p) return YGZ_E_INVALID;
    *out = nullptr;
    if (p->lanes < 1 || p->max_points < 1 || p->ba_iterations < 0 || (p->frame_channels != 1 && p->frame_channels != 3) || p->depth_w < 1 || p->depth_h < 1 ||
        p->depth_kind < 0 || p->depth_kind > 2 || (p->ba_rounds != 1 && p->ba_rounds != 2))
        return YGZ_E_INVALID;
    if (p->world > 1 && !rccl_id && !(hook && hook->all_gather && hook->send_recv)) return YGZ_E_INVALID;
    ygz_offline *o = new (std::nothrow) ygz_offline();
    if (!o) return YGZ_E_INVALID;
    o->p = *p;
    int rc = make_plan(*p, o->plan);
    if (rc != YGZ_OK) { delete o; return rc; }
    const Plan &P = o->plan;
    const int n_total = p->n_frames, last = P.start + P.count;
    // chunks: frames, slots, pairs, keyframes -- fixed for the life of the handle, so a run only issues ABI calls
    size_t n_slots = 2;
    o->chunks.resize(P.chunks.size());
    for (size_t ci = 0; ci < P.chunks.size(); ++ci) {
        Chunk &c = o->chunks[ci];
        c.ranges = P.chunks[ci];
        for (const Range &r : c.ranges) {
            const int f0 = r.first > 0 ? r.first - 1 : r.first;
            c.spans.push_back(std::make_pair((int)c.frames.size(), r.second - f0));
            for (int f = f0; f < r.second; ++f) c.frames.push_back(f);
        }
        auto slot_of = [&](int f) { for (size_t k = 0; k < c.frames.size(); ++k) if (c.frames[k] == f) return (int)k; return -1; };
        for (const Range &r : c.ranges)
            for (int f = r.first; f < r.second; ++f) {
                const int sr = slot_of(f - 1);
                if (sr >= 0) { c.pairs.push_back(f); c.pairs.push_back(f - 1); c.q.push_back(slot_of(f)); c.t.push_back(sr); }
                if (f % p->kf_stride == 0) { c.kf_slot.push_back(slot_of(f)); c.kf_row.push_back(f / p->kf_stride); }
            }
        for (size_t p0 = 0; p0 < c.q.size();) {
            size_t p1 = p0 + 1;
            while (p1 < c.q.size() && c.pairs[2 * p1] == c.pairs[2 * (p1 - 1)] + 1) ++p1;
            c.trel_runs.push_back(std::make_pair((int)p0, (int)(p1 - p0)));
            p0 = p1;
        }
        n_slots = std::max(n_slots, c.frames.size());
    }
    do {
        const int n_lanes = std::max(1, std::min(p->lanes, (int)o->chunks.size()));
        for (int i = 0; i < n_lanes; ++i) {
            ygz_hip_ctx *L = nullptr;
            if ((rc = create_ctx(o, (int)n_slots, &L)) != YGZ_OK) break;
            o->lanes.push_back(L);
            if ((rc = ygz_hip_set_overlap(L, p->stage_overlap)) != YGZ_OK) break;
        }
        if (rc != YGZ_OK) break;
        for (Chunk &c : o->chunks) {
            const size_t rows = std::max<size_t>(2, c.frames.size());
            void *a = nullptr, *b = nullptr;
            if ((rc = ygz_hip_pinned_alloc(&a, rows * YGZ_OFFLINE_SUMMARY_FIELDS * 8)) != YGZ_OK) break;
            c.pin_sum = (double *)a;
            if ((rc = ygz_hip_pinned_alloc(&b, rows * 4)) != YGZ_OK) break;
            c.pin_cnt = (int32_t *)b;
        }
        if (rc != YGZ_OK) break;
        // windows and their owners; one more context holds the keyframe store and the BA windows of this rank
        for (size_t wi = 0; wi < P.wins.size(); ++wi) {
            const int ow = frame_owner(P.wins[wi][0], n_total, p->world);
            o->owner.push_back(ow);
            if (frame_owner(P.wins[wi].back(), n_total, p->world) != ow) o->any_cross = true;
            if (ow == p->rank) { o->mine.push_back((int)wi); if (P.wins[wi].back() < last) o->local.push_back((int)wi); }
        }
        o->n_kf = (n_total + p->kf_stride - 1) / p->kf_stride;
        const int K1 = std::max(1, p->window_kfs - 1);
        o->build_group = std::max(1, std::min((int)o->mine.size(), 16));           // windows per build call (matcher rows: group x (K - 1) pairs)
        // A resident-LM launch takes as long for two windows as for eight (latency-bound) and the launches queue on one stream, so they only
        // hide behind the tracking of the following chunks if there are few of them: half of this rank's windows per launch (at most 8) on a
        // long shard, all of them in one launch on a short one
        o->lm_group = p->lm_group > 0 ? p->lm_group : (P.count > 256 ? std::min(8, (int)o->mine.size() / 2) : std::min(8, (int)o->mine.size()));
        o->lm_group = std::max(1, o->lm_group);
        if ((rc = create_ctx(o, std::max(8, o->build_group * K1), &o->ba)) != YGZ_OK) break;
        if ((rc = ygz_hip_kf_store_create(o->ba, o->n_kf, n_total, o->build_group, nullptr, 0, p->obs_mode ? 1 : 0)) != YGZ_OK) { o->err = "kf_store_create"; break; }
        if (!o->mine.empty() && (rc = ygz_hip_ba_reserve_windows(o->ba, 0, (int)o->mine.size(), p->window_kfs, p->max_points, 5.991)) != YGZ_OK) { o->err = "ba_reserve_windows"; break; }
        o->S = 6 * p->window_kfs + 3 * p->max_points + YGZ_OFFLINE_STATE_TAIL;
        o->ident.resize(7 * n_slots);
        for (size_t i = 0; i < n_slots; ++i) memcpy(&o->ident[7 * i], kI7, sizeof(kI7));
        // exchange buffers: the larger of [world][max frames per rank][7] and [world][max windows per rank][S] doubles
        if (p->world > 1 || rccl_id) {                  // (a communicator of ONE rank is allowed: the single-GPU test of the RCCL path)
            int mxf = 0, mxw = 1;
            std::vector<int> cnt(p->world, 0);
            for (int ow : o->owner) cnt[ow]++;
            for (int r = 0; r < p->world; ++r) { int s, c, h; shard_frames(n_total, r, p->world, s, c, h); mxf = std::max(mxf, c); mxw = std::max(mxw, cnt[r]); }
            const size_t per = std::max((size_t)mxf * 56, (size_t)mxw * o->S * 8);
            o->x_bytes = per * p->world;
            void *a = nullptr, *b = nullptr;
            if ((rc = ygz_hip_pinned_alloc(&a, per)) != YGZ_OK || (rc = ygz_hip_pinned_alloc(&b, o->x_bytes)) != YGZ_OK) { o->h_send = (uint8_t *)a; break; }
            o->h_send = (uint8_t *)a; o->h_recv = (uint8_t *)b;
            memset(o->h_send, 0, per);
            if (hook && hook->all_gather && hook->send_recv) { o->hook = *hook; o->backend = 2; }
            else {
                if (!rccl().ok) { o->err = "librccl.so.1 could not be loaded"; rc = YGZ_E_STATE; break; }
                if ((rc = ygz_hip_get_stream(o->ba, &o->ba_stream)) != YGZ_OK) break;
                if ((rc = ygz_hip_device_alloc(o->ba, &o->d_send, per)) != YGZ_OK || (rc = ygz_hip_device_alloc(o->ba, &o->d_recv, o->x_bytes)) != YGZ_OK) break;
                if ((rc = ygz_hip_make_current(o->ba)) != YGZ_OK) break;
                ncclUniqueId id;
                memcpy(&id, rccl_id, sizeof(id));
                const ncclResult_t r = rccl().CommInitRank(&o->comm, p->world, id, p->rank);
                if (r != ncclSuccess) { o->err = std::string("ncclCommInitRank: ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "error"); o->comm = nullptr; rc = YGZ_E_HIP; break; }
                o->backend = 1;
            }
        }
        o->T_rel.assign((size_t)n_total * 7, 0.0); o->traj.assign((size_t)n_total * 7, 0.0);
        o->summary.assign((size_t)n_total * YGZ_OFFLINE_SUMMARY_FIELDS, 0.0); o->n_kp.assign(n_total, 0);
        o->wstate.assign(P.wins.size() * (size_t)o->S, 0.0);
        o->wowner.assign(o->owner.begin(), o->owner.end());
        o->wkfs.assign(P.wins.size() * (size_t)p->window_kfs, -1);
        for (size_t wi = 0; wi < P.wins.size(); ++wi) for (size_t j = 0; j < P.wins[wi].size(); ++j) o->wkfs[wi * p->window_kfs + j] = P.wins[wi][j];
    } while (0);
    if (rc != YGZ_OK) {
        if (getenv("YGZ_OFFLINE_VERBOSE")) fprintf(stderr, "[ygz_offline] create failed: %s (%d)\n", o->err.c_str(), rc);
        ygz_offline_destroy(o);
        return rc;
    }
    *out = o;
    return YGZ_OK;
}

int ygz_offline_set_chunk_callback(ygz_offline *o, ygz_offline_chunk_fn fn, void *user)
{
    if (!o) return YGZ_E_INVALID;
    o->cb = fn; o->cb_user = user;
    return YGZ_OK;
}

int ygz_offline_contexts(ygz_offline *o, ygz_hip_ctx **ba, ygz_hip_ctx **lanes, int capacity, int *n_lanes)
{
    if (!o) return YGZ_E_INVALID;
    if (ba) *ba = o->ba;
    if (n_lanes) *n_lanes = (int)o->lanes.size();
    if (lanes) for (int i = 0; i < capacity && i < (int)o->lanes.size(); ++i) lanes[i] = o->lanes[i];
    return YGZ_OK;
}

int ygz_offline_owned_windows(const ygz_offline *o, int32_t *owned, int capacity, int *n)
{
    if (!o || !n) return YGZ_E_INVALID;
    *n = (int)o->mine.size();
    if (owned) for (int i = 0; i < capacity && i < *n; ++i) owned[i] = o->mine[i];
    return YGZ_OK;
}

int ygz_offline_build_windows(ygz_offline *o, int slot0, int n, int optimize)
{
    if (!o || slot0 < 0 || n < 0 || slot0 + n > (int)o->mine.size()) return YGZ_E_INVALID;
    if (o->ba_done.size() != o->mine.size()) o->ba_done.assign(o->mine.size(), 0);
    return ba_launch(o, slot0, n, optimize != 0);
}

int ygz_offline_retry_windows(ygz_offline *o, const int32_t *slots, int n)
{
    if (!o || n < 0 || (n > 0 && !slots)) return YGZ_E_INVALID;
    for (int i = 0; i < n; ++i) if (slots[i] < 0 || slots[i] >= (int)o->mine.size()) return YGZ_E_INVALID;
    if (o->ba_done.size() != o->mine.size()) o->ba_done.assign(o->mine.size(), 0);
    return retry_windows(o, slots, n);
}

// phase 1: the hot path over this shard, chunk by chunk on alternating lanes (+ the BA windows it completes).  The host hands a lane its next
// chunk when it has taken the results of the lane's previous one: `lanes` chunks are in flight (everything queued at once was measured
// slower: the resident-LM teams and the tracking kernels of all lanes then compete for the CUs)
int ygz_offline_track(ygz_offline *o, const uint8_t *frames, const void *depth, int first_in_buffer)
{
    if (!o || !frames || !depth) return YGZ_E_INVALID;
    const Plan &P = o->plan;
    if (first_in_buffer > P.start - P.halo) { o->err = "the frame buffer starts behind the shard's halo frame"; return YGZ_E_INVALID; }
    const double t0 = now_ms();
    o->trace = getenv("YGZ_OFFLINE_TRACE") != nullptr; o->t_run0 = t0;
    o->frames = frames; o->depth = (const uint8_t *)depth; o->first_in_buffer = first_in_buffer;
    o->tracked.assign(o->p.n_frames, 0); o->ba_done.assign(o->mine.size(), 0); o->ba_built.clear();
    o->lm_launches = 0; o->lm_retries = 0; o->n_degenerate = 0; o->last_upload = nullptr;
    std::fill(o->summary.begin(), o->summary.end(), 0.0); std::fill(o->n_kp.begin(), o->n_kp.end(), 0);
        std::vector<int> pending;
    size_t n_done = 0;
    const int n_chunks = (int)o->chunks.size();
    for (int ci = 0; ci < n_chunks; ++ci) {
        while (pending.size() >= o->lanes.size()) { const int rc = collect_chunk(o, pending.front()); if (rc != YGZ_OK) return rc; pending.erase(pending.begin()); }
        int rc = enqueue_chunk(o, ci);
        if (rc != YGZ_OK) return rc;
        if (o->cb) { if ((rc = collect_chunk(o, ci)) != YGZ_OK) return rc; }       // a caller that inspects the lane reads it before the lane moves on
        else pending.push_back(ci);
        if (!o->p.pipeline_ba) continue;
        // windows whose keyframes are all in: built at once (a few small kernels); the resident LM is launched per lm_group windows (and for
        // the rest after the last chunk), so that its launches fit beside the tracking of the following chunks instead of queueing up
        for (const Range &r : o->chunks[ci].ranges) for (int f = r.first; f < r.second; ++f) o->tracked[f] = 1;
        int s_lo = -1, s_n = 0;
        for (size_t k = 0; k < o->mine.size(); ++k) {
            const std::vector<int> &w = P.wins[o->mine[k]];
            if (w.back() >= P.start + P.count || o->ba_done[k] || std::find(o->ba_built.begin(), o->ba_built.end(), (int)k) != o->ba_built.end()) continue;
            bool all = true;
            for (int f = w[0]; f <= w.back() && all; ++f) all = o->tracked[f] == 1;
            if (!all) continue;
            if (s_lo < 0) s_lo = (int)k;
            if ((int)k != s_lo + s_n) { o->err = "windows become ready out of order"; return YGZ_E_STATE; }
            ++s_n;
        }
        if (s_n) { const double tb = now_ms(); if ((rc = ba_launch(o, s_lo, s_n, false)) != YGZ_OK) return rc; for (int k = s_lo; k < s_lo + s_n; ++k) o->ba_built.push_back(k); trace_line(o, "build", ci, tb); }
        const bool all_built = n_done + o->ba_built.size() == o->local.size();     // nothing more will come: the last launch need not wait for the last chunk
        if (!o->ba_built.empty() && ((int)o->ba_built.size() >= lm_next(o) || ci == n_chunks - 1 || all_built)) {
            // beside the tracking of the next chunks a launch may be held to a few CUs; the last launch, which nothing runs beside, takes the default
            // placement: a launch that chunks still follow must not own an XCD (every tracking kernel has workgroups there and would wait for
            // the whole LM: measured): spread over the XCDs, and held to bg_team_budget workgroups if that is set
            const bool company = ci < n_chunks - 1;
            OCHK(o, ygz_hip_ba_set_team_budget(o->ba, company ? o->p.bg_team_budget : 0), "ba_set_team_budget");
            OCHK(o, ygz_hip_ba_set_team_placement(o->ba, company && o->p.bg_team_spread), "ba_set_team_placement");
            const double tl = now_ms();
            if ((rc = lm(o, o->ba_built.front(), (int)o->ba_built.size())) != YGZ_OK) return rc;
            trace_line(o, "lm", ci, tl);
            for (int k : o->ba_built) o->ba_done[k] = 1;
            n_done += o->ba_built.size();
            o->ba_built.clear();
            o->lm_launches++;
        }
    }
    for (int ci : pending) { const int rc = collect_chunk(o, ci); if (rc != YGZ_OK) return rc; }
    o->ms_track = now_ms() - t0;
    return YGZ_OK;
}

// phase 2: all-gather of the per-shard relative poses -> the chained global trajectory, identical on every rank; keyframe rows of
// straddling windows to the windows' owners
int ygz_offline_gather(ygz_offline *o)
{
    if (!o) return YGZ_E_INVALID;
    const double t0 = now_ms();
    const Plan &P = o->plan;
    const int n = o->p.n_frames, W = o->p.world;
    auto local_row = [&](int f, double *dst) {
        const double *s = &o->summary[(size_t)f * YGZ_OFFLINE_SUMMARY_FIELDS];
        if (f == 0) memcpy(dst, kI7, 56); else memcpy(dst, s + 24, 56);            // [24..30]: the pose-only pose as (q, t)
    };
    if (o->backend == 0) { for (int f = 0; f < n; ++f) local_row(f, &o->T_rel[7 * (size_t)f]); }
    else {
        std::vector<int32_t> cn(W);
        for (int r = 0; r < W; ++r) { int st, c, h; shard_frames(n, r, W, st, c, h); cn[r] = c; }
        double *send = (double *)o->h_send;
        for (int k = 0; k < P.count; ++k) local_row(P.start + k, send + 7 * (size_t)k);
        const int rc = ragged_gather(o, cn, 56, false, o->T_rel.data());            // shards are contiguous: the concatenation IS the sequence
        if (rc != YGZ_OK) return rc;
    }
    memcpy(&o->T_rel[0], kI7, 56);
    OCHK(o, ygz_hip_se3_chain(o->T_rel.data(), n, o->traj.data()), "se3_chain");
    const int rc = exchange_rows(o);
    o->ms_gather = now_ms() - t0;
    return rc;
}

// phase 3: the windows this rank owns that are not optimised yet (those that straddle a shard boundary; all of them without pipeline_ba),
// then the map exchange: every owner's refined window states to every rank in one all-gather
int ygz_offline_ba_round(ygz_offline *o)
{
    if (!o) return YGZ_E_INVALID;
    const double t0 = now_ms();
    const Plan &P = o->plan;
    const int W = o->p.world, S = o->S, n_w = (int)P.wins.size();
    if (o->ba_done.size() != o->mine.size()) o->ba_done.assign(o->mine.size(), 0);
    int r0 = -1, rn = 0;
    for (size_t k = 0; k < o->mine.size(); ++k) if (!o->ba_done[k]) { if (r0 < 0) r0 = (int)k; ++rn; }
    if (rn) {
        if (r0 + rn != (int)o->mine.size()) { o->err = "the windows left for the BA round are not the last ones"; return YGZ_E_STATE; }
        if (W > 1) OCHK(o, ygz_hip_kf_store_set_trel(o->ba, 0, o->p.n_frames, o->T_rel.data()), "kf_store_set_trel");   // relative poses of the frames other ranks tracked
        OCHK(o, ygz_hip_ba_set_team_budget(o->ba, 0), "ba_set_team_budget");
        OCHK(o, ygz_hip_ba_set_team_placement(o->ba, 0), "ba_set_team_placement");
        const int rc = ba_launch(o, r0, rn, true);
        if (rc != YGZ_OK) return rc;
    }
    OCHK(o, ygz_hip_synchronize(o->ba), "synchronize");
    { const int rc = retry_timed_out(o); if (rc != YGZ_OK) return rc; }
    const double t1 = now_ms();
    const int n_mine = (int)o->mine.size();
    if (o->backend == 0) {
        if (n_w) OCHK(o, ygz_hip_ba_pack_states(o->ba, 0, n_w, o->wstate.data(), (size_t)S, 0, 1), "ba_pack_states");
    } else {
        std::vector<int32_t> cnt(W, 0);
        for (int wi = 0; wi < n_w; ++wi) cnt[o->owner[wi]]++;                       // owners hold contiguous runs of windows (ordered by anchor frame)
        const bool dev = o->backend == 1;
        if (n_mine) OCHK(o, ygz_hip_ba_pack_states(o->ba, 0, n_mine, dev ? (double *)o->d_send : (double *)o->h_send, (size_t)S, dev ? 1 : 0, dev ? 0 : 1), "ba_pack_states");
        const int rc = ragged_gather(o, cnt, (size_t)S * 8, dev, o->wstate.data());
        if (rc != YGZ_OK) return rc;
    }
    o->n_degenerate = 0;
    for (int wi = 0; wi < n_w; ++wi) {
        const double *tail = &o->wstate[(size_t)wi * S + 6 * (size_t)o->p.window_kfs + 3 * (size_t)o->p.max_points];
        if (tail[3] < 0) { o->err = "BA window " + std::to_string(wi) + ": the resident LM did not finish (team barrier time-out)"; return YGZ_E_HIP; }
        if (tail[1] == 0 || tail[2] == 0) o->n_degenerate++;          // no map point of the anchor was observed in another keyframe: not optimised
    }
    const double t2 = now_ms();
    o->ms_ba_tail = t1 - t0; o->ms_exchange = t2 - t1;
    return YGZ_OK;
}

int ygz_offline_run(ygz_offline *o, const uint8_t *frames, const void *depth, int first_in_buffer)
{
    int rc = ygz_offline_track(o, frames, depth, first_in_buffer);
    if (rc == YGZ_OK) rc = ygz_offline_gather(o);
    if (rc == YGZ_OK) rc = ygz_offline_ba_round(o);
    return rc;
}

int ygz_offline_get_results(const ygz_offline *o, ygz_offline_results *r)
{
    if (!o || !r) return YGZ_E_INVALID;
    r->n_frames = o->p.n_frames; r->first_frame = o->plan.start; r->n_own = o->plan.count;
    r->T_rel = o->T_rel.data(); r->trajectory = o->traj.data(); r->summary = o->summary.data(); r->n_kp = o->n_kp.data();
    r->n_windows = (int)o->plan.wins.size(); r->state_doubles = o->S;
    r->window_state = o->wstate.data(); r->window_owner = o->wowner.data(); r->window_kfs = o->wkfs.data();
    r->n_chunks = (int)o->chunks.size(); r->lm_launches = o->lm_launches; r->lm_retries = o->lm_retries; r->n_degenerate = o->n_degenerate;
    r->ms_track = o->ms_track; r->ms_gather = o->ms_gather; r->ms_ba_tail = o->ms_ba_tail; r->ms_exchange = o->ms_exchange;
    r->backend = o->backend;
    return YGZ_OK;
}

}  // extern "C"
