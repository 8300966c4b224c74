/* ORACLE (test infrastructure) -- 256-bit Hamming matcher.  See ygz_oracle.h. */
#include "ygz_oracle.h"
#include <limits.h>
#include <stdlib.h>
#include <string.h>

/* Matcher::DescriptorDistance -- src/Algorithm/Matcher.cpp:30-43 (SWAR popcount) */
int yo_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t wa, wb;
        memcpy(&wa, a + 4 * i, 4); memcpy(&wb, b + 4 * i, 4);
        uint32_t v = wa ^ wb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

void yo_hamming_nn(const uint8_t *q, int nq, const uint8_t *t, int nt,
                   int32_t *idx, int32_t *dist, int32_t *dist2)
{
    for (int i = 0; i < nq; ++i) {
        int best = INT_MAX, second = INT_MAX, bi = -1;
        for (int j = 0; j < nt; ++j) {
            const int d = yo_descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < best) { second = best; best = d; bi = j; }
            else if (d < second) second = d;
        }
        idx[i] = bi; dist[i] = best;
        if (dist2) dist2[i] = second;
    }
}

/* cv::BFMatcher(NORM_HAMMING, crossCheck=true).match(desp1, desp2) -- test/test_orb_match.cpp:86-93.
 * [frozen spec] OpenCV batchDistance(..., crosscheck=true): first the nearest QUERY of
 * every TRAIN row is found (first minimum); then, visiting train rows in order, query
 * idx keeps train row i iff d(i) < its current best (strict).  A query that no train
 * row voted for has no match.  DMatches come out in query order. */
int yo_bf_match(const uint8_t *q, int nq, const uint8_t *t, int nt, int cross_check,
                int32_t *train_idx, int32_t *dist)
{
    int n = 0;
    if (cross_check == 0) {
        yo_hamming_nn(q, nq, t, nt, train_idx, dist, NULL);
        for (int i = 0; i < nq; ++i) n += train_idx[i] >= 0;
        return n;
    }
    int32_t *tq = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nt + 1));
    int32_t *td = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nt + 1));
    yo_hamming_nn(t, nt, q, nq, tq, td, NULL);
    for (int i = 0; i < nq; ++i) { train_idx[i] = -1; dist[i] = INT_MAX; }
    if (cross_check == 1) {
        for (int j = 0; j < nt; ++j) {
            const int i = tq[j];
            if (i >= 0 && td[j] < dist[i]) { dist[i] = td[j]; train_idx[i] = j; }
        }
    } else {
        int32_t *qi = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nq + 1));
        int32_t *qd = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nq + 1));
        yo_hamming_nn(q, nq, t, nt, qi, qd, NULL);
        for (int i = 0; i < nq; ++i)
            if (qi[i] >= 0 && tq[qi[i]] == i) { train_idx[i] = qi[i]; dist[i] = qd[i]; }
        free(qi); free(qd);
    }
    for (int i = 0; i < nq; ++i) n += train_idx[i] >= 0;
    free(tq); free(td);
    return n;
}

/* test/test_orb_match.cpp:97-104 */
int yo_good_match_filter(const int32_t *train_idx, const int32_t *dist, int nq, uint8_t *keep)
{
    double min_dis = 1e300;
    for (int i = 0; i < nq; ++i) if (train_idx[i] >= 0 && dist[i] < min_dis) min_dis = dist[i];
    min_dis = min_dis < 20 ? 20 : min_dis;
    min_dis = min_dis > 50 ? 50 : min_dis;
    int n = 0;
    for (int i = 0; i < nq; ++i) {
        keep[i] = (uint8_t)(train_idx[i] >= 0 && dist[i] < 3 * min_dis);
        n += keep[i];
    }
    return n;
}

/* Matcher::CheckFrameDescriptors -- src/Algorithm/Matcher.cpp:45-84.  Distances of the (idx1[i], idx2[i]) feature pairs,
 * best clamped to [init_low, init_high] (Matcher.h:27-28: 30, 80; the YAML values are read at Matcher.cpp:15-16), keep[i] =
 * distance[i] < initMatchRatio * best_dist (float x int -> float, :72).  Returns cnt_good; *best_out = the clamped best.
 * n == 0 is undefined in the reference (min_element of an empty vector is dereferenced): defined here as 0 / init_low. */
int yo_check_frame_descriptors(const uint8_t *desc1, const uint8_t *desc2, const int32_t *idx1, const int32_t *idx2, int n,
                               int init_low, int init_high, float ratio, int32_t *dist, uint8_t *keep, int *best_out)
{
    if (n <= 0) { if (best_out) *best_out = init_low; return 0; }
    int best_dist = INT_MAX;
    for (int i = 0; i < n; ++i) {
        dist[i] = yo_descriptor_distance(desc1 + 32 * (size_t)idx1[i], desc2 + 32 * (size_t)idx2[i]);
        if (dist[i] < best_dist) best_dist = dist[i];
    }
    best_dist = best_dist > init_low ? best_dist : init_low;
    best_dist = best_dist < init_high ? best_dist : init_high;
    int cnt_good = 0;
    for (int i = 0; i < n; ++i) {
        keep[i] = (uint8_t)((float)dist[i] < ratio * (float)best_dist);
        cnt_good += keep[i];
    }
    if (best_out) *best_out = best_dist;
    return cnt_good;
}
