/* ORACLE (test infrastructure) -- the BoW-guided matching rows (8f-2).
 *
 * Restates, from sources that ARE in the reference tree:
 *   thirdparty/DBoW3/src/Vocabulary.cpp:790-835   Vocabulary::transform(feature, word, weight, nid, levelsup): tree descent,
 *                                                  first child with the strictly smallest Hamming distance wins
 *   thirdparty/DBoW3/src/Vocabulary.cpp:706-774   transform(features, BowVector, FeatureVector, levelsup) for TF_IDF / TF
 *                                                  weighting (what an ORB vocabulary file carries) + BowVector::normalize (L1)
 *   thirdparty/DBoW3/src/Vocabulary.cpp (loadFromBinaryFile)  node record = int parent, 32 bytes, float weight, byte is_leaf
 *   src/Basic/Frame.cpp:190-201                    Frame::ComputeBoW (levelsup = 4)
 *   src/Algorithm/Matcher.cpp:196-292              Matcher::SearchByBoW
 *   src/Algorithm/Matcher.cpp:86-193,338-354       Matcher::SearchForTriangulation + CheckDistEpipolarLine (float arithmetic)
 * The vocabulary blob itself (vocab/ORBvoc.bin) is not shipped with the reference; tests build synthetic vocabularies in the
 * same binary format.  The reference's loader reads one record past the end (while(!f.eof())): here exactly nb_nodes.
 * See ygz_oracle.h for the rules. */
#include "ygz_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int yo_vocab_parse(const void *blob, size_t bytes, yo_vocab *v)
{
    memset(v, 0, sizeof(*v));
    if (bytes < 24) return -1;
    const uint8_t *p = (const uint8_t *)blob;
    uint32_t nb_nodes, size_node;
    memcpy(&nb_nodes, p, 4); memcpy(&size_node, p + 4, 4);
    memcpy(&v->k, p + 8, 4); memcpy(&v->L, p + 12, 4); memcpy(&v->scoring, p + 16, 4); memcpy(&v->weighting, p + 20, 4);
    if (size_node < 41 || bytes < 24 + (size_t)nb_nodes * size_node) return -1;
    const int n = (int)nb_nodes + 1;
    v->n_nodes = n;
    v->parent = (int32_t *)calloc((size_t)n, 4); v->desc = (uint8_t *)calloc((size_t)n, 32); v->weight = (double *)calloc((size_t)n, 8);
    v->word_id = (int32_t *)malloc((size_t)n * 4); v->child_off = (int32_t *)calloc((size_t)n + 1, 4); v->child = (int32_t *)malloc((size_t)n * 4);
    int32_t *cnt = (int32_t *)calloc((size_t)n, 4);
    int words = 0;
    for (int i = 0; i < n; ++i) v->word_id[i] = -1;
    for (int nid = 1; nid < n; ++nid) {
        const uint8_t *rec = p + 24 + (size_t)(nid - 1) * size_node;
        int32_t par; float w;
        memcpy(&par, rec, 4); memcpy(v->desc + 32 * (size_t)nid, rec + 4, 32); memcpy(&w, rec + 36, 4);
        if (par < 0 || par >= nid) { free(cnt); yo_vocab_free(v); return -1; }     /* parents precede their children */
        v->parent[nid] = par; v->weight[nid] = (double)w;                           /* WordValue is double */
        if (rec[40]) v->word_id[nid] = words++;
        cnt[par]++;
    }
    v->n_words = words;
    for (int i = 0; i < n; ++i) v->child_off[i + 1] = v->child_off[i] + cnt[i];
    memset(cnt, 0, (size_t)n * 4);
    for (int nid = 1; nid < n; ++nid) { const int par = v->parent[nid]; v->child[v->child_off[par] + cnt[par]++] = nid; }   /* push_back order */
    free(cnt);
    return 0;
}

void yo_vocab_free(yo_vocab *v)
{
    free(v->parent); free(v->desc); free(v->weight); free(v->word_id); free(v->child_off); free(v->child);
    memset(v, 0, sizeof(*v));
}

/* Vocabulary::transform for one feature (Vocabulary.cpp:790-835); a node is a leaf iff it has no children */
void yo_bow_transform_one(const yo_vocab *v, const uint8_t *d, int levelsup, int32_t *word, double *weight, int32_t *nid)
{
    const int nid_level = v->L - levelsup;
    *nid = 0;                                                   /* root when nid_level <= 0 */
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const int c0 = v->child_off[final_id], c1 = v->child_off[final_id + 1];
        if (c0 == c1) break;                                    /* (an empty vocabulary) */
        final_id = v->child[c0];
        double best_d = (double)yo_descriptor_distance(d, v->desc + 32 * (size_t)final_id);
        for (int c = c0 + 1; c < c1; ++c) {
            const int id = v->child[c];
            const double dd = (double)yo_descriptor_distance(d, v->desc + 32 * (size_t)id);
            if (dd < best_d) { best_d = dd; final_id = id; }
        }
        if (current_level == nid_level) *nid = final_id;
    } while (v->child_off[final_id] != v->child_off[final_id + 1]);
    *word = v->word_id[final_id];
    *weight = v->weight[final_id];
}

/* transform(features, BowVector, FeatureVector, levelsup), TF_IDF / TF weighting with L1 normalisation.
 * Per feature: word, weight, node (node = -1 when the word is stopped, i.e. the feature is not in the FeatureVector).
 * BowVector as parallel arrays sorted by word id: bow_word / bow_value [<= n]; returns its size. */
int yo_bow_transform(const yo_vocab *v, const uint8_t *desc, int n, int levelsup, int32_t *word, double *weight, int32_t *node,
                     int32_t *bow_word, double *bow_value)
{
    double *acc = (double *)calloc((size_t)(v->n_words > 0 ? v->n_words : 1), 8);
    uint8_t *has = (uint8_t *)calloc((size_t)(v->n_words > 0 ? v->n_words : 1), 1);
    for (int i = 0; i < n; ++i) {
        int32_t nid;
        yo_bow_transform_one(v, desc + 32 * (size_t)i, levelsup, &word[i], &weight[i], &nid);
        if (weight[i] > 0 && word[i] >= 0) { acc[word[i]] += weight[i]; has[word[i]] = 1; node[i] = nid; }     /* addWeight / addFeature */
        else node[i] = -1;
    }
    int m = 0;
    double norm = 0.0;
    for (int w = 0; w < v->n_words; ++w) if (has[w]) { bow_word[m] = w; bow_value[m] = acc[w]; norm += fabs(acc[w]); ++m; }
    if (norm > 0.0) for (int i = 0; i < m; ++i) bow_value[i] /= norm;                                           /* BowVector::normalize(L1) */
    free(acc); free(has);
    return m;
}

/* Matcher::SearchByBoW (Matcher.cpp:196-292) without the checkOrientation part (yo_bow_orientation below: it only changes the count).
 * node1/node2: FeatureVector membership (-1 = not in it).  match12 [n1] = index in frame 2 or -1.  Returns cnt_matches. */
int yo_search_by_bow(const uint8_t *desc1, const int32_t *node1, int n1, const uint8_t *desc2, const int32_t *node2, int n2,
                     int th_low, float knn_ratio, int32_t *match12)
{
    int cnt = 0;
    for (int i = 0; i < n1; ++i) {
        match12[i] = -1;
        if (node1[i] < 0) continue;
        int bestDist1 = 256, bestIdxF2 = -1, bestDist2 = 256, any = 0;
        for (int j = 0; j < n2; ++j) {                              /* indices_f2 is in ascending feature order */
            if (node2[j] != node1[i]) continue;
            any = 1;
            const int dist = yo_descriptor_distance(desc1 + 32 * (size_t)i, desc2 + 32 * (size_t)j);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF2 = j; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (!any) continue;                                         /* the node is absent from frame 2's FeatureVector */
        if (bestDist1 < th_low && (float)bestDist1 < knn_ratio * (float)bestDist2) { match12[i] = bestIdxF2; ++cnt; }
    }
    return cnt;
}

/* Matcher::ComputeThreeMaxima (Matcher.cpp:293-336): the three fullest bins (-1: fewer than a tenth of the fullest) */
static void compute_three_maxima(const int32_t *hist, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; ++i) {
        const int s = hist[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* The checkOrientation part of Matcher::SearchByBoW (Matcher.cpp:247-256, 271-289; HISTO_LENGTH = 30, Matcher.h:66) on its result
 * match12: the histogram of rot = angle1 - angle2 (float; + 360 when negative) over bin = round(rot * (1.0f / 30)) -- the factor as the
 * reference has it, so only bins 0 .. 12 fill --, the three maxima, and cnt_matches minus the entries of every other bin.  The
 * reference does NOT remove those matches from the map (its TODO at :284); only the returned count changes.  hist [30], ind [3].
 * (SearchForTriangulation fills the same histogram and never reads it, :157-165, 180-182: nothing to restate.) */
int yo_bow_orientation(const double *angle1, const double *angle2, const int32_t *match12, int n1, int32_t *hist, int32_t *ind)
{
    const float factor = 1.0f / 30;
    int cnt = 0;
    for (int b = 0; b < 30; ++b) hist[b] = 0;
    for (int i = 0; i < n1; ++i) {
        if (match12[i] < 0) continue;
        float rot = (float)(angle1[i] - angle2[match12[i]]);
        if (rot < 0) rot += 360;
        int bin = (int)round(rot * factor);
        if (bin == 30) bin = 0;
        if (bin < 0 || bin >= 30) continue;                          /* the reference asserts */
        hist[bin]++; ++cnt;
    }
    int i1 = -1, i2 = -1, i3 = -1;
    compute_three_maxima(hist, 30, &i1, &i2, &i3);
    ind[0] = i1; ind[1] = i2; ind[2] = i3;
    for (int b = 0; b < 30; ++b) if (b != i1 && b != i2 && b != i3) cnt -= hist[b];
    return cnt;
}

/* Matcher::CheckDistEpipolarLine (Matcher.cpp:338-354): float arithmetic as written */
static int epipolar_ok(const double pt1[3], const double pt2[3], const double E[9], double dsqr_thr)
{
    const float a = (float)(pt1[0] * E[0] + pt1[1] * E[3] + E[6]);
    const float b = (float)(pt1[0] * E[1] + pt1[1] * E[4] + E[7]);
    const float c = (float)(pt1[0] * E[2] + pt1[1] * E[5] + E[8]);
    const float num = (float)((double)a * pt2[0] + (double)b * pt2[1] + (double)c);
    const float den = a * a + b * b;
    if ((double)den < 1e-6) return 0;
    const float dsqr = num * num / den;
    return (double)fabsf(dsqr) < dsqr_thr;
}

/* Matcher::SearchForTriangulation (Matcher.cpp:86-193): best = the LAST candidate whose distance is <= th_low and <= the best
 * so far and that satisfies the epipolar constraint (`dist > bestDist` rejects, equality replaces).  E12 row-major. */
int yo_search_for_triangulation(const yo_camera *cam, const uint8_t *desc1, const int32_t *node1, const double *px1, int n1,
                                const uint8_t *desc2, const int32_t *node2, const double *px2, int n2,
                                const double E12[9], int th_low, double epipolar_dsqr, int32_t *match12)
{
    int matches = 0;
    const double fx = (double)cam->fx, fy = (double)cam->fy, cx = (double)cam->cx, cy = (double)cam->cy;
    for (int i = 0; i < n1; ++i) {
        match12[i] = -1;
        if (node1[i] < 0) continue;
        const double pt1[3] = { (px1[2 * i] - cx) * 1.0 / fx, (px1[2 * i + 1] - cy) * 1.0 / fy, 1.0 };      /* Pixel2Camera(p, depth = 1) */
        int bestDist = 256, bestIdx2 = -1;
        for (int j = 0; j < n2; ++j) {
            if (node2[j] != node1[i]) continue;
            const int dist = yo_descriptor_distance(desc1 + 32 * (size_t)i, desc2 + 32 * (size_t)j);
            if (dist > th_low || dist > bestDist) continue;
            const double pt2[3] = { (px2[2 * j] - cx) * 1.0 / fx, (px2[2 * j + 1] - cy) * 1.0 / fy, 1.0 };
            if (epipolar_ok(pt1, pt2, E12, epipolar_dsqr)) { bestIdx2 = j; bestDist = dist; }
        }
        if (bestIdx2 >= 0) { match12[i] = bestIdx2; ++matches; }
    }
    return matches;
}
