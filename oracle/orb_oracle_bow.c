/* CPU oracle (test infrastructure only, never the product path; parity unpinned: the reference holds no fixtures for
 * these routines): the two Hamming consumers that follow extraction in the reference.
 *   - MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:483-548
 *   - DBoW2 TemplatedVocabulary::transform, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1139-1272, with
 *     FORB::distance (FORB.cpp:81-101) and BowVector (BowVector.cpp:34-84), as Frame::ComputeBoW calls it
 *     (src/Frame.cc:395-402, levelsup = 4). */
#include "orb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

static int cmp_int(const void* a, const void* b)
{
    const int x = *(const int*)a, y = *(const int*)b;
    return (x > y) - (x < y);
}

/* src/MapPoint.cc:513-547 */
int orc_distinctive_descriptor(const uint8_t* desc, int n)
{
    if (n <= 0) return -1;
    float* D = (float*)malloc(sizeof(float) * (size_t)n * n);           /* float Distances[N][N], :516 */
    int* row = (int*)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        D[(size_t)i * n + i] = 0;
        for (int j = i + 1; j < n; ++j) {
            const int dij = orc_descriptor_distance(desc + (size_t)i * 32, desc + (size_t)j * 32);
            D[(size_t)i * n + j] = (float)dij;
            D[(size_t)j * n + i] = (float)dij;
        }
    }
    int best_median = INT_MAX, best_idx = 0;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) row[j] = (int)D[(size_t)i * n + j];  /* vector<int> vDists(Distances[i], ...), :533 */
        qsort(row, (size_t)n, sizeof(int), cmp_int);
        const int median = row[(size_t)(0.5 * (n - 1))];                 /* :535 */
        if (median < best_median) { best_median = median; best_idx = i; }
    }
    free(D); free(row);
    return best_idx;
}

/* children lists in node order, as loadFromTextFile builds them (TemplatedVocabulary.h:1398-1405) */
typedef struct { int32_t* first; int32_t* list; int32_t* word_of; int32_t n_words; } tree_index;

static int build_tree(const orc_vocabulary* v, tree_index* t)
{
    const int n = v->n_nodes;
    t->first = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
    t->list = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 1 ? n - 1 : 1));
    t->word_of = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    t->n_words = 0;
    for (int i = 1; i < n; ++i) {
        if (v->parent[i] < 0 || v->parent[i] >= i) return -1;
        t->first[v->parent[i] + 1]++;
    }
    for (int i = 0; i < n; ++i) t->first[i + 1] += t->first[i];
    int32_t* fill = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    memcpy(fill, t->first, sizeof(int32_t) * (size_t)n);
    for (int i = 1; i < n; ++i) t->list[fill[v->parent[i]]++] = i;
    free(fill);
    for (int i = 0; i < n; ++i) {
        const int leaf = i > 0 && v->is_leaf[i];
        t->word_of[i] = leaf ? t->n_words++ : -1;                         /* :1421-1428 */
        if (i > 0 && leaf != (t->first[i + 1] == t->first[i])) return -1;
    }
    if (n < 2) return -1;
    return 0;
}

static void free_tree(tree_index* t) { free(t->first); free(t->list); free(t->word_of); }

int orc_vocabulary_transform(const orc_vocabulary* voc, const uint8_t* desc, int n, int levelsup,
                             int32_t* word, int32_t* node,
                             uint32_t* bow_ids, double* bow_values, int32_t* n_words,
                             uint32_t* fv_node_ids, int32_t* fv_node_ptr, int32_t* fv_features, int32_t* fv_n_nodes)
{
    tree_index t;
    if (build_tree(voc, &t)) { free_tree(&t); return -1; }
    const int nid_level = voc->L - levelsup;                                         /* :1239 */
    int32_t* leaf_of = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        const uint8_t* f = desc + (size_t)i * 32;
        int final_id = 0, current_level = 0, nid = -1;
        if (nid_level <= 0) nid = 0;                                                 /* :1240 */
        do {
            ++current_level;
            const int32_t* ch = t.list + t.first[final_id];
            const int nch = t.first[final_id + 1] - t.first[final_id];
            final_id = ch[0];
            double best_d = orc_descriptor_distance(f, voc->descriptors + (size_t)final_id * 32);
            for (int c = 1; c < nch; ++c) {
                const double d = orc_descriptor_distance(f, voc->descriptors + (size_t)ch[c] * 32);
                if (d < best_d) { best_d = d; final_id = ch[c]; }                    /* :1257-1261 */
            }
            if (current_level == nid_level) nid = final_id;                          /* :1264-1265 */
        } while (t.first[final_id + 1] != t.first[final_id]);
        word[i] = t.word_of[final_id];
        leaf_of[i] = final_id;
        node[i] = nid >= 0 ? nid : final_id;
    }
    /* BowVector / FeatureVector: std::map semantics written out as insertion into sorted arrays */
    int nw = 0, nn = 0;
    int32_t* fv_count = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
    uint32_t* fnode = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    const int tf = voc->weighting == 0 || voc->weighting == 1;                       /* TF_IDF, TF */
    for (int i = 0; i < n; ++i) {
        const double w = voc->weight[leaf_of[i]];                                    /* :1271 */
        if (!(w > 0)) continue;                                                      /* :1170, stopped words */
        int pos = 0;
        while (pos < nw && bow_ids[pos] < (uint32_t)word[i]) ++pos;                  /* lower_bound */
        if (pos < nw && bow_ids[pos] == (uint32_t)word[i]) {
            if (tf) bow_values[pos] += w;                                            /* addWeight / addIfNotExist */
        } else {
            memmove(bow_ids + pos + 1, bow_ids + pos, sizeof(uint32_t) * (size_t)(nw - pos));
            memmove(bow_values + pos + 1, bow_values + pos, sizeof(double) * (size_t)(nw - pos));
            bow_ids[pos] = (uint32_t)word[i]; bow_values[pos] = w; ++nw;
        }
        pos = 0;
        while (pos < nn && fv_node_ids[pos] < (uint32_t)node[i]) ++pos;
        if (!(pos < nn && fv_node_ids[pos] == (uint32_t)node[i])) {
            memmove(fv_node_ids + pos + 1, fv_node_ids + pos, sizeof(uint32_t) * (size_t)(nn - pos));
            fv_node_ids[pos] = (uint32_t)node[i]; ++nn;
        }
        fnode[i] = (uint32_t)node[i];
        fv_count[i] = 1;
    }
    /* features of each node in feature order (FeatureVector::addFeature appends) */
    int filled = 0;
    for (int k = 0; k < nn; ++k) {
        fv_node_ptr[k] = filled;
        for (int i = 0; i < n; ++i) if (fv_count[i] && fnode[i] == fv_node_ids[k]) fv_features[filled++] = i;
    }
    fv_node_ptr[nn] = filled;
    const int must = voc->scoring != 5;                                              /* all but DOT_PRODUCT, ScoringObject.h:74-89 */
    if (tf && nw > 0 && !must) {
        const double nd = nw;                                                        /* :1177-1183 */
        for (int k = 0; k < nw; ++k) bow_values[k] /= nd;
    }
    if (must) {                                                                      /* BowVector::normalize */
        double norm = 0.0;
        if (voc->scoring != 1) { for (int k = 0; k < nw; ++k) norm += fabs(bow_values[k]); }
        else { for (int k = 0; k < nw; ++k) norm += bow_values[k] * bow_values[k]; norm = sqrt(norm); }
        if (norm > 0.0) for (int k = 0; k < nw; ++k) bow_values[k] /= norm;
    }
    *n_words = nw; *fv_n_nodes = nn;
    free(fv_count); free(fnode); free(leaf_of); free_tree(&t);
    return 0;
}
