/*
 * orb_oracle.h -- CPU oracle for the ORB front-end hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A dependency-free C restatement of the reference's ORBextractor::operator(),
 * ORBmatcher descriptor matching and the Frame helpers those use, written from
 *   /root/reference/src/ORBextractor.cc, src/ORBmatcher.cc, src/Frame.cc
 * plus the OpenCV 2.4.9-3.4.0 generic (non-IPP, non-OpenCL) semantics of the
 * primitives the reference calls (resize INTER_LINEAR, copyMakeBorder
 * REFLECT_101, FAST 9_16 with NMS, GaussianBlur 7x7 sigma 2, fastAtan2, cvRound).
 *
 * PARITY UNPINNED at the OpenCV boundary: the reference ships no tests, golden
 * vectors or fixtures for this path and OpenCV is neither vendored in the
 * reference nor installed in this image, so the primitives are pinned only by
 * hand-derived known-answer values (quotas, umax, level sizes, scale tables,
 * Gaussian taps, hand-computed FAST scores).  The oracle is the checker for the
 * HIP library; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  The product library never links or calls it.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_params {
    int32_t nfeatures;
    float   scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t blur_rounding; /* 0 = (sum+32768)>>16 everywhere; 1 = x86 SIMD column pass (ties to even for x < w&~3) */
    int32_t trig_mode;     /* cos/sin in computeOrbDescriptor: 0 = (float)cos((double)t); 1 = restatement of glibc cosf/sinf as
                              its x86-64 FMA build evaluates it; 2 = the same without fused multiply-adds; 3 = this host's libm
                              cosf/sinf called directly (the function src/ORBextractor.cc:112 calls) */
} orc_params;

/* cos and sin of theta under trig_mode (see orc_params) */
void orc_trig(int trig_mode, float theta, float* cos_out, float* sin_out);
/* the same for angle * (float)(pi/180), angle = the floats with bit patterns first_bits .. first_bits + count - 1 */
void orc_trig_range(int trig_mode, uint32_t first_bits, uint32_t count, float* cos_out, float* sin_out);
/* number of those angles for which modes a and b differ in cos or sin (bitwise) */
uint64_t orc_trig_count_differences(int mode_a, int mode_b, uint32_t first_bits, uint32_t count);

/* Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct orc_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

typedef struct orc_best2 { int32_t best, best_idx, second, pad; } orc_best2;

typedef struct orc_extractor orc_extractor;

/* ---- primitives (OpenCV semantics) ---- */
int   orc_cv_round(double v);
float orc_fast_atan2(float y, float x);
void orc_fast_atan2_array(const float* y, const float* x, int n, float* out);
void  orc_resize_linear_u8(const uint8_t* src, int sw, int sh, ptrdiff_t sstride,
                           uint8_t* dst, int dw, int dh, ptrdiff_t dstride);
/* fills the `pad`-pixel frame around the w x h ROI at roi (buffer stride `stride`) by REFLECT_101 of the ROI */
void  orc_border101(uint8_t* roi, int w, int h, ptrdiff_t stride, int pad);
/* FAST-9_16 on a w x h sub-image; returns count; out arrays (x,y cell-local, score) sized cap */
int   orc_fast9(const uint8_t* img, int w, int h, ptrdiff_t stride, int threshold, int nonmax,
                int16_t* ox, int16_t* oy, uint8_t* oscore, int cap);
int   orc_fast9_score(const uint8_t* center, ptrdiff_t stride, int threshold);
void  orc_gauss7(const uint8_t* src, int w, int h, ptrdiff_t sstride, uint8_t* dst, ptrdiff_t dstride, int rounding);
void  orc_gauss_taps(int32_t taps[7]);

/* ---- extractor ---- */
orc_extractor* orc_extractor_create(const orc_params* p);
void  orc_extractor_destroy(orc_extractor* ex);
int   orc_extractor_tables(const orc_extractor* ex, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                           int32_t* features_per_level, int32_t* umax16);
/* returns 0 ok; 2 = geometry the reference cannot process (divide-by-zero cases); 4 = cap too small */
int   orc_extract(orc_extractor* ex, const uint8_t* img, int w, int h, ptrdiff_t stride,
                  orc_keypoint* kps, uint8_t* desc, int cap, int* n);
/* observable intermediates of the last orc_extract call */
int   orc_level_size(const orc_extractor* ex, int level, int* w, int* h);
/* pointer to the padded buffer of a level: (w+38) x (h+38), stride w+38; ROI starts at +19,+19 */
const uint8_t* orc_level_padded(const orc_extractor* ex, int level, int* pw, int* ph);
const uint8_t* orc_level_blurred(const orc_extractor* ex, int level); /* w x h, stride w; NULL if level had no keypoints */
/* candidates handed to DistributeOctTree (coords relative to minBorder), in reference order */
int   orc_level_candidates(const orc_extractor* ex, int level, int16_t* x, int16_t* y, uint8_t* score, int cap);
/* keypoints of one level after distribution + orientation (level coordinates, before pt*=scale) */
int   orc_level_keypoints(const orc_extractor* ex, int level, orc_keypoint* kps, int cap);

/* DistributeOctTree on its own (src/ORBextractor.cc:538-762). keys: x,y relative to (minX,minY); returns count */
int   orc_distribute_octree(const int16_t* x, const int16_t* y, const uint8_t* score, int n,
                            int minX, int maxX, int minY, int maxY, int N,
                            int32_t* out_index, int cap);

/* ---- matcher ---- */
int   orc_descriptor_distance(const uint8_t* a, const uint8_t* b);
void  orc_hamming_best2(const uint8_t* q, int nq, const uint8_t* t, int64_t nt, orc_best2* out);
void  orc_compute_three_maxima(const int32_t* sizes, int L, int* ind1, int* ind2, int* ind3);

typedef struct orc_frame_grid orc_frame_grid; /* 64x48 grid of Frame (src/Frame.cc:230-245, 327-392) */
orc_frame_grid* orc_grid_create(const orc_keypoint* kps, int n, float minX, float maxX, float minY, float maxY);
void  orc_grid_destroy(orc_frame_grid* g);
int   orc_grid_features_in_area(const orc_frame_grid* g, float x, float y, float r, int minLevel, int maxLevel,
                                int32_t* out, int cap);

int   orc_search_for_initialization(const orc_keypoint* k1, const uint8_t* d1, int n1,
                                    const orc_keypoint* k2, const uint8_t* d2, int n2,
                                    float minX, float maxX, float minY, float maxY,
                                    float* prev_matched_xy /* n1 x 2, in/out */, int window,
                                    float nnratio, int check_ori, int32_t* matches12 /* n1 */);

/* SearchByProjection(Frame, vector<MapPoint*>) inner part, src/ORBmatcher.cc:45-129, on plain arrays.
 * query i: projected (u,v), radius already multiplied by scale, predicted level, descriptor,
 * right-image u (proj_ur) used only when the frame key has u_right > 0. claimed[] (n2) is updated in place. */
int   orc_search_by_projection_mappoints(const float* qu, const float* qv, const float* qr, const float* qur,
                                         const int32_t* qlevel, const uint8_t* qdesc, int nq,
                                         const orc_keypoint* k2, const uint8_t* d2, const float* uright2, int n2,
                                         float minX, float maxX, float minY, float maxY,
                                         float nnratio, const uint8_t* q_transient /* may be null; 1: the query's map point has no
                                         observations, so its match (stored as -2 - query) leaves the feature free, :87-89 */,
                                         int32_t* claimed /* n2: -1 free, else query idx */);

/* SearchByProjection(Frame, LastFrame) inner part, src/ORBmatcher.cc:1381-1470: best-1, TH_HIGH, rot-hist.
 * mode: 0 = both neighbours (level-1..level+1), 1 = forward (>= level), 2 = backward (0..level) */
int   orc_search_by_projection_lastframe(const float* qu, const float* qv, const float* qradius, const float* qur,
                                         const int32_t* qoctave, const float* qangle, const uint8_t* qdesc, int nq,
                                         const orc_keypoint* k2, const uint8_t* d2, const float* uright2, int n2,
                                         float minX, float maxX, float minY, float maxY,
                                         int mode, int th_dist, int check_ori, const uint8_t* q_transient /* as above, :1406-1408 */,
                                         int32_t* claimed);

/* Frame::ComputeStereoMatches Hamming part, src/Frame.cc:466-549. best_dist/best_idx per left keypoint
 * (best_dist = 100 and idx = 0 when nothing beat the start value; -1 marks "skipped" queries). */
void  orc_stereo_match(const orc_keypoint* kl, const uint8_t* dl, int nl,
                       const orc_keypoint* kr, const uint8_t* dr, int nr,
                       const float* scale_factors, int nlevels, int n_rows,
                       float min_d, float max_d, int32_t* best_dist, int32_t* best_idx);

/* ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) candidate loop, src/ORBmatcher.cc:362-401 */
int   orc_search_by_projection_keyframe(const float* qu, const float* qv, const float* qradius, const int32_t* qlevel,
                                        const uint8_t* qdesc, int nq, const orc_keypoint* k2, const uint8_t* d2, int n2,
                                        float minX, float maxX, float minY, float maxY, int32_t* claimed);

/* Candidate loop of ORBmatcher::Fuse, src/ORBmatcher.cc:893-952 (reproj 1) and :1052-1082 (reproj 0) */
void  orc_fuse_candidates(const float* qu, const float* qv, const float* qur, const float* qradius, const int32_t* qlevel,
                          const uint8_t* qdesc, int nq, const orc_keypoint* k2, const uint8_t* d2, const float* uright2, int n2,
                          const float* inv_level_sigma2, float minX, float maxX, float minY, float maxY, int reproj,
                          int32_t* best_idx, int32_t* best_dist);
/* ORBmatcher::SearchBySim3 candidate loops and agreement, src/ORBmatcher.cc:1147-1328 */
int   orc_search_by_sim3(const uint8_t* valid1, const float* u1in2, const float* v1in2, const float* radius1, const int32_t* level1,
                         const uint8_t* mpdesc1, const orc_keypoint* k1, const uint8_t* d1, int n1,
                         const uint8_t* valid2, const float* u2in1, const float* v2in1, const float* radius2, const int32_t* level2,
                         const uint8_t* mpdesc2, const orc_keypoint* k2, const uint8_t* d2, int n2,
                         float minX, float maxX, float minY, float maxY, int32_t* match12);

/* DBoW2::FeatureVector flattened: node ids ascending; features of node k at features[node_ptr[k]..node_ptr[k+1]). */
typedef struct orc_feature_vector {
    const uint32_t* node_ids; const int32_t* node_ptr; const int32_t* features; int32_t n_nodes;
} orc_feature_vector;

/* ORBmatcher::SearchByBoW: keyframe_form 0 = (KeyFrame*, Frame&) src/ORBmatcher.cc:159-291, 1 = (KeyFrame*, KeyFrame*) :525-658.
 * eligible = "has a good map point" (side 2 of form 0: may be NULL = all).  Returns nmatches. */
int   orc_search_by_bow(const uint8_t* d1, const float* angle1, const uint8_t* eligible1, int n1, const orc_feature_vector* fv1,
                        const uint8_t* d2, const float* angle2, const uint8_t* eligible2, int n2, const orc_feature_vector* fv2,
                        float nnratio, int check_ori, int keyframe_form, int32_t* match12, int32_t* match21);

/* ORBmatcher::SearchForTriangulation src/ORBmatcher.cc:660-826 with CheckDistEpipolarLine :140-157.  Returns nmatches. */
int   orc_search_for_triangulation(const orc_keypoint* k1, const uint8_t* d1, const uint8_t* has_mp1, const float* uright1, int n1,
                                   const orc_feature_vector* fv1,
                                   const orc_keypoint* k2, const uint8_t* d2, const uint8_t* has_mp2, const float* uright2, int n2,
                                   const orc_feature_vector* fv2,
                                   const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                   int only_stereo, int check_ori, int32_t* match12);

/* Frame::ComputeStereoMatches sub-pixel part, src/Frame.cc:551-639, on the pyramids the two extractors built for the
 * pair: for every left keypoint with best_dist < (TH_HIGH+TH_LOW)/2 the 11x11 SAD search over +-5 columns at the
 * keypoint's level, parabola fit, disparity test; then the median filter over the accepted SAD distances.
 * uright/depth get -1 where there is no match; sad gets the winning SAD (-1 where none, kept for filtered ones). */
void  orc_stereo_refine(const orc_extractor* left, const orc_extractor* right,
                        const orc_keypoint* kl, int nl, const orc_keypoint* kr,
                        const int32_t* best_dist, const int32_t* best_idx,
                        float bf, float min_d, float max_d, float* uright, float* depth, int32_t* sad);

/* cv::cvtColor(src, dst, CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY) for 8-bit images, the conversion of
 * src/Tracking.cc:254-267 (OpenCV 2.4 / 3.x modules/imgproc/src/color.cpp, RGB2Gray<uchar>: Q14 weights R 4899, G 9617,
 * B 1868, (sum + 2^13) >> 14).  channels 3 or 4 (the fourth is ignored), blue_first 1 for BGR / BGRA order. */
void  orc_cvt_color_to_gray(const uint8_t* src, int w, int h, ptrdiff_t src_stride, int channels, int blue_first,
                            uint8_t* dst, ptrdiff_t dst_stride);

/* cv::undistortPoints(src, dst, K, dist, noArray(), K) as Frame::UndistortKeyPoints / ComputeImageBounds call it
 * (src/Frame.cc:404-456), OpenCV 2.4.9 - 3.4.1 modules/imgproc/src/undistort.cpp cvUndistortPoints: K and k1,k2,p1,p2,k3
 * widened to double, five iterations, projection through P = K, result rounded to float.  xy / xy_out: n (x, y) pairs. */
void  orc_undistort_points(const float* xy, int n, float fx, float fy, float cx, float cy,
                           float k1, float k2, float p1, float p2, float k3, float* xy_out);

/* cv::remap(src, dst, map_x, map_y, INTER_LINEAR, BORDER_CONSTANT, 0) for CV_8UC1 images and CV_32FC1 maps, the call of
 * Examples/Stereo/stereo_euroc.cc:136-137 (OpenCV 2.4 / 3.x modules/imgproc/src/imgwarp.cpp: initInterTab2D's Q15
 * bilinear table, RemapInvoker's conversion of the float maps to shorts + 1/32 fractions, remapBilinear). */
void  orc_remap_linear_u8(const uint8_t* src, int sw, int sh, ptrdiff_t sstride,
                          const float* map_x, const float* map_y, ptrdiff_t map_stride,
                          uint8_t* dst, int dw, int dh, ptrdiff_t dstride);

/* MapPoint::ComputeDistinctiveDescriptors selection, src/MapPoint.cc:513-547: of the n descriptors of one map point, the
 * index of the one with the least median Hamming distance to the set (median = sorted row [(int)(0.5*(n-1))], the
 * row including the zero distance to itself; first index wins ties).  -1 for n == 0. */
int   orc_distinctive_descriptor(const uint8_t* desc, int n);

/* DBoW2 vocabulary tree on plain arrays (what TemplatedVocabulary::loadFromTextFile reads, Thirdparty/DBoW2/DBoW2/
 * TemplatedVocabulary.h:1351-1437): node 0 is the root; node i >= 1 has parent[i] < i, a leaf flag, a 32-byte descriptor
 * and a weight.  Children keep node order; word ids number the leaves in node order. */
typedef struct orc_vocabulary {
    int32_t k, L, scoring, weighting;     /* ScoringType / WeightingType values of BowVector.h:36-53 */
    int32_t n_nodes;
    const int32_t* parent; const uint8_t* is_leaf; const uint8_t* descriptors; const double* weight;
} orc_vocabulary;

/* TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup), TemplatedVocabulary.h:1139-1207 with the
 * single-feature descent :1230-1272 and BowVector::addWeight/addIfNotExist/normalize (BowVector.cpp:34-84).
 * Per feature: word[i] and node[i] (the node at level L-levelsup; the reference leaves it unset when the descent ends
 * above that level, here it is then the leaf reached).  bow_ids/bow_values (capacity n) ascending by word id;
 * fv_* as orc_feature_vector (capacities n, n+1, n).  Returns 0, or -1 for a malformed tree. */
int   orc_vocabulary_transform(const orc_vocabulary* voc, const uint8_t* desc, int n, int levelsup,
                               int32_t* word, int32_t* node,
                               uint32_t* bow_ids, double* bow_values, int32_t* n_words,
                               uint32_t* fv_node_ids, int32_t* fv_node_ptr, int32_t* fv_features, int32_t* fv_n_nodes);

#ifdef __cplusplus
}
#endif
#endif
