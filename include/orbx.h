/*
 * orbx.h -- C ABI of the MI355X-native ORB front-end (liborbx.so).
 *
 * The reference has no FFI: the path sits behind two plain C++ classes,
 * ORB_SLAM2::ORBextractor (include/ORBextractor.h:74-282) and ORB_SLAM2::ORBmatcher
 * (include/ORBmatcher.h:56-330), plus the Frame helpers the matchers call
 * (src/Frame.cc:230-245, 327-392, 466-549).  Each entry point below names the reference
 * interface it replaces.  Plain pointers and sizes only; no exceptions cross the boundary.
 *
 * Conventions
 *  - every function returns an orbx_status (0 = ok) unless it is a pure scalar helper;
 *  - "host" entry points take host pointers and are synchronous;
 *  - "*_device" entry points take device pointers (HBM resident), enqueue on the given
 *    hipStream_t (passed as void*, NULL = the HIP default stream) and do not synchronise;
 *  - a context (extractor / matcher) owns one HIP stream and its workspace; distinct contexts
 *    may be used concurrently from different host threads, one context is not re-entrant
 *    (same rule as the reference: src/Frame.cc:78-81 runs two extractor instances in two threads).
 */
#ifndef ORBX_H
#define ORBX_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum orbx_status {
    ORBX_OK = 0,
    ORBX_ERR_BAD_ARG = 1,
    ORBX_ERR_UNSUPPORTED_GEOMETRY = 2, /* the reference is undefined for this image size (width/height < 0.5: empty root list, src/ORBextractor.cc:542-568; a level smaller than its own 16-pixel borders) or the size exceeds the packed-coordinate range.  Levels too small for one FAST cell (:783-786) are accepted: they yield no keypoints */
    ORBX_ERR_HIP = 3,
    ORBX_ERR_CAPACITY = 4,             /* caller's output capacity too small */
    ORBX_ERR_NO_DEVICE = 5
} orbx_status;

const char* orbx_status_string(int status);
const char* orbx_last_error(void);     /* thread-local text of the last failure */
/* Short hash of the library's sources, fixed when it is built: lets a stored counter profile say which kernels it was
 * taken on (no counterpart in the reference). */
const char* orbx_build_stamp(void);

/* The one partition both multi-GPU paths use (no counterpart in the reference, whose only parallelism is the two
 * extractor threads of src/Frame.cc:78-81): rank r of `world` owns units [begin, end) = [r*total/world,
 * (r+1)*total/world) of a frame batch (no collective follows) or of the map rows of a sharded brute-force search
 * (orbx_best2_merge* follows the all-gather).  Host-only. */
int orbx_shard_range(int64_t total, int rank, int world, int64_t* begin, int64_t* end);

/* Constructor arguments of ORBextractor (include/ORBextractor.h:83-84) plus one knob for the
 * OpenCV build the caller wants to match. */
typedef struct orbx_params {
    int32_t nfeatures;
    float   scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t blur_rounding; /* 0: generic C++ column pass, (sum+32768)>>16.  1: x86 SIMD column pass of OpenCV <= 3.4.0 (exact value rounded half-to-even for x < (w & ~3), generic for the last w%4 columns) */
    int32_t trig_mode;     /* cos / sin of the keypoint angle in computeOrbDescriptor (src/ORBextractor.cc:111-112, which calls the float overloads, i.e. glibc cosf / sinf).  0: the correctly rounded (float)cos((double)theta).  1: glibc >= 2.28 cosf / sinf as x86-64 CPUs with FMA and AVX2 run it (the contracted __cosf_fma / __sinf_fma), bit for bit.  2: the same routine evaluated without fused multiply-adds (x86-64 without FMA) */
} orbx_params;

/* Binary layout of cv::KeyPoint: pt.x, pt.y, size, angle, response, octave, class_id (28 bytes). */
typedef struct orbx_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbx_keypoint;

typedef struct orbx_best2 { int32_t best, best_idx, second, pad; } orbx_best2;

/* ------------------------------------------------------------------------------------------ */
/* Extractor: replaces ORBextractor (src/ORBextractor.cc:409-469 ctor, :1042-1104 operator()) */
/* ------------------------------------------------------------------------------------------ */
typedef struct orbx_extractor orbx_extractor;

/* A context is bound to one image size and a maximum batch (frames per launch).  The C++ drop-in
 * class re-creates it when the image size changes.  device < 0: current device. */
int orbx_extractor_create(const orbx_params* p, int width, int height, int max_batch, int device,
                          orbx_extractor** out);
void orbx_extractor_destroy(orbx_extractor* ex);

/* Host-only tables; valid without a GPU.  GetScaleFactors / GetInverseScaleFactors /
 * GetScaleSigmaSquares / GetInverseScaleSigmaSquares (include/ORBextractor.h:121-143),
 * mnFeaturesPerLevel and umax (src/ORBextractor.cc:434-468).  Any pointer may be NULL. */
int orbx_scale_tables(const orbx_params* p, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                      int32_t* features_per_level, int32_t* umax16);
/* Level size as ComputePyramid derives it (src/ORBextractor.cc:1110-1111). */
int orbx_level_size(const orbx_params* p, int width, int height, int level, int* w, int* h);
/* Upper bound of keypoints one frame can return (quota + 3 per level: the quadtree can overshoot). */
int orbx_keypoint_capacity(const orbx_params* p, int width, int height);

/* Layout of the images the extract calls take.  The reference converts colour input to grey on the host before
 * extraction (Tracking::GrabImageMonocular / GrabImageStereo / GrabImageRGBD, src/Tracking.cc:254-267 and the two
 * sibling blocks: cvtColor with CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY by channel count and mbRGB);
 * here the first pyramid kernel does it while it reads the frame: grey = (4899 R + 9617 G + 1868 B + 8192) >> 14, the
 * 8-bit arithmetic of cv::cvtColor.  Default ORBX_INPUT_GRAY.  With a colour format, row_stride and frame_stride of the
 * extract calls stay in bytes and a row holds width * channels bytes.  Call between extract calls, not during one. */
enum { ORBX_INPUT_GRAY = 0, ORBX_INPUT_RGB = 1, ORBX_INPUT_BGR = 2, ORBX_INPUT_RGBA = 3, ORBX_INPUT_BGRA = 4 };
int orbx_extractor_set_input_format(orbx_extractor* ex, int format);

/* ORBextractor::operator() for one host image.  kps/desc sized cap (desc: cap x 32 bytes). */
int orbx_extract(orbx_extractor* ex, const uint8_t* image, ptrdiff_t stride,
                 orbx_keypoint* kps, uint8_t* desc, int cap, int* n);
/* The same for n_frames host images (uploaded, processed as one batch, downloaded). */
int orbx_extract_batch(orbx_extractor* ex, const uint8_t* images, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                       int n_frames, orbx_keypoint* kps, uint8_t* desc, int32_t* counts, int cap_per_frame);
/* Device-resident batch: images, keypoints, descriptors and counts all in HBM; asynchronous, on `stream`.  The calls
 * that read the context's state of the last batch afterwards (orbx_pyramid_level(s), the orbx_debug_* taps,
 * orbx_stereo_refine_batch_device on any stream) order themselves behind it through an event the batch records: the
 * caller needs no synchronisation of its own for them, only for its own reads of the output arrays.  That guarantee
 * holds for batches issued outside stream capture, one stream at a time per context: a batch recorded into the caller's
 * own graph records no event (it would become a node of that graph), and after replaying such a graph, or after
 * issuing batches of one context on several streams, the caller synchronises before those calls itself. */
int orbx_extract_batch_device(orbx_extractor* ex, const uint8_t* d_images, ptrdiff_t row_stride,
                              ptrdiff_t frame_stride, int n_frames, orbx_keypoint* d_kps, uint8_t* d_desc,
                              int32_t* d_counts, int cap_per_frame, void* stream);

/* One call, several contexts (no counterpart in the reference, whose callers run one extractor per thread and two for
 * stereo, src/Frame.cc:78-81): the batch is cut into n_contexts contiguous shares, context c extracts share c on its own
 * internal stream, and `stream` is joined behind all of them, so a caller with one stream gets the overlap of several
 * contexts without managing streams (kernels of different contexts fill each other's gaps; four contexts is what a
 * process's four hardware queues run side by side).  The contexts must have been created with equal parameters, image
 * size, input format and device, each with max_batch >= ceil(n_frames / n_contexts).  Work queued on `stream` before the
 * call (the producer of the frames) is waited for by every context.  d_images == NULL: context c reads the frames of its
 * share from its own input view (orbx_extractor_input_view), where the caller has put them.  Outputs as for
 * orbx_extract_batch_device, frame f of the batch at f * cap_per_frame.  Not under stream capture.
 * flags: ORBX_MULTI_NO_JOIN leaves `stream` unjoined, for callers that queue several batches back to back (other
 * output buffers, or results not needed in between): the contexts then run on without meeting at the end of every call,
 * which is worth about 8 % at 1024 frames per call; orbx_extract_multi_join(exs, n_contexts, stream) makes `stream` wait
 * for everything the contexts have been given so far. */
enum { ORBX_MULTI_NO_JOIN = 1 };
int orbx_extract_batch_device_multi(orbx_extractor* const* exs, int n_contexts, const uint8_t* d_images, ptrdiff_t row_stride,
                                    ptrdiff_t frame_stride, int n_frames, orbx_keypoint* d_kps, uint8_t* d_desc,
                                    int32_t* d_counts, int cap_per_frame, void* stream, int flags);
int orbx_extract_multi_join(orbx_extractor* const* exs, int n_contexts, void* stream);

/* Zero-copy input (no counterpart in the reference, whose ComputePyramid copies the image into its padded level 0,
 * src/ORBextractor.cc:1124-1128): where level 0 of the context's pyramid lives.  A producer that can write its
 * grey frames there -- frame f, row y, column x at d_images[f * frame_stride + y * row_stride + x], f < max_batch --
 * (orbx_remap_batch_device, a colour conversion, an upload) and then passes exactly this pointer and these strides to
 * orbx_extract_batch_device spares the batch its copy of the input: only the REFLECT_101 frame around the frames is
 * written.  The bytes between the rows belong to the pyramid's frame and may be overwritten by either side.  Grey
 * input only; any other pointer or stride is copied as before. */
int orbx_extractor_input_view(orbx_extractor* ex, uint8_t** d_images, ptrdiff_t* row_stride, ptrdiff_t* frame_stride);

/* Pinhole camera with the radial-tangential model of the reference's settings files (Camera.fx .. Camera.k3,
 * src/Tracking.cc:56-80): K = [fx 0 cx; 0 fy cy; 0 0 1] and dist = k1, k2, p1, p2, k3 as mK / mDistCoef hold them (floats). */
typedef struct orbx_camera { float fx, fy, cx, cy; float k1, k2, p1, p2, k3; } orbx_camera;
/* Frame::UndistortKeyPoints (src/Frame.cc:404-434) on the output of a batched extraction: cv::undistortPoints(pts, pts,
 * mK, mDistCoef, Mat(), mK), i.e. normalise, five fixed-point iterations of the inverse distortion in double, project
 * back through K, round to float.  d_kps_un receives copies of the keypoints with pt replaced (mvKeysUn); with
 * cam->k1 == 0 they are plain copies, as in the reference (:406-410).  Layout as orbx_extract_batch_device leaves it
 * (blocks of cap_per_frame, d_counts valid rows each).  Asynchronous on `stream`. */
int orbx_undistort_keypoints_batch_device(orbx_extractor* ex, const orbx_camera* cam, const orbx_keypoint* d_kps,
                                          const int32_t* d_counts, int cap_per_frame, int n_frames,
                                          orbx_keypoint* d_kps_un, void* stream);
/* The same for n host points (x, y pairs), e.g. one frame's keypoints or the four image corners of
 * Frame::ComputeImageBounds (src/Frame.cc:436-456); unlike the call above it always undistorts.  Synchronous. */
int orbx_undistort_points(orbx_extractor* ex, const orbx_camera* cam, const float* xy, int n, float* xy_out);

/* mvImagePyramid[level] of frame `frame` of the last batch (include/ORBextractor.h:150): the padded
 * (w+38) x (h+38) buffer is copied to dst (host); the image ROI starts at row 19, column 19. */
int orbx_pyramid_level(orbx_extractor* ex, int frame, int level, uint8_t* dst, ptrdiff_t dst_stride);
/* All levels of one frame in one round trip (what ORBextractor::operator() needs to refill mvImagePyramid):
 * dst[l] / dst_stride[l] as for orbx_pyramid_level. */
int orbx_pyramid_levels(orbx_extractor* ex, int frame, uint8_t* const* dst, const ptrdiff_t* dst_stride);
/* Device view of the same level: pointer to the ROI origin and its row stride.  The batch path fills the
 * 3 innermost pixels of the REFLECT_101 frame (all any stage reads); orbx_pyramid_level completes the
 * 19-pixel frame of the level it copies. */
int orbx_pyramid_level_device(orbx_extractor* ex, int frame, int level, const uint8_t** d_roi,
                              int* w, int* h, ptrdiff_t* stride);

/* Parity taps on the last batch.  Candidates handed to DistributeOctTree (src/ORBextractor.cc:817-825),
 * in the reference's order; packed as x<<20 | y<<8 | score with x,y relative to minBorder (16). */
int orbx_debug_candidates(orbx_extractor* ex, int frame, int level, uint32_t* packed, int cap, int* n);
/* Keypoints chosen by DistributeOctTree for one level, list order, same packing. */
int orbx_debug_level_keypoints(orbx_extractor* ex, int frame, int level, uint32_t* packed, int cap, int* n);

/* GaussianBlur(7x7, sigma 2) of one level of the last batch (the workingMat of src/ORBextractor.cc:1084-1085),
 * w x h pixels copied to dst (host). */
int orbx_debug_blurred_level(orbx_extractor* ex, int frame, int level, uint8_t* dst, ptrdiff_t dst_stride);
/* Test tap: cos and sin as orbx_params.trig_mode defines them, of angle * (float)(pi/180) (src/ORBextractor.cc:111-112)
 * for the `count` float angles whose bit patterns start at first_bits; d_cos / d_sin are device arrays of `count`
 * floats; runs on `stream`.  Lets a test walk every float angle in [0, 360) against the host's cosf / sinf. */
int orbx_debug_trig_range(int trig_mode, uint32_t first_bits, uint32_t count, float* d_cos, float* d_sin, void* stream);

/* DistributeOctTree on its own (src/ORBextractor.cc:538-762): host arrays in, device kernel, host out.
 * keys packed as above; out receives the selected keys in list order. */
int orbx_distribute_octree(int device, const uint32_t* packed_keys, int n, int minX, int maxX, int minY, int maxY,
                           int N, uint32_t* out, int cap, int* n_out);
/* Same, choosing where the kernel keeps the keys: 0 = in LDS when they fit (the production path),
 * 1 = always device memory (the path taken when a level has more keys than the LDS arrays hold).  For parity tests. */
int orbx_distribute_octree_ex(int device, const uint32_t* packed_keys, int n, int minX, int maxX, int minY, int maxY,
                              int N, uint32_t* out, int cap, int* n_out, int key_storage);

/* Per-stage device time of the last batch, measured with HIP events on the launch stream when
 * profiling is switched on (no counterpart in the reference).  Stage order: ORBX_STAGE_*, which is the order of the
 * launches: level 0 and the resizes, the Gaussian of every level, the FAST cells, the quadtree selection, orientation +
 * descriptors (each of the last four is one kernel). */
enum { ORBX_STAGE_PYRAMID = 0, ORBX_STAGE_BLUR = 1, ORBX_STAGE_FAST = 2, ORBX_STAGE_OCTREE = 3, ORBX_STAGE_DESCRIBE = 4, ORBX_STAGE_COUNT = 5 };
int orbx_extractor_set_profiling(orbx_extractor* ex, int on);
int orbx_extractor_stage_ms(orbx_extractor* ex, float* ms /* ORBX_STAGE_COUNT */);

/* ------------------------------------------------------------------------------------------ */
/* Matcher: replaces the Hamming loops of ORBmatcher (src/ORBmatcher.cc)                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct orbx_matcher orbx_matcher;
int orbx_matcher_create(int device, orbx_matcher** out);
void orbx_matcher_destroy(orbx_matcher* m);

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1650-1666): static and pure, host scalar. */
int orbx_descriptor_distance(const uint8_t* a, const uint8_t* b);
/* ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1604-1645). */
void orbx_compute_three_maxima(const int32_t* sizes, int L, int* ind1, int* ind2, int* ind3);

/* Best / second-best Hamming distance of every query over a whole descriptor set, with the
 * sequential loop's semantics (strict '<': lowest index wins ties; start values 256 / -1 / 256),
 * i.e. the inner loop of every Search* routine (e.g. src/ORBmatcher.cc:439-460) without a window.
 * Runs on the matrix cores as an exact int8 product (distance = popc(q) + dot of the query's +-1 bytes with the row's
 * 0/1 bytes).  d_q must be 16-byte aligned, d_t 8-byte aligned (ORBX_ERR_BAD_ARG otherwise).  The matcher's scratch is shared by its calls: calls on one
 * matcher must not overlap in time (use one matcher per stream). */
int orbx_hamming_best2(orbx_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int64_t nt, orbx_best2* out);
int orbx_hamming_best2_device(orbx_matcher* m, const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt,
                              int64_t index_base, orbx_best2* d_out, void* stream);
/* Measurement tap (no counterpart in the reference): the brute-force kernel multiplies every tile of 16 rows x 16 queries
 * on the matrix cores but walks a tile's keys only when one of its distances can still change a result, so its speed
 * depends on the data.  *walked = tiles walked on the matcher's GPU since the last reset, over all calls of the process
 * on that GPU (synchronises the device); the tiles multiplied are ceil(rows / 16) x ceil(queries / 16) per call. */
int orbx_debug_best2_tile_walks(orbx_matcher* m, uint64_t* walked, int reset);
/* No counterpart in the reference.  orbx_hamming_best2_device for callers that cannot allocate device memory: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same arguments, same results; the kernels'
 * scratch (per-run partial results and the seed records of the two-pass search) is a fixed array in the library's device
 * image (4 MiB + 256 KiB, held by every GPU the library runs on for the life of the process) whose address on the matcher's
 * device was resolved at orbx_matcher_create.  The call itself makes no hipMalloc / hipFree.  Limits: nq <= 16384 and
 * orbx_hamming_best2_workspace_bytes(nq, nt) - 16 * nq <= 4 MiB (the row-run split keeps this near 2 MiB for every map
 * size at nq <= 16384; e.g. 4096 x 2^24 fits); beyond them ORBX_ERR_CAPACITY.  All calls on one GPU share the array, whichever matcher
 * they go through.  Outside stream capture the library orders them itself: every call waits (on its stream) for an event
 * that the previous call on that GPU recorded behind its kernels, so calls from several matchers, streams and host threads
 * run one after the other instead of racing.  A call issued while its stream is being captured takes no part in that
 * (an event wait would become a node of the caller's graph): captured calls, and replays of graphs that hold them, must
 * not overlap with other users of the array on the same GPU -- that ordering stays the caller's. */
int orbx_hamming_best2_device_static(orbx_matcher* m, const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt,
                                     int64_t index_base, orbx_best2* d_out, void* stream);
/* Merge per-shard results (ranks x nq records, shard-major; best_idx already global) into nq records:
 * the step after the RCCL all-gather when the map is sharded over GPUs. */
int orbx_best2_merge_device(orbx_matcher* m, const orbx_best2* d_parts, int n_parts, int nq, orbx_best2* d_out, void* stream);
/* The same merge on host arrays (shards gathered over a network instead of xGMI). */
int orbx_best2_merge(const orbx_best2* parts, int n_parts, int nq, orbx_best2* out);
/* Bytes of scratch the brute-force kernel keeps per call, for sizing checks. */
int64_t orbx_hamming_best2_workspace_bytes(int nq, int64_t nt);

/* Windowed form: distances of query i to candidates cand[rowptr[i] .. rowptr[i+1]) (CSR), for the
 * order-dependent routines whose claim pass stays sequential on the host. */
int orbx_hamming_csr(orbx_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt,
                     const int32_t* rowptr, const int32_t* cand, int32_t* dist_out);

/* Frame::AssignFeaturesToGrid / GetFeaturesInArea (src/Frame.cc:230-245, 327-392) on plain arrays. */
typedef struct orbx_frame_grid orbx_frame_grid;
int orbx_grid_create(const orbx_keypoint* kps, int n, float minX, float maxX, float minY, float maxY, orbx_frame_grid** out);
void orbx_grid_destroy(orbx_frame_grid* g);
int orbx_grid_features_in_area(const orbx_frame_grid* g, float x, float y, float r, int minLevel, int maxLevel,
                               int32_t* out, int cap, int* n);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:408-523) on plain arrays: frame bounds are
 * Frame::mnMinX.. (src/Frame.cc:457-463); prev_matched_xy is vbPrevMatched (updated); matches12 is
 * vnMatches12; *nmatches is the return value. */
int orbx_search_for_initialization(orbx_matcher* m,
                                   const orbx_keypoint* k1, const uint8_t* d1, int n1,
                                   const orbx_keypoint* k2, const uint8_t* d2, int n2,
                                   float minX, float maxX, float minY, float maxY,
                                   float* prev_matched_xy, int window, float nnratio, int check_ori,
                                   int32_t* matches12, int* nmatches);

/* The same search for a batch of frame pairs, everything on the device and no host round trip (one wavefront per pair
 * walks the queries in order; candidates, distances, the claim / steal step of :447 and :466-473, ratio test, rotation
 * histogram and the vbPrevMatched update of :518-520 all happen there).  Keypoint, descriptor and count arrays are laid
 * out like the outputs of orbx_extract_batch_device (pair p at offset p * cap); pass a batch's own outputs, or its
 * undistorted copy, directly.  d_prev_matched_xy[n_pairs][cap][2] in / out, d_matches12[n_pairs][cap] (-1 past each
 * count), d_nmatches[n_pairs].  While 53 bytes per keypoint of cap fit 150 KB of LDS (cap up to about 2890) everything a
 * pair needs lives there and nothing is allocated; above that -- the reference's initialisation extractor runs at
 * 2 * nFeatures, src/Tracking.cc:136-137 -- frame 2's descriptors and positions go to a device-memory arena of the matcher
 * (44 bytes per keypoint and pair, grown with hipMalloc when a call needs more than the last: calls on one matcher must
 * not overlap) and LDS keeps the 9 bytes per keypoint of the sequential pass.  ORBX_ERR_CAPACITY above about 17000.
 * Runs on `stream`. */
int orbx_search_for_initialization_batch_device(orbx_matcher* m,
                                                const orbx_keypoint* d_k1, const uint8_t* d_d1, const int32_t* d_n1,
                                                const orbx_keypoint* d_k2, const uint8_t* d_d2, const int32_t* d_n2, int cap, int n_pairs,
                                                float minX, float maxX, float minY, float maxY,
                                                float* d_prev_matched_xy, int window, float nnratio, int check_ori,
                                                int32_t* d_matches12, int32_t* d_nmatches, void* stream);


/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), candidate loop and acceptance
 * (src/ORBmatcher.cc:68-125).  Query i = one visible map point: projection (qu,qv), search radius
 * qr = r*scale[level] (:69), right-image projection qur (:93), predicted level, descriptor.
 * claimed[n2]: -1 free, >=0 taken (stands for F.mvpMapPoints[idx] with observations, :87-89); a match
 * stores the query index (:123).  q_transient (may be null): 1 marks a query whose map point has no observations
 * (the temporal points stereo / RGB-D tracking creates, src/Tracking.cc UpdateLastFrame): the reference's test
 * `F.mvpMapPoints[idx]->Observations()>0` then leaves the feature free for later queries, which may overwrite it;
 * such a match is stored as -2 - query (still negative, still free). */
int orbx_search_by_projection_mappoints(orbx_matcher* m,
                                        const float* qu, const float* qv, const float* qr, const float* qur,
                                        const int32_t* qlevel, const uint8_t* qdesc, int nq,
                                        const orbx_keypoint* k2, const uint8_t* d2, const float* uright2, int n2,
                                        float minX, float maxX, float minY, float maxY,
                                        float nnratio, const uint8_t* q_transient, int32_t* claimed, int* nmatches);

/* ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (src/ORBmatcher.cc:1381-1470) and the
 * KeyFrame form (:1531-1598): best-1 within th_dist (TH_HIGH=100 or ORBdist), rotation histogram.
 * mode 0: levels octave-1..octave+1, 1: forward (>= octave), 2: backward (0..octave).  q_transient as above
 * (:1406-1408 is the same test); the key-frame form blocks on any map point (:1544-1545): pass null. */
int orbx_search_by_projection_lastframe(orbx_matcher* m,
                                        const float* qu, const float* qv, const float* qradius, const float* qur,
                                        const int32_t* qoctave, const float* qangle, const uint8_t* qdesc, int nq,
                                        const orbx_keypoint* k2, const uint8_t* d2, const float* uright2, int n2,
                                        float minX, float maxX, float minY, float maxY,
                                        int mode, int th_dist, int check_ori, const uint8_t* q_transient, int32_t* claimed, int* nmatches);

/* Both searches above for a batch of calls, everything on the device and no host round trip (one wavefront per call:
 * the frame's features are bucketed by grid cell in LDS, the queries are walked in order, candidates, distances, the
 * claim step, ratio test / rotation histogram all happen there).  form 0: the last-frame / relocalisation search
 * (mode, th_dist, check_ori as above); form 1: the map-point search (nnratio).  Call c reads its queries at c * qcap
 * (d_nq[c] of them; d_qur may be null when d_uright2 is, d_qangle when check_ori is 0, d_q_transient always) and its
 * frame at c * cap (d_n2[c] features; d_uright2 may be null), and updates d_claimed[c * cap ..] in place as the host
 * forms do; d_nmatches[c] receives the count.  While 57 * cap + 3 * qcap + 12 KB fit 150 KB of LDS (cap about 2400 at
 * qcap 1000) nothing is allocated; above that the frame's descriptors and per-feature constants go to a device-memory
 * arena of the matcher (53 bytes per feature and call, grown with hipMalloc when needed: calls on one matcher must not
 * overlap) and LDS keeps the claim states and the cell table (4 * cap + 3 * qcap + 12 KB): ORBX_ERR_CAPACITY only when
 * that exceeds 150 KB (cap about 30000 at qcap 4000) or cap reaches 65536.  Runs on `stream`. */
int orbx_search_by_projection_batch_device(orbx_matcher* m, int form,
                                           const float* d_qu, const float* d_qv, const float* d_qradius, const float* d_qur,
                                           const int32_t* d_qlevel, const float* d_qangle, const uint8_t* d_qdesc, const uint8_t* d_q_transient,
                                           const int32_t* d_nq, int qcap,
                                           const orbx_keypoint* d_k2, const uint8_t* d_d2, const float* d_uright2, const int32_t* d_n2, int cap,
                                           int n_calls, float minX, float maxX, float minY, float maxY, int mode, int th_dist, float nnratio,
                                           int check_ori, int32_t* d_claimed, int32_t* d_nmatches, void* stream);

/* ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, vector<MapPoint*>& vpMatched, th),
 * candidate loop and acceptance (src/ORBmatcher.cc:362-401).  Query i = one projected map point that passed the
 * reference's visibility tests: projection (qu,qv), radius th*scale[level] (:363), predicted level, descriptor.
 * claimed[n2]: -1 free, >= 0 taken (vpMatched[idx] non-null, :378); a match stores the query index (:399).
 * Best-1 among free key-frame features of level-1..level, accepted when <= TH_LOW. */
int orbx_search_by_projection_keyframe(orbx_matcher* m,
                                       const float* qu, const float* qv, const float* qradius, const int32_t* qlevel,
                                       const uint8_t* qdesc, int nq,
                                       const orbx_keypoint* k2, const uint8_t* d2, int n2,
                                       float minX, float maxX, float minY, float maxY, int32_t* claimed, int* nmatches);

/* The candidate loop that ORBmatcher::Fuse (both forms, src/ORBmatcher.cc:893-952 and :1052-1082) runs per projected
 * map point: the key-frame features within qradius of (qu, qv) at level qlevel-1..qlevel, nearest descriptor wins (first
 * on ties).  With qur != NULL the reprojection test of Fuse(pKF, vpMapPoints, th) is applied first (:917-941: squared
 * error times inv_level_sigma2[octave] above 7.8 with a right coordinate, 5.99 without).  best_idx[i] = feature or -1,
 * best_dist[i] = its distance or INT_MAX; the caller accepts <= TH_LOW and does the map-point bookkeeping (:955-974). */
int orbx_project_best1(orbx_matcher* m,
                       const float* qu, const float* qv, const float* qradius, const int32_t* qlevel,
                       const uint8_t* qdesc, int nq,
                       const orbx_keypoint* k2, const uint8_t* d2, int n2,
                       float minX, float maxX, float minY, float maxY,
                       const float* qur, const float* uright2, const float* inv_level_sigma2, int nlevels,
                       int32_t* best_idx, int32_t* best_dist);

/* ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1105-1329), candidate loops and agreement: feature i1 of key frame 1 with
 * valid1[i1] (has a good map point, not matched yet, projection into key frame 2 passed the depth / image / distance tests,
 * :1153-1186) searches key frame 2 around (u1in2, v1in2) within radius1 at level1-1..level1 with its map point's descriptor
 * mpdesc1 (best-1, accepted <= TH_HIGH); the same from key frame 2 into 1; match12[i1] = i2 where both agree, else -1. */
int orbx_search_by_sim3(orbx_matcher* m,
                        const uint8_t* valid1, const float* u1in2, const float* v1in2, const float* radius1, const int32_t* level1,
                        const uint8_t* mpdesc1, const orbx_keypoint* k1, const uint8_t* d1, int n1,
                        const uint8_t* valid2, const float* u2in1, const float* v2in1, const float* radius2, const int32_t* level2,
                        const uint8_t* mpdesc2, const orbx_keypoint* k2, const uint8_t* d2, int n2,
                        float minX, float maxX, float minY, float maxY, int32_t* match12, int* nfound);

/* A DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>) flattened: node ids ascending, the feature
 * indices of node k in features[node_ptr[k] .. node_ptr[k+1]) in their list order. */
typedef struct orbx_feature_vector {
    const uint32_t* node_ids;
    const int32_t* node_ptr;
    const int32_t* features;
    int32_t n_nodes;
} orbx_feature_vector;

/* ORBmatcher::SearchByBoW, both forms: features of side 1 are matched against the features of side 2 that share their
 * vocabulary node, best-2 with ratio test, TH_LOW, rotation histogram.
 *   keyframe_form 0: SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:159-291): side 1 = key frame (eligible1[i]:
 *     feature i has a good map point), side 2 = frame (eligible2 all ones, or NULL); accepts best <= TH_LOW;
 *     the reference's vpMapPointMatches[i2] is the map point of key-frame feature match21[i2].
 *   keyframe_form 1: SearchByBoW(KeyFrame*, KeyFrame*, ...) (:525-658): eligible on both sides = has a good map point;
 *     accepts best < TH_LOW; the reference's vpMatches12[i1] is the map point of feature match12[i1] of key frame 2.
 * angle1 / angle2: keypoint angles the reference feeds the histogram with.  match12[n1], match21[n2]: -1 = none. */
int orbx_search_by_bow(orbx_matcher* m,
                       const uint8_t* d1, const float* angle1, const uint8_t* eligible1, int n1, const orbx_feature_vector* fv1,
                       const uint8_t* d2, const float* angle2, const uint8_t* eligible2, int n2, const orbx_feature_vector* fv2,
                       float nnratio, int check_ori, int keyframe_form,
                       int32_t* match12, int32_t* match21, int* nmatches);

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:660-826) with CheckDistEpipolarLine (:140-157): features of key
 * frame 1 without a map point against features of key frame 2 without one in the same vocabulary node, best-1 within
 * TH_LOW subject to the epipole-distance and epipolar-line tests.  k1/k2 = mvKeysUn, uright = mvuRight (>= 0: stereo),
 * F12 row-major 3x3, (ex, ey) the epipole in image 2 (:667-673), scale_factors2 / level_sigma2_2 of key frame 2.
 * match12[n1]: index in key frame 2 or -1 (vMatchedPairs is its list form). */
int orbx_search_for_triangulation(orbx_matcher* m,
                                  const orbx_keypoint* k1, const uint8_t* d1, const uint8_t* has_mp1, const float* uright1, int n1,
                                  const orbx_feature_vector* fv1,
                                  const orbx_keypoint* k2, const uint8_t* d2, const uint8_t* has_mp2, const float* uright2, int n2,
                                  const orbx_feature_vector* fv2,
                                  const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                  int nlevels, int only_stereo, int check_ori, int32_t* match12, int* nmatches);

/* Frame::ComputeStereoMatches, Hamming part (src/Frame.cc:466-549): per left keypoint the best right
 * keypoint in its row band (start value TH_HIGH=100, index 0); best_dist = -1 marks a left keypoint the
 * reference skips before the loop (:513-520). */
int orbx_stereo_match(orbx_matcher* m,
                      const orbx_keypoint* kl, const uint8_t* dl, int nl,
                      const orbx_keypoint* kr, const uint8_t* dr, int nr,
                      const float* scale_factors, int nlevels, int n_rows, float min_d, float max_d,
                      int32_t* best_dist, int32_t* best_idx);
/* Device-resident batch of stereo pairs straight from two orbx_extract_batch_device outputs. */
int orbx_stereo_match_batch_device(orbx_matcher* m,
                                   const orbx_keypoint* d_kl, const uint8_t* d_dl, const int32_t* d_nl,
                                   const orbx_keypoint* d_kr, const uint8_t* d_dr, const int32_t* d_nr,
                                   int cap_per_frame, int n_pairs,
                                   const float* scale_factors, int nlevels, int n_rows, float min_d, float max_d,
                                   int32_t* d_best_dist, int32_t* d_best_idx, void* stream);

/* Frame::ComputeStereoMatches, sub-pixel part (src/Frame.cc:551-639), on the pyramids that `left` and `right` hold
 * for frames 0..n_pairs-1 of their last batches (the pair's two images, extracted with identical settings): for every
 * left keypoint whose Hamming match is below (TH_HIGH+TH_LOW)/2 the 11x11 SAD search over +-5 columns at the
 * keypoint's level, the parabola fit and the disparity test (min_d <= d < max_d, depth = bf / d), then the median
 * filter of the SAD distances (1.5 * 1.4 * median).  d_uright / d_depth are mvuRight / mvDepth (-1 where there is
 * no match); d_sad is scratch of the same shape (the winning SAD per keypoint).  Device pointers, asynchronous. */
int orbx_stereo_refine_batch_device(orbx_extractor* left, orbx_extractor* right,
                                    const orbx_keypoint* d_kl, const int32_t* d_nl, const orbx_keypoint* d_kr,
                                    int cap_per_frame, int n_pairs,
                                    const int32_t* d_best_dist, const int32_t* d_best_idx,
                                    float bf, float min_d, float max_d,
                                    float* d_uright, float* d_depth, int32_t* d_sad, void* stream);
/* The same for one pair with host arrays (frame 0 of each extractor's last batch); synchronous. */
int orbx_stereo_refine(orbx_extractor* left, orbx_extractor* right,
                       const orbx_keypoint* kl, int nl, const orbx_keypoint* kr, int nr,
                       const int32_t* best_dist, const int32_t* best_idx, float bf, float min_d, float max_d,
                       float* uright, float* depth);

/* ------------------------------------------------------------------------------------------ */
/* Stereo rectification ahead of extraction                                                    */
/* ------------------------------------------------------------------------------------------ */
/* cv::remap(src, dst, M1, M2, cv::INTER_LINEAR) for 8-bit single-channel images with the two CV_32F maps that
 * cv::initUndistortRectifyMap returns (Examples/Stereo/stereo_euroc.cc:97-98, 136-137): map_x / map_y give, per
 * destination pixel, the source position (rows map_stride floats apart).  Fixed-point as in OpenCV: positions rounded
 * to 1/32 pixel, Q15 weights, (sum + 2^14) >> 15, constant border 0.  The maps are converted once at create. */
typedef struct orbx_remapper orbx_remapper;
int orbx_remap_create(int device, const float* map_x, const float* map_y, ptrdiff_t map_stride, int dst_w, int dst_h,
                      int src_w, int src_h, orbx_remapper** out);
void orbx_remap_destroy(orbx_remapper* r);
/* n_frames device-resident images (the output can be handed to orbx_extract_batch_device); asynchronous. */
int orbx_remap_batch_device(orbx_remapper* r, const uint8_t* d_src, ptrdiff_t src_row_stride, ptrdiff_t src_frame_stride,
                            int n_frames, uint8_t* d_dst, ptrdiff_t dst_row_stride, ptrdiff_t dst_frame_stride, void* stream);
/* One host image; synchronous. */
int orbx_remap_image(orbx_remapper* r, const uint8_t* src, ptrdiff_t src_stride, uint8_t* dst, ptrdiff_t dst_stride);

/* ------------------------------------------------------------------------------------------ */
/* The Hamming consumers that follow extraction                                                */
/* ------------------------------------------------------------------------------------------ */
/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:483-548), the selection at :513-542, for a batch of map
 * points: group g holds the descriptors desc[group_ptr[g] .. group_ptr[g+1]) (the rows of the key frames that observe
 * the point, in the order of mObservations); best[g] = index inside the group of the descriptor with the least median
 * distance to the group (median = element (n-1)/2 of the sorted row that includes the zero distance to itself, :535;
 * the first index wins ties), -1 for an empty group (the reference returns early, :510-511). */
int orbx_distinctive_descriptors(orbx_matcher* m, const uint8_t* desc, const int32_t* group_ptr, int n_groups, int32_t* best);
/* Device-resident form; asynchronous on `stream`. */
int orbx_distinctive_descriptors_device(orbx_matcher* m, const uint8_t* d_desc, const int32_t* d_group_ptr, int n_groups,
                                        int64_t n_rows, int32_t* d_best, void* stream);

/* A DBoW2 vocabulary tree (ORBVocabulary, include/ORBVocabulary.h:52-53 = TemplatedVocabulary<FORB::TDescriptor, FORB>)
 * on plain arrays, as TemplatedVocabulary::loadFromTextFile reads it (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:
 * 1351-1437): node 0 is the root (its entries are ignored), node i >= 1 has parent[i] < i, a leaf flag, a 32-byte
 * descriptor and a weight; children keep node order and words number the leaves in node order.  scoring / weighting are
 * the ScoringType / WeightingType values of BowVector.h:36-53 (L1_NORM 0 .. DOT_PRODUCT 5; TF_IDF 0, TF 1, IDF 2,
 * BINARY 3).  The arrays are copied; the tree lives in device memory until destroy. */
typedef struct orbx_vocabulary orbx_vocabulary;
int orbx_vocabulary_create(int device, int k, int L, int scoring, int weighting, int n_nodes,
                           const int32_t* parent, const uint8_t* is_leaf, const uint8_t* descriptors, const double* weight,
                           orbx_vocabulary** out);
void orbx_vocabulary_destroy(orbx_vocabulary* v);
/* number of words (TemplatedVocabulary::size) */
int orbx_vocabulary_words(const orbx_vocabulary* v);
/* TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (:1230-1272) for n features: the descent
 * by least FORB::distance (first child wins ties).  d_word[i] = word id, d_node[i] = the node passed at level
 * L - levelsup (0 = root when that level is <= 0; the reference leaves it unset when the descent reaches a leaf above
 * that level: here it is then that leaf).  d_counts != NULL: the features are n_frames blocks of cap_per_frame rows of
 * which the first d_counts[f] are valid (the layout orbx_extract_batch_device leaves); other rows get -1.  With
 * d_counts == NULL all n_frames * cap_per_frame rows are features.  Asynchronous on `stream`. */
int orbx_vocabulary_descend_device(orbx_vocabulary* v, const uint8_t* d_desc, const int32_t* d_counts, int cap_per_frame,
                                   int n_frames, int levelsup, int32_t* d_word, int32_t* d_node, void* stream);
/* TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) (:1139-1207) as Frame::ComputeBoW and
 * KeyFrame::ComputeBoW call it (src/Frame.cc:395-402, levelsup 4): the descent on the device, then the BowVector
 * (addWeight / addIfNotExist by weighting type, normalisation by scoring type, BowVector.cpp:34-84; doubles, summed in
 * feature order) and the FeatureVector on the host.  bow_ids / bow_values (capacity n) come back ascending by word id;
 * fv_node_ids (capacity n), fv_node_ptr (n+1), fv_features (n) in the orbx_feature_vector layout. */
int orbx_vocabulary_transform(orbx_vocabulary* v, const uint8_t* desc, int n, int levelsup,
                              uint32_t* bow_ids, double* bow_values, int32_t* n_words,
                              uint32_t* fv_node_ids, int32_t* fv_node_ptr, int32_t* fv_features, int32_t* fv_n_nodes);

/* ------------------------------------------------------------------------------------------ */
/* Descriptor matrices in the reference's map files                                            */
/* ------------------------------------------------------------------------------------------ */
/* The cv::Mat payload the reference's Boost binary archives hold for KeyFrame::mDescriptors and MapPoint::mDescriptor
 * (save / load in include/MapPoint.h:466-501; archives opened with no_header in src/System.cc:385, 400): int cols,
 * int rows, size_t elemSize, size_t type, then cols*rows*elemSize raw bytes, all little-endian, 24 + data bytes.
 * Only this payload is covered: the class-tracking bytes Boost puts in front of the first cv::Mat of an archive depend
 * on the Boost version.  Host-only helpers, no device work. */
int64_t orbx_mat_record_bytes(int rows, int cols, int elem_size);
/* writes the record of a continuous rows x cols matrix; returns the bytes written or -1 when cap is too small */
int64_t orbx_mat_record_write(const uint8_t* data, int rows, int cols, int elem_size, int elem_type, uint8_t* out, int64_t cap);
/* parses a record; *data points into rec.  Returns the record's length or -1 for a truncated / inconsistent record */
int64_t orbx_mat_record_read(const uint8_t* rec, int64_t len, int* rows, int* cols, int* elem_size, int* elem_type, const uint8_t** data);

/* ------------------------------------------------------------------------------------------ */
/* Synthetic inputs for benchmarks and tests (no counterpart in the reference)                 */
/* ------------------------------------------------------------------------------------------ */
/* Frame generator: PCG32 seeded 0x5EED0000 + frame_index; W*H/200 random rectangles (side 6..48,
 * intensity 16..240) painted in order over mid-grey, then uniform noise in [-6,6]. */
void orbx_synth_frame(uint32_t frame_index, int w, int h, uint8_t* out, ptrdiff_t stride);
/* Stereo pair: the right image paints every rectangle shifted left by its own disparity 0..40. */
void orbx_synth_stereo(uint32_t frame_index, int w, int h, uint8_t* left, uint8_t* right, ptrdiff_t stride);
/* n iid uniform 256-bit descriptors from PCG32(seed). */
void orbx_synth_descriptors(uint32_t seed, int64_t n, uint8_t* out);
/* nq queries: map rows picked by PCG32(seed) with Binomial(256, 0.08) bit flips; src_idx receives the rows. */
void orbx_synth_queries(uint32_t seed, const uint8_t* map, int64_t n_map, int nq, uint8_t* out, int64_t* src_idx);
/* A synthetic vocabulary: the full k-ary tree of depth L in breadth-first numbering; a node's descriptor is its
 * parent's with PCG32(seed) bit flips (fewer with depth), leaves weigh 0.5 + (word id % 7).  Returns the node count
 * (k^(L+1)-1)/(k-1), the capacity the four arrays need; with parent == NULL only that count; -1 for bad k / L. */
int orbx_synth_vocabulary(uint32_t seed, int k, int L, int32_t* parent, uint8_t* is_leaf, uint8_t* descriptors, double* weight);

#ifdef __cplusplus
}
#endif
#endif
