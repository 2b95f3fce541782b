/*
 * orb_oracle.h -- C API of the CPU oracle (TEST INFRASTRUCTURE, not product).
 *
 * The oracle is a CPU restatement of the reference's ORB front-end
 * (src/ORBextractor.cc, the Hamming loops of src/ORBmatcher.cc, and the OpenCV
 * primitives they call).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product
 * (orb_slam3_detailed_comments_kor_amd/) never does.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for
 * this path and cannot be compiled here (it needs OpenCV, which is absent), so
 * this oracle is pinned only by closed-form known-answer tests and by its own
 * committed fixtures (see DESIGN.md "Oracle").
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orb_oracle_kp;

typedef struct orb_oracle orb_oracle;

/* trig modes for the steered-BRIEF rotation (reference src/ORBextractor.cc:111) */
enum { ORB_ORACLE_TRIG_LIBM = 0,   /* host libm cosf/sinf, what the reference calls   */
       ORB_ORACLE_TRIG_CR = 1 };   /* correctly-rounded float sin/cos (shared routine) */

orb_oracle* orb_oracle_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
void orb_oracle_destroy(orb_oracle*);
void orb_oracle_set_gauss_taps(orb_oracle*, const int* taps7);
void orb_oracle_set_trig_mode(orb_oracle*, int mode);
/* Timing-only fast path (never the checker of a parity test): SIMD prefilter in FAST (AVX2 builds) + a blur the compiler
 * vectorises; results identical to the scalar path.  Returns 1 when the build has the AVX2 prefilter. */
int orb_oracle_set_fastpath(orb_oracle*, int on);
/* Wall time accumulated per stage over the extract calls so far: pyramid, FAST cell loop, quadtree, orientation, blur,
 * descriptors (seconds); reset != 0 clears the accumulators. */
void orb_oracle_get_stage_seconds(orb_oracle*, double* s6, long* calls, int reset);
void orb_oracle_set_atan_fma(orb_oracle*, int on); /* fused Horner steps in fastAtan2 */

/* ORBextractor::operator() -- returns monoIndex (>=0), -1 for an empty image, -2 bad args/too small. */
int orb_oracle_extract(orb_oracle*, const uint8_t* img, int rows, int cols, size_t stride,
                       int lap0, int lap1, orb_oracle_kp* kps, uint8_t* desc, int cap, int* n_out);

/* CPU-baseline helper: nthreads threads x reps extractions, one extractor per thread. */
long orb_oracle_extract_many(int nthreads, int reps, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh,
                             const uint8_t* imgs, int nimg, int rows, int cols, int lap0, int lap1, double* seconds);
/* ... with the timing-only fast path on (fastpath != 0) and the per-stage wall seconds summed over the threads (stage_s6:
 * pyramid, FAST, quadtree, orientation, blur, descriptors; NULL = not wanted). */
long orb_oracle_extract_many2(int nthreads, int reps, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh,
                              const uint8_t* imgs, int nimg, int rows, int cols, int lap0, int lap1, int fastpath, double* seconds,
                              double* stage_s6);

/* tables (ctor) */
void orb_oracle_get_scale_tables(orb_oracle*, float* sf, float* inv, float* sigma2, float* inv_sigma2);
void orb_oracle_get_features_per_level(orb_oracle*, int* n_per_level);
void orb_oracle_get_umax(orb_oracle*, int* umax16);

/* stage taps, valid after orb_oracle_extract() */
int orb_oracle_get_level(orb_oracle*, int level, const uint8_t** data /* padded buffer */, int* rows, int* cols,
                         size_t* stride); /* rows/cols of the padded buffer = level + 38 */
int orb_oracle_get_blurred(orb_oracle*, int level, const uint8_t** data, int* rows, int* cols, size_t* stride);
int orb_oracle_get_candidates(orb_oracle*, int level, const orb_oracle_kp** kps);      /* vToDistributeKeys  */
int orb_oracle_get_level_keypoints(orb_oracle*, int level, const orb_oracle_kp** kps); /* allKeypoints[level] */

/* OpenCV primitives as restated */
void orb_oracle_resize_linear(const uint8_t* src, int sh, int sw, size_t sstride, uint8_t* dst, int dh, int dw,
                              size_t dstride);
void orb_oracle_border_reflect101(uint8_t* buf, int rows, int cols, size_t stride, int border);
int orb_oracle_fast(const uint8_t* img, int rows, int cols, size_t stride, int threshold, int nms,
                    orb_oracle_kp* out, int cap);
int orb_oracle_fast_score_closed(const uint8_t* center, size_t stride);
int orb_oracle_fast_score_2loop(const uint8_t* center, size_t stride, int threshold);
void orb_oracle_gaussian_blur7(const uint8_t* src, int rows, int cols, size_t sstride, uint8_t* dst, size_t dstride,
                               const int* taps7);
float orb_oracle_fast_atan2(float y, float x);
float orb_oracle_fast_atan2_fma(float y, float x);
void orb_oracle_sincos_cr(float angle_rad, float* s, float* c);
int orb_oracle_distribute_octree(const orb_oracle_kp* cands, int n, int minX, int maxX, int minY, int maxY, int N,
                                 orb_oracle_kp* out, int cap);

/* matcher (src/ORBmatcher.cc) */
int orb_oracle_descriptor_distance(const uint8_t* a, const uint8_t* b);
void orb_oracle_hamming_matrix(const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* D);
void orb_oracle_bfknn2(const uint8_t* Q, int nQ, const uint8_t* T, int nT, int32_t* idx, int32_t* dist);
void orb_oracle_three_maxima(const int* histo_counts, int L, int* ind1, int* ind2, int* ind3);

/* CSR FeatureVector: node_ids[nn] ascending, offsets[nn+1], indices[offsets[nn]] */
typedef struct {
    int nn;
    const uint32_t* node_ids;
    const int32_t* offsets;
    const int32_t* indices;
} orb_oracle_fv;

/* SearchByBoW(KeyFrame*,Frame&) src/ORBmatcher.cc:269-471.
 * match[N_F] = KF feature index or -1.  Returns nmatches. */
int orb_oracle_search_bow_kf_f(const uint8_t* descKF, int nKF, const uint8_t* maskKF /*1=good MapPoint*/,
                               const float* angKF, const orb_oracle_fv* fvKF, const uint8_t* descF, int nF,
                               const float* angF, const orb_oracle_fv* fvF, int Nleft /* -1 = mono */,
                               float nnratio, int checkOri, int32_t* match);
/* SearchByBoW(KeyFrame*,KeyFrame*) src/ORBmatcher.cc:823-963.  match12[n1] = idx2 or -1. */
int orb_oracle_search_bow_kf_kf(const uint8_t* desc1, int n1, const uint8_t* mask1, const float* ang1,
                                const orb_oracle_fv* fv1, int lim1 /* -1 or mvKeysUn.size() */, const uint8_t* desc2,
                                int n2, const uint8_t* mask2, const float* ang2, const orb_oracle_fv* fv2, int lim2,
                                float nnratio, int checkOri, int32_t* match12);
/* SearchForTriangulation_ src/ORBmatcher.cc:1208-1449, monocular/rectified pinhole case
 * (no second camera).  kp = {x,y} pairs, uRight<0 => monocular feature.
 * F12 as Pinhole::epipolarConstrain_ would form it (src/CameraModels/Pinhole.cpp:159-181).
 * pairs[2*k] = idx1, pairs[2*k+1] = idx2, sorted by idx1.  Returns npairs. */
int orb_oracle_search_triangulation(const uint8_t* desc1, int n1, const uint8_t* hasMP1, const float* kp1xy,
                                    const float* ang1, const int32_t* oct1, const float* uRight1,
                                    const orb_oracle_fv* fv1, const uint8_t* desc2, int n2, const uint8_t* hasMP2,
                                    const float* kp2xy, const float* ang2, const int32_t* oct2, const float* uRight2,
                                    const orb_oracle_fv* fv2, const float* F12 /*9, row-major*/, float epx, float epy,
                                    const float* scaleFactors2, const float* levelSigma2_2, int bOnlyStereo,
                                    int bCoarse, int checkOri, int32_t* pairs);
/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:706-821) over flattened frames; vnMatches12[n1] = index
 * into F2 or -1; returns nmatches.  (vbPrevMatched is updated by the caller from the result, :813-816.) */
typedef struct {
    const uint8_t* desc1; int n1; const int32_t* octave1; const float* angle1; const float* prev_xy;
    const uint8_t* desc2; int n2; const float* kx2; const float* ky2; const int32_t* octave2; const float* angle2;
    float minX, minY, gridWInv, gridHInv;
    int window_size; float nnratio; int check_orientation;
} orb_oracle_init_args;
int orb_oracle_search_initialization(const orb_oracle_init_args* a, int32_t* vnMatches12);
/* KannalaBrandt8::TriangulateMatches_ (src/CameraModels/KannalaBrandt8.cpp:409-480): depth of the triangulated
 * point in camera 1, or -1 (low parallax, behind a camera, reprojection error).  P = fx,fy,cx,cy,k0..k3;
 * R12 row-major 3x3, t12; p3D may be NULL.  epipolarConstrain_ (:239-242) is `> 0.0001f`. */
float orb_oracle_kb8_triangulate_matches(const float* P1, const float* P2, const float* kp1xy, const float* kp2xy,
                                         const float* R12, const float* t12, float sigmaLevel, float unc, float* p3D);
/* Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1119-1159) on the lapping-area slices; returns nMatches. */
int orb_oracle_stereo_fisheye_matches(const uint8_t* descL, const float* kpL_xy, const int32_t* octL, int nL,
                                      const uint8_t* descR, const float* kpR_xy, const int32_t* octR, int nR,
                                      const float* P1, const float* P2, const float* Rlr, const float* tlr,
                                      const float* levelSigma2, int32_t* leftToRight, int32_t* rightToLeft,
                                      float* depth, float* p3D);
/* SearchForTriangulation_ with the KannalaBrandt8 gate (fisheye monocular pair or two-camera rig). */
typedef struct {
    const uint8_t* desc1; int n1; const uint8_t* hasMP1; const float* kp1xy; const float* ang1; const int32_t* oct1;
    const float* uRight1; const orb_oracle_fv* fv1; int Nleft1;
    const uint8_t* desc2; int n2; const uint8_t* hasMP2; const float* kp2xy; const float* ang2; const int32_t* oct2;
    const float* uRight2; const orb_oracle_fv* fv2; int Nleft2;
    const float* kb8_1L; const float* kb8_1R; const float* kb8_2L; const float* kb8_2R; /* 8 floats each; R: rigs */
    const float* R12; const float* t12;   /* 4 x 9 and 4 x 3: ll, lr, rl, rr (:1238-1248); only [0] without a rig */
    float ep[2];
    const float* scaleFactors2; const float* levelSigma2_1; const float* levelSigma2_2;
    int only_stereo, coarse, check_orientation;
} orb_oracle_tri_kb8_args;
int orb_oracle_search_triangulation_kb8(const orb_oracle_tri_kb8_args* a, int32_t* pairs);
/* KannalaBrandt8::matchAndtriangulate (src/CameraModels/KannalaBrandt8.cpp:244-335, with Triangulate :498-512):
 * 1 and the triangulated world point, or 0.  Tcw = rows 0..2 of the camera pose (3x4 row-major). */
int orb_oracle_kb8_match_and_triangulate(const float* P1, const float* P2, const float* kp1xy, const float* kp2xy,
                                         const float* Tcw1, const float* Tcw2, float sigmaLevel1, float sigmaLevel2,
                                         float* x3D);
/* The SearchForTriangulation overload that also returns the triangulated points (src/ORBmatcher.cc:1452-1641; no
 * caller in the reference).  Its gate is pCamera1->matchAndtriangulate: Pinhole's returns false (include/
 * CameraModels/Pinhole.h:88-91), so kb8_1L == NULL (pinhole camera 1) yields no pairs.  Poses are 3x4 row-major
 * (GetPose / GetRightPose); points[3*k..] belongs to pairs[2*k..]. */
typedef struct {
    const uint8_t* desc1; int n1; const uint8_t* hasMP1; const float* kp1xy; const float* ang1; const int32_t* oct1;
    const orb_oracle_fv* fv1; int Nleft1;
    const uint8_t* desc2; int n2; const uint8_t* hasMP2; const float* kp2xy; const float* ang2; const int32_t* oct2;
    const orb_oracle_fv* fv2; int Nleft2;
    const float* kb8_1L; const float* kb8_1R; const float* kb8_2L; const float* kb8_2R;
    const float* Tcw1L; const float* Tcw1R; const float* Tcw2L; const float* Tcw2R;
    const float* levelSigma2_1; const float* levelSigma2_2;
    int check_orientation;
} orb_oracle_tri3d_args;
int orb_oracle_search_triangulation_3d(const orb_oracle_tri3d_args* a, int32_t* pairs, float* points);
/* Inner loops of ORBmatcher::SearchByProjection (src/ORBmatcher.cc:44-197 mode 0, :2193-2419 / :2421-2541
 * mode 1) over flattened inputs; same layout as orbfe_proj_args (include/orbfe.h).  Mode 1 also covers the
 * Sim3 overloads (:473-586, :588-704), and with qblocks all zero (independent queries) Fuse (:1643-1841 with
 * chi2_gate, :1843-1965 without) and SearchBySim3's two directions (:1967-2191). */
typedef struct {
    const uint8_t* desc; int n;
    const float* kx; const float* ky; const int32_t* octave; const float* angle;
    const float* uright; const uint8_t* taken;
    int Nleft; const int32_t* left_to_right; const int32_t* right_to_left;
    float minX, minY, gridWInv, gridHInv;
    int nq; const uint8_t* qdesc;
    const float* qx; const float* qy; const float* qr;
    const int32_t* qmin_level; const int32_t* qmax_level;
    const float* qxr; const uint8_t* qflags; const float* qangle; const uint8_t* qblocks;
    int mode; float nnratio; int th_high; int check_orientation;
    const float* inv_level_sigma2; int n_levels; int chi2_gate; /* Fuse's per-candidate reprojection test, src/ORBmatcher.cc:1773-1799 */
} orb_oracle_proj_args;
int orb_oracle_search_projection(const orb_oracle_proj_args* a, int32_t* q_match, int32_t* feat_match);
/* MapPoint::ComputeDistinctiveDescriptors src/MapPoint.cc:387-419 for npts points with pooled descriptors. */
void orb_oracle_distinctive_descriptors(const uint8_t* pool, const int32_t* offsets, int npts, int32_t* best);
/* DBoW2 TemplatedVocabulary::transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1259) per feature:
 * leaf word id, leaf weight and the node id `levelsup` levels above the leaves.  Tree in CSR form:
 * children of node i = child_ids[child_off[i] .. child_off[i+1]) in stored order; node 0 is the root. */
void orb_oracle_vocab_transform(int nnodes, const uint8_t* node_desc, const int32_t* child_off, const int32_t* child_ids,
                                const int32_t* node_word, const double* node_weight, int L, const uint8_t* feats, int n,
                                int levelsup, int32_t* word_id, int32_t* node_id, double* weight);
/* TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) (TemplatedVocabulary.h:1127-1192) -- what
 * Frame::ComputeBoW (src/Frame.cc:724-731) and KeyFrame::ComputeBoW (src/KeyFrame.cc:105-114) call with levelsup = 4 -- with
 * BowVector::addWeight / addIfNotExist / normalize (BowVector.cpp:34-86) and FeatureVector::addFeature (FeatureVector.cpp:31-45)
 * done on real std::maps.  weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; scoring: 0 L1_NORM .. 5 DOT_PRODUCT (BowVector.h:39-56;
 * ORBvoc.txt is "10 6 0 0").  Outputs: the BowVector as (word id, value) in map order (ascending id), the FeatureVector as CSR
 * (node ids ascending, offsets[nn + 1], feature indices in push_back order); every array sized n (offsets n + 1).
 * Returns the number of words; *nn_out = number of nodes. */
int orb_oracle_compute_bow(int nnodes, const uint8_t* node_desc, const int32_t* child_off, const int32_t* child_ids,
                           const int32_t* node_word, const double* node_weight, int L, const uint8_t* feats, int n, int levelsup,
                           int weighting, int scoring, uint32_t* bow_ids, double* bow_vals, uint32_t* node_ids, int32_t* offsets,
                           int32_t* indices, int* nn_out);
/* Frame::ComputeStereoMatches src/Frame.cc:797-967 (rectified stereo).  L, R = the oracle extractors that
 * processed the left / right image.  uRight/depth sized N (left keypoints), -1 = no match.  Returns #matches. */
int orb_oracle_compute_stereo_matches(orb_oracle* L, orb_oracle* R, const orb_oracle_kp* kpsL, const uint8_t* descL,
                                      int N, const orb_oracle_kp* kpsR, const uint8_t* descR, int Nr, float mb,
                                      float mbf, float* uRight, float* depth);
/* KannalaBrandt8::unproject src/CameraModels/KannalaBrandt8.cpp:96-123; params = fx,fy,cx,cy,k0..k3 */
void orb_oracle_kb8_unproject(const float* params8, const float* uv, int n, float* rays3);
/* cv::SVD::compute(A, w, u, vt) of a 4x4 CV_32F matrix: vt only (KannalaBrandt8::Triangulate_, KannalaBrandt8.cpp:514-535) */
void orb_oracle_svd_vt_4x4(const float* A16, float* vt16);

#ifdef __cplusplus
}
#endif
#endif
