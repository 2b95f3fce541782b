/*
 * adapters/ORBmatcher.h -- ORB_SLAM3::ORBmatcher with its reference signatures (include/ORBmatcher.h:39-97),
 * its Hamming loops served by liborbfe.so through the C ABI (include/orbfe.h).
 *
 * The methods keep what only the host can do -- the walk over the KeyFrame / Frame / MapPoint object graph, the
 * projections, the write-back into vpMapPointMatches / mvpMapPoints / vMatchedPairs -- and hand the inner loops
 * (every DescriptorDistance, best / second-best, ratio and orientation tests) to the device in ONE call each:
 *
 *   SearchByBoW(KeyFrame*, Frame&, ...)                 src/ORBmatcher.cc:269-471    orbfe_search_bow, variant 0
 *   SearchByBoW(KeyFrame*, KeyFrame*, ...)              src/ORBmatcher.cc:823-963    orbfe_search_bow, variant 1
 *   SearchForTriangulation_(KF1, KF2, F12, ...)         src/ORBmatcher.cc:1208-1449  orbfe_search_tri (pinhole gate) /
 *                                                                                    orbfe_search_tri_kb8 (fisheye, rigs)
 *   SearchByProjection(Frame&, vpMapPoints, th, ...)    src/ORBmatcher.cc:44-197     orbfe_search_projection, mode 0
 *   SearchByProjection(CurrentFrame, LastFrame, ...)    src/ORBmatcher.cc:2193-2419  orbfe_search_projection, mode 1
 *   Fuse(KeyFrame*, vpMapPoints, th, bRight)            src/ORBmatcher.cc:1643-1841  orbfe_search_projection, mode 1 + chi2
 *   SearchByProjection(CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist)  :2421-2541  mode 1 (relocalisation)
 *   SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratio) and its vpMatchedKF twin  :473-704  mode 1
 *   Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)        src/ORBmatcher.cc:1843-1965  mode 1, independent queries
 *   SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)  :1967-2191           mode 1, one call per direction
 *   SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)  :706-821  orbfe_search_initialization
 *   SearchForTriangulation(KF1, KF2, cv::Mat F12, ...)  src/ORBmatcher.cc:965-1206  = SearchForTriangulation_
 *   SearchForTriangulation(KF1, KF2, F12, ..., vMatchedPoints)  :1452-1641  orbfe_search_tri_3d (no caller in the
 *                                                                 reference; KannalaBrandt8::matchAndtriangulate gate)
 *   DescriptorDistance                                  src/ORBmatcher.cc:2591-2607  (host, one pair: a call per pair
 *                                                                                     would cost more than it computes)
 *
 * Inside ORB-SLAM3 include the real Frame.h / KeyFrame.h / MapPoint.h before this header (and drop src/ORBmatcher.cc
 * from the build); here, where OpenCV / Eigen / DBoW2 are absent, adapters/orbslam_standins.h provides classes with
 * the same member names, and adapters/test_matcher_adapter.cpp drives every method through the C ABI against the
 * oracle (tests/test_gpu_matcher_adapter.py).  Two-camera (fisheye) rigs -- Nleft / NLeft != -1, mpCamera2 -- are
 * handled wherever the reference handles them: both SearchByBoW, both frame-side SearchByProjection overloads (right-
 * camera queries into the right grid), Fuse(..., bRight), and SearchForTriangulation_ with the KannalaBrandt8 gate
 * (orbfe_search_tri_kb8: monocular fisheye pair or the four relative poses of a rig).
 */
#ifndef ORBFE_ADAPTER_ORBMATCHER_H
#define ORBFE_ADAPTER_ORBMATCHER_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "../include/orbfe.h"
#include "cv_standins.h"
#ifndef ORBFE_HAVE_ORBSLAM
#include "orbslam_standins.h"
#endif

namespace ORB_SLAM3 {

namespace orbfe_adapter {

// DBoW2::FeatureVector (std::map<NodeId, vector<unsigned>>) -> CSR.  Build once per Frame / KeyFrame and keep it
// next to mFeatVec in a real integration; the methods below rebuild it per call to stay drop-in.
struct CSR {
    std::vector<uint32_t> ids;
    std::vector<int32_t> off, ind;
    orbfe_fv view() const
    {
        orbfe_fv f;
        f.nn = (int)ids.size();
        f.node_ids = ids.data();
        f.offsets = off.data();
        f.indices = ind.data();
        return f;
    }
};
inline CSR toCSR(const DBoW2::FeatureVector& fv)
{
    CSR c;
    c.off.push_back(0);
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        c.ids.push_back((uint32_t)it->first);
        for (size_t k = 0; k < it->second.size(); k++) c.ind.push_back((int32_t)it->second[k]);
        c.off.push_back((int32_t)c.ind.size());
    }
    return c;
}

// rows of a CV_8U descriptor matrix as one dense n x 32 block (cv::Mat rows of 32 bytes are contiguous unless the
// matrix is a view: then they are gathered)
inline const uint8_t* dense_descriptors(const cv::Mat& D, int n, std::vector<uint8_t>& tmp)
{
    if (n == 0) return nullptr;
    if ((size_t)D.step == 32) return D.data;
    tmp.resize((size_t)n * 32);
    for (int i = 0; i < n; i++) std::memcpy(tmp.data() + (size_t)i * 32, D.data + (size_t)i * D.step, 32);
    return tmp.data();
}

// Frame handles (orbfe_frame, include/orbfe.h): Tracking searches the SAME Frame two or three times per image -- the last
// frame at th, again at 2 th when that finds too little (src/Tracking.cc:2817-2827), then the local map (:2927) -- and every
// search used to flatten and upload the frame side (descriptors, keypoints, grid) again.  The adapter keeps the handles of
// the last few Frames it has seen, per calling thread (Tracking is the one thread that searches Frames; a thread-local
// table needs no lock and no handle is ever freed under another thread's search).  A Frame is recognised by its address,
// its mnId (Frame.cc: mnId = nNextId++, a new value per image although mCurrentFrame keeps its address), its feature
// count and descriptor buffer; its features never change after construction, its map points do and travel per search.
struct FrameHandles {
    struct Entry {
        const void* obj = nullptr;
        unsigned long id = 0;
        int n = 0, device = 0;
        const void* desc = nullptr;
        orbfe_frame* h = nullptr;
        unsigned long used = 0;
    };
    Entry e[4];
    unsigned long clock = 0;
    long creates = 0, hits = 0; // (what the tests look at)
    ~FrameHandles()
    {
        for (Entry& x : e)
            if (x.h) orbfe_frame_destroy(x.h);
    }
    Entry* find(const void* obj, unsigned long id, int n, const void* desc, int device)
    {
        for (Entry& x : e)
            if (x.h && x.obj == obj && x.id == id && x.n == n && x.desc == desc && x.device == device) {
                x.used = ++clock;
                hits++;
                return &x;
            }
        return nullptr;
    }
    Entry* slot()
    { // an empty entry, else the least recently used one
        Entry* best = &e[0];
        for (Entry& x : e) {
            if (!x.h) return &x;
            if (x.used < best->used) best = &x;
        }
        orbfe_frame_destroy(best->h);
        best->h = nullptr;
        return best;
    }
};
inline FrameHandles& frame_handles()
{
    static thread_local FrameHandles f;
    return f;
}
inline bool& use_frame_handles()
{
    static bool on = true; // (tests switch it off to compare the two paths)
    return on;
}

// Keyframe handles (orbfe_keyframe, include/orbfe.h; round 4): SearchByBoW and SearchForTriangulation_ run one current
// frame or keyframe against keyframes whose descriptors, keypoints and FeatureVector never change after construction, and
// used to flatten and upload both sides on every call.  The adapter keeps a handle per KeyFrame it has searched -- process
// wide, because Tracking (relocalisation), LocalMapping and LoopClosing search the same keyframes from three threads: a
// mutex guards the table, an entry is reference-counted so that an eviction cannot free a handle under another thread's
// search.  What DOES change -- which features hold a MapPoint -- travels with every call (mask1 / mask2 / hasMP arguments of
// the handle searches), never through the shared handle.  A KeyFrame is recognised by its address, mnId, feature count and
// descriptor buffer; at most 512 handles (~60 KB of device memory each), least recently used out.
struct KeyFrameHandles {
    struct Handle {
        orbfe_keyframe* h = nullptr;
        bool tri = false; // created with keypoints / octaves / mvuRight
        ~Handle()
        {
            if (h) orbfe_keyframe_destroy(h);
        }
    };
    struct Entry {
        const void* obj = nullptr;
        unsigned long id = 0;
        int n = 0, device = 0;
        const void* desc = nullptr;
        size_t fvNodes = 0, fvFeat = 0; // the FeatureVector the handle was built from (nodes, indexed features)
        std::shared_ptr<Handle> h;
        unsigned long used = 0;
    };
    std::mutex m;
    std::vector<Entry> e;
    unsigned long clock = 0;
    long creates = 0, hits = 0; // (what the tests look at)
    // (i) a KeyFrame's mFeatVec is EMPTY between its construction from a frame the motion model tracked
    // (KeyFrame(F, ...) copies F.mFeatVec, src/KeyFrame.cc) and LocalMapping's ComputeBoW (src/LocalMapping.cc:380), while
    // Tracking already searches it as mpReferenceKF (src/Tracking.cc:3290): no handle is made or cached in that window -- the
    // caller takes the per-call path, which reads the vector as it is, like the reference -- and the vector's size is part of
    // an entry's key, so a handle never outlives the vector it was built from; (ii) handles are created and destroyed OUTSIDE
    // the table's lock (a creation uploads and waits for its stream, a destruction waits for the handle's users): Tracking,
    // LocalMapping and LoopClosing only ever contend for the table walk.
    template <class KF>
    std::shared_ptr<Handle> get(KF* pKF, int device)
    {
        const int n = pKF->N;
        const size_t fvNodes = pKF->mFeatVec.size();
        if (fvNodes == 0 || n < 1 || pKF->mDescriptors.rows < n) return nullptr;
        size_t fvFeat = 0;
        for (DBoW2::FeatureVector::const_iterator it = pKF->mFeatVec.begin(); it != pKF->mFeatVec.end(); ++it) fvFeat += it->second.size();
        auto same = [&](const Entry& x) {
            return x.h && x.obj == pKF && x.id == pKF->mnId && x.n == n && x.desc == pKF->mDescriptors.data && x.device == device &&
                   x.fvNodes == fvNodes && x.fvFeat == fvFeat;
        };
        {
            std::lock_guard<std::mutex> lock(m);
            for (Entry& x : e)
                if (same(x)) {
                    x.used = ++clock;
                    hits++;
                    return x.h;
                }
        }
        // flatten once: what rig_keypoint reads (mvKeysUn without a second camera, else mvKeys / mvKeysRight)
        std::vector<uint8_t> tmp, mask((size_t)n, 0);
        std::vector<float> xy(2 * (size_t)n), ang((size_t)n);
        std::vector<int32_t> oct((size_t)n);
        for (int i = 0; i < n; i++) {
            const cv::KeyPoint& kp = pKF->NLeft == -1 ? pKF->mvKeysUn[i]
                                                      : (i < pKF->NLeft ? pKF->mvKeys[i] : pKF->mvKeysRight[i - pKF->NLeft]);
            xy[2 * (size_t)i] = kp.pt.x;
            xy[2 * (size_t)i + 1] = kp.pt.y;
            ang[i] = kp.angle;
            oct[i] = kp.octave;
        }
        const CSR c = toCSR(pKF->mFeatVec);
        orbfe_keyframe_args a;
        std::memset(&a, 0, sizeof a);
        a.desc = dense_descriptors(pKF->mDescriptors, n, tmp);
        a.n = n;
        a.mask = mask.data(); // (every search sends its own flags)
        a.angle = ang.data();
        const bool tri = (int)pKF->mvuRight.size() >= n;
        if (tri) {
            a.kp_xy = xy.data();
            a.octave = oct.data();
            a.uRight = pKF->mvuRight.data();
        }
        a.fv = c.view();
        std::shared_ptr<Handle> H = std::make_shared<Handle>();
        if (orbfe_keyframe_create(&H->h, device, &a) < 0) return nullptr;
        H->tri = tri;
        std::shared_ptr<Handle> evicted; // (released after the lock: its destructor may wait for the device)
        {
            std::lock_guard<std::mutex> lock(m);
            for (Entry& x : e)
                if (same(x)) { // another thread made it meanwhile: use theirs, ours dies behind the lock
                    x.used = ++clock;
                    hits++;
                    evicted = std::move(H);
                    H = x.h;
                    break;
                }
            if (!evicted) {
                creates++;
                Entry* slot = nullptr;
                for (Entry& x : e) // a stale entry of the same KeyFrame (its FeatureVector has been computed since) goes first
                    if (x.obj == pKF && x.id == pKF->mnId && x.device == device) slot = &x;
                if (!slot && e.size() < 512) {
                    e.emplace_back();
                    slot = &e.back();
                } else if (!slot) {
                    slot = &e[0];
                    for (Entry& x : e)
                        if (x.used < slot->used) slot = &x;
                }
                evicted = std::move(slot->h); // (dies with its last user, and not under the lock)
                slot->obj = pKF;
                slot->id = pKF->mnId;
                slot->n = n;
                slot->device = device;
                slot->desc = pKF->mDescriptors.data;
                slot->fvNodes = fvNodes;
                slot->fvFeat = fvFeat;
                slot->h = H;
                slot->used = ++clock;
            }
        }
        return H;
    }
    void clear()
    {
        std::vector<Entry> dead;
        {
            std::lock_guard<std::mutex> lock(m);
            dead.swap(e);
        }
    }
};
inline KeyFrameHandles& keyframe_handles()
{
    static KeyFrameHandles k;
    return k;
}
inline bool& use_keyframe_handles()
{
    static bool on = true; // (tests switch it off to compare the two paths)
    return on;
}

// the keypoint a method reads for feature idx of a keyframe / frame with an optional second camera
// (e.g. :377-381, :1278-1283): mvKeysUn without a rig, else mvKeys / mvKeysRight
template <class T>
inline const cv::KeyPoint& rig_keypoint(const T& f, int nLeft, size_t idx)
{
    return nLeft == -1 ? f.mvKeysUn[idx] : ((int)idx < nLeft ? f.mvKeys[idx] : f.mvKeysRight[idx - nLeft]);
}

#ifdef ORBFE_HAVE_OPENCV_MATS // inside ORB-SLAM3: poses and positions are cv::Mat (CV_32F)
inline void frame_pose(const Frame& F, float R[9], float t[3])
{
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R[3 * i + j] = F.mTcw.at<float>(i, j);
        t[i] = F.mTcw.at<float>(i, 3);
    }
}
inline void world_pos(MapPoint* p, float x[3])
{
    const cv::Mat w = p->GetWorldPos();
    for (int i = 0; i < 3; i++) x[i] = w.at<float>(i);
}
inline void normal_of(MapPoint* p, float x[3])
{
    const cv::Mat w = p->GetNormal();
    for (int i = 0; i < 3; i++) x[i] = w.at<float>(i);
}
inline void kf_pose(KeyFrame* pKF, bool bRight, float R[9], float t[3], float O[3])
{
    const cv::Mat Rm = bRight ? pKF->GetRightRotation() : pKF->GetRotation();
    const cv::Mat tm = bRight ? pKF->GetRightTranslation() : pKF->GetTranslation();
    const cv::Mat Om = bRight ? pKF->GetRightCameraCenter() : pKF->GetCameraCenter();
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R[3 * i + j] = Rm.at<float>(i, j);
        t[i] = tm.at<float>(i);
        O[i] = Om.at<float>(i);
    }
}
#else
inline void frame_pose(const Frame& F, float R[9], float t[3])
{
    for (int i = 0; i < 9; i++) R[i] = F.mRcw_.val[i];
    for (int i = 0; i < 3; i++) t[i] = F.mtcw_.val[i];
}
inline void world_pos(MapPoint* p, float x[3])
{
    const cv::Matx31f w = p->GetWorldPos_();
    for (int i = 0; i < 3; i++) x[i] = w(i);
}
inline void normal_of(MapPoint* p, float x[3])
{
    const cv::Matx31f w = p->GetNormal_();
    for (int i = 0; i < 3; i++) x[i] = w(i);
}
inline void kf_pose(KeyFrame* pKF, bool bRight, float R[9], float t[3], float O[3])
{
    const cv::Matx33f Rm = bRight ? pKF->GetRightRotation_() : pKF->GetRotation_();
    const cv::Matx31f tm = bRight ? pKF->GetRightTranslation_() : pKF->GetTranslation_();
    const cv::Matx31f Om = bRight ? pKF->GetRightCameraCenter_() : pKF->GetCameraCenter_();
    for (int i = 0; i < 9; i++) R[i] = Rm.val[i];
    for (int i = 0; i < 3; i++) {
        t[i] = tm(i);
        O[i] = Om(i);
    }
}
#endif

// rotation / translation of the 3x4 CV_32F Frame::mTrl (left camera -> right camera, include/Frame.h)
inline void frame_trl(const Frame& F, float R[9], float t[3])
{
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R[3 * i + j] = F.mTrl.at<float>(i, j);
        t[i] = F.mTrl.at<float>(i, 3);
    }
}

// x_c = R x_w + t as cv::Mat evaluates `Rcw*x3Dw+tcw` for CV_32F (one gemm with double accumulators, rounded once)
inline void transform(const float R[9], const float t[3], const float xw[3], float xc[3])
{
    for (int i = 0; i < 3; i++)
        xc[i] = (float)((double)R[3 * i] * xw[0] + (double)R[3 * i + 1] * xw[1] + (double)R[3 * i + 2] * xw[2] + (double)t[i]);
}

// Scw (4x4, CV_32F) -> Rcw, tcw, Ow as :479-486 / :1850-1856 form them with cv::Mat expressions: sRcw / scw and
// tcw / scw are scalings by the double 1/scw rounded to float, Ow = -Rcw^T tcw one gemm
inline void decompose_sim3(const cv::Mat& Scw, float Rcw[9], float tcw[3], float Ow[3])
{
    double d = 0;
    for (int j = 0; j < 3; j++) d += (double)Scw.at<float>(0, j) * (double)Scw.at<float>(0, j);
    const float scw = (float)std::sqrt(d);
    const double inv = 1.0 / (double)scw;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Rcw[3 * i + j] = (float)((double)Scw.at<float>(i, j) * inv);
        tcw[i] = (float)((double)Scw.at<float>(i, 3) * inv);
    }
    for (int i = 0; i < 3; i++)
        Ow[i] = (float)(-((double)Rcw[i] * tcw[0] + (double)Rcw[3 + i] * tcw[1] + (double)Rcw[6 + i] * tcw[2]));
}
inline float norm3(const float v[3])
{
    return (float)std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); // cv::norm (L2, double)
}

// queries of one orbfe_search_projection call, appended in the reference's loop order
struct Queries {
    std::vector<uint8_t> desc, flags, blocks;
    std::vector<float> x, y, r, xr, angle;
    std::vector<int32_t> minLevel, maxLevel;
    std::vector<MapPoint*> mp;
    std::vector<int> src; // index of the point in the caller's vector (set by the caller when it needs it)
    void push(MapPoint* p, const uint8_t* d, float qx, float qy, float qr, int lo, int hi, float qxr, int fl, float ang, int blk)
    {
        mp.push_back(p);
        desc.insert(desc.end(), d, d + 32);
        x.push_back(qx);
        y.push_back(qy);
        r.push_back(qr);
        minLevel.push_back(lo);
        maxLevel.push_back(hi);
        xr.push_back(qxr);
        flags.push_back((uint8_t)fl);
        angle.push_back(ang);
        blocks.push_back((uint8_t)blk);
    }
    int size() const { return (int)mp.size(); }
};

// the frame side of orbfe_proj_args: the keypoints GetFeaturesInArea reads (mvKeysUn without a rig, else
// mvKeys ++ mvKeysRight), split into the arrays the shim takes
struct FeatureArrays {
    std::vector<float> kx, ky, angle;
    std::vector<int32_t> octave;
    template <class T>
    void fill(const T& f, int n, int nLeft)
    {
        kx.resize(n);
        ky.resize(n);
        angle.resize(n);
        octave.resize(n);
        for (int i = 0; i < n; i++) {
            const cv::KeyPoint& kp = rig_keypoint(f, nLeft, (size_t)i);
            kx[i] = kp.pt.x;
            ky[i] = kp.pt.y;
            angle[i] = kp.angle;
            octave[i] = kp.octave;
        }
    }
};

} // namespace orbfe_adapter

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri), mDevice(device)
    {
    }

    // Computes the Hamming distance between two ORB descriptors (src/ORBmatcher.cc:2591-2607)
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b)
    {
        const uint32_t* pa = reinterpret_cast<const uint32_t*>(a.data);
        const uint32_t* pb = reinterpret_cast<const uint32_t*>(b.data);
        int dist = 0;
        for (int i = 0; i < 8; i++) dist += __builtin_popcount(pa[i] ^ pb[i]);
        return dist;
    }

    // ---- src/ORBmatcher.cc:269-471
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches)
    {
        using namespace orbfe_adapter;
        const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
        const int n1 = (int)vpMapPointsKF.size();
        std::vector<uint8_t> good(n1), tmp1, tmp2;
        std::vector<float> ang1(n1), ang2(F.N);
        for (int i = 0; i < n1; i++) { // :301-307: only features with a good MapPoint search
            MapPoint* pMP = vpMapPointsKF[i];
            good[i] = (pMP && !pMP->isBad()) ? 1 : 0;
            // :377-381: the keyframe keypoint is mvKeysUn without a second camera
            ang1[i] = (!pKF->mpCamera2) ? pKF->mvKeysUn[i].angle
                                        : (i >= pKF->NLeft ? pKF->mvKeysRight[i - pKF->NLeft].angle : pKF->mvKeys[i].angle);
        }
        for (int i = 0; i < F.N; i++) // :385-388 / :414-417: mvKeys, or mvKeysRight for the right half of a rig
            ang2[i] = (F.Nleft == -1 || !F.mpCamera2 || i < F.Nleft) ? F.mvKeys[i].angle : F.mvKeysRight[i - F.Nleft].angle;
        const CSR c1 = toCSR(pKF->mFeatVec), c2 = toCSR(F.mFeatVec);
        orbfe_bow_args a;
        std::memset(&a, 0, sizeof(a));
        a.desc1 = dense_descriptors(pKF->mDescriptors, n1, tmp1);
        a.n1 = n1;
        a.mask1 = good.data();
        a.angle1 = ang1.data();
        a.fv1 = c1.view();
        a.limit1 = -1;
        a.desc2 = dense_descriptors(F.mDescriptors, F.N, tmp2);
        a.n2 = F.N;
        a.mask2 = nullptr;
        a.angle2 = ang2.data();
        a.fv2 = c2.view();
        a.limit2 = -1;
        a.Nleft = F.Nleft;
        a.nnratio = mfNNratio;
        a.check_orientation = mbCheckOrientation ? 1 : 0;
        a.variant = 0;
        std::vector<int32_t> match((size_t)std::max(F.N, 1), -1);
        int nmatches;
        std::shared_ptr<KeyFrameHandles::Handle> H =
            use_keyframe_handles() && n1 == pKF->N ? keyframe_handles().get(pKF, mDevice) : nullptr;
        if (H) { // the keyframe side is resident: only this call's flags and the frame side travel
            orbfe_keyframe* k1[1] = {H->h};
            int32_t* mp[1] = {match.data()};
            int nm = 0;
            const int r = orbfe_search_bow_keyframes(mDevice, 1, k1, nullptr, &a, mp, &nm);
            nmatches = r < 0 ? r : nm;
        } else {
            nmatches = orbfe_search_bow(mDevice, &a, match.data());
        }
        if (nmatches < 0) throw std::runtime_error("orbfe_search_bow failed");
        for (int i = 0; i < F.N; i++)
            if (match[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[match[i]]; // :376, :406
        return nmatches;
    }

    // ---- src/ORBmatcher.cc:823-963
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12)
    {
        using namespace orbfe_adapter;
        const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
        const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
        const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
        vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));
        std::vector<uint8_t> good1(n1), good2(n2), tmp1, tmp2;
        std::vector<float> ang1(n1, 0.f), ang2(n2, 0.f);
        for (int i = 0; i < n1; i++) {
            good1[i] = (vpMapPoints1[i] && !vpMapPoints1[i]->isBad()) ? 1 : 0; // :862-866
            if (i < (int)pKF1->mvKeysUn.size()) ang1[i] = pKF1->mvKeysUn[i].angle;
        }
        for (int i = 0; i < n2; i++) {
            good2[i] = (vpMapPoints2[i] && !vpMapPoints2[i]->isBad()) ? 1 : 0; // :884-888
            if (i < (int)pKF2->mvKeysUn.size()) ang2[i] = pKF2->mvKeysUn[i].angle;
        }
        const CSR c1 = toCSR(pKF1->mFeatVec), c2 = toCSR(pKF2->mFeatVec);
        orbfe_bow_args a;
        std::memset(&a, 0, sizeof(a));
        a.desc1 = dense_descriptors(pKF1->mDescriptors, n1, tmp1);
        a.n1 = n1;
        a.mask1 = good1.data();
        a.angle1 = ang1.data();
        a.fv1 = c1.view();
        a.limit1 = pKF1->NLeft != -1 ? (int)pKF1->mvKeysUn.size() : -1; // :855-857: right-camera features are skipped
        a.desc2 = dense_descriptors(pKF2->mDescriptors, n2, tmp2);
        a.n2 = n2;
        a.mask2 = good2.data();
        a.angle2 = ang2.data();
        a.fv2 = c2.view();
        a.limit2 = pKF2->NLeft != -1 ? (int)pKF2->mvKeysUn.size() : -1; // :874-876
        a.Nleft = -1;
        a.nnratio = mfNNratio;
        a.check_orientation = mbCheckOrientation ? 1 : 0;
        a.variant = 1;
        std::vector<int32_t> match((size_t)std::max(n1, 1), -1);
        int nmatches;
        // (handles for keyframes without a second camera: the angle this overload reads, mvKeysUn[i].angle, is then what a
        // handle holds for every feature)
        const bool plain = pKF1->NLeft == -1 && pKF2->NLeft == -1 && n1 == pKF1->N && n2 == pKF2->N &&
                           (int)pKF1->mvKeysUn.size() >= n1 && (int)pKF2->mvKeysUn.size() >= n2;
        std::shared_ptr<KeyFrameHandles::Handle> H1 = use_keyframe_handles() && plain ? keyframe_handles().get(pKF1, mDevice) : nullptr;
        std::shared_ptr<KeyFrameHandles::Handle> H2 = H1 ? keyframe_handles().get(pKF2, mDevice) : nullptr;
        if (H1 && H2) {
            orbfe_keyframe* k1[1] = {H1->h};
            orbfe_keyframe* k2[1] = {H2->h};
            int32_t* mp[1] = {match.data()};
            int nm = 0;
            const int r = orbfe_search_bow_keyframes(mDevice, 1, k1, k2, &a, mp, &nm);
            nmatches = r < 0 ? r : nm;
        } else {
            nmatches = orbfe_search_bow(mDevice, &a, match.data());
        }
        if (nmatches < 0) throw std::runtime_error("orbfe_search_bow failed");
        for (int i = 0; i < n1; i++)
            if (match[i] >= 0) vpMatches12[i] = vpMapPoints2[match[i]]; // :910
        return nmatches;
    }

    // ---- src/ORBmatcher.cc:1208-1449 (pinhole cameras without a second camera; the F12 argument is not read by
    // the reference either: its gate, Pinhole::epipolarConstrain_, rebuilds the matrix from R12, t12)
    int SearchForTriangulation_(KeyFrame* pKF1, KeyFrame* pKF2, cv::Matx33f /*F12*/,
                                std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo,
                                const bool bCoarse = false)
    {
        using namespace orbfe_adapter;
        if ((pKF1->mpCamera2 != nullptr) != (pKF2->mpCamera2 != nullptr))
            throw std::runtime_error("SearchForTriangulation_: one keyframe of a rig, one without (the reference reads both second cameras)");
        if (pKF1->mpCamera2 || pKF1->mpCamera->GetType() == 1u /* GeometricCamera::CAM_FISHEYE */)
            return SearchForTriangulationKB8(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse);
        // epipole in the second image (:1215-1220)
        const cv::Matx31f Cw = pKF1->GetCameraCenter_();
        const cv::Matx33f R2w = pKF2->GetRotation_();
        const cv::Matx31f t2w = pKF2->GetTranslation_();
        const cv::Matx31f C2 = R2w * Cw + t2w;
        const cv::Point2f ep = pKF2->mpCamera->project(C2);
        const cv::Matx33f R1w = pKF1->GetRotation_();
        const cv::Matx31f t1w = pKF1->GetTranslation_();
        const cv::Matx33f R12 = R1w * R2w.t();                  // :1234
        const cv::Matx31f t12 = -R1w * R2w.t() * t2w + t1w;     // :1235
        // what Pinhole::epipolarConstrain_ evaluates per candidate (Pinhole.cpp:161-164), once per keyframe pair
        cv::Matx33f t12x;
        t12x(0, 1) = -t12(2); t12x(0, 2) = t12(1);
        t12x(1, 0) = t12(2);  t12x(1, 2) = -t12(0);
        t12x(2, 0) = -t12(1); t12x(2, 1) = t12(0);
        const cv::Matx33f K1 = pKF1->mpCamera->toK_(), K2 = pKF2->mpCamera->toK_();
        const cv::Matx33f F12 = K1.t().inv() * t12x * R12 * K2.inv();
        lastF12 = F12;
        lastEp = ep;

        const int n1 = pKF1->N, n2 = pKF2->N;
        std::vector<uint8_t> has1(n1), has2(n2), tmp1, tmp2;
        std::vector<float> xy1(2 * (size_t)n1), xy2(2 * (size_t)n2), ang1(n1), ang2(n2);
        std::vector<int32_t> oct1(n1), oct2(n2);
        for (int i = 0; i < n1; i++) {
            has1[i] = pKF1->GetMapPoint(i) ? 1 : 0; // :1268-1274
            const cv::KeyPoint& kp = rig_keypoint(*pKF1, pKF1->NLeft, (size_t)i);
            xy1[2 * i] = kp.pt.x; xy1[2 * i + 1] = kp.pt.y; ang1[i] = kp.angle; oct1[i] = kp.octave;
        }
        for (int i = 0; i < n2; i++) {
            has2[i] = pKF2->GetMapPoint(i) ? 1 : 0; // :1307-1311
            const cv::KeyPoint& kp = rig_keypoint(*pKF2, pKF2->NLeft, (size_t)i);
            xy2[2 * i] = kp.pt.x; xy2[2 * i + 1] = kp.pt.y; ang2[i] = kp.angle; oct2[i] = kp.octave;
        }
        const CSR c1 = toCSR(pKF1->mFeatVec), c2 = toCSR(pKF2->mFeatVec);
        orbfe_tri_args a;
        std::memset(&a, 0, sizeof(a));
        a.desc1 = dense_descriptors(pKF1->mDescriptors, n1, tmp1); a.n1 = n1; a.hasMP1 = has1.data();
        a.kp1_xy = xy1.data(); a.angle1 = ang1.data(); a.octave1 = oct1.data(); a.uRight1 = pKF1->mvuRight.data();
        a.fv1 = c1.view();
        a.desc2 = dense_descriptors(pKF2->mDescriptors, n2, tmp2); a.n2 = n2; a.hasMP2 = has2.data();
        a.kp2_xy = xy2.data(); a.angle2 = ang2.data(); a.octave2 = oct2.data(); a.uRight2 = pKF2->mvuRight.data();
        a.fv2 = c2.view();
        for (int i = 0; i < 9; i++) a.F12[i] = F12.val[i];
        a.ep[0] = ep.x; a.ep[1] = ep.y;
        a.scaleFactors2 = pKF2->mvScaleFactors.data();
        a.levelSigma2_2 = pKF2->mvLevelSigma2.data();
        a.nlevels2 = (int)pKF2->mvScaleFactors.size();
        a.only_stereo = bOnlyStereo ? 1 : 0;
        a.coarse = bCoarse ? 1 : 0;
        a.check_orientation = mbCheckOrientation ? 1 : 0;
        std::vector<int32_t> pairs(2 * (size_t)std::max(n1, 1));
        int np;
        std::shared_ptr<KeyFrameHandles::Handle> H1 = use_keyframe_handles() ? keyframe_handles().get(pKF1, mDevice) : nullptr;
        std::shared_ptr<KeyFrameHandles::Handle> H2 = H1 && H1->tri ? keyframe_handles().get(pKF2, mDevice) : nullptr;
        if (H1 && H2 && H1->tri && H2->tri) { // both keyframes resident: the pair geometry and the has-MapPoint flags travel
            orbfe_tri_pair q;
            std::memset(&q, 0, sizeof q);
            for (int i = 0; i < 9; i++) q.F12[i] = F12.val[i];
            q.ep[0] = ep.x; q.ep[1] = ep.y;
            q.scaleFactors2 = a.scaleFactors2; q.levelSigma2_2 = a.levelSigma2_2; q.nlevels2 = a.nlevels2;
            q.only_stereo = a.only_stereo; q.coarse = a.coarse; q.check_orientation = a.check_orientation;
            q.hasMP2 = has2.data();
            orbfe_keyframe* k2[1] = {H2->h};
            int32_t* pp[1] = {pairs.data()};
            int cnt = 0;
            const int r = orbfe_search_tri_batch(H1->h, has1.data(), 1, k2, &q, pp, &cnt);
            np = r < 0 ? r : cnt;
        } else {
            np = orbfe_search_tri(mDevice, &a, pairs.data());
        }
        if (np < 0) throw std::runtime_error("orbfe_search_tri failed");
        vMatchedPairs.clear(); // :1435-1446
        vMatchedPairs.reserve(np);
        for (int k = 0; k < np; k++) vMatchedPairs.push_back(std::make_pair((size_t)pairs[2 * k], (size_t)pairs[2 * k + 1]));
        return np;
    }

    // ---- src/ORBmatcher.cc:1208-1449 for KannalaBrandt8 cameras: a monocular fisheye pair, or a two-camera rig with its
    // four relative poses (:1238-1248) and per-candidate camera pair (:1342-1370); the gate is
    // KannalaBrandt8::epipolarConstrain_ (a triangulation per candidate, on the device: orbfe_search_tri_kb8)
    int SearchForTriangulationKB8(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                                  const bool bOnlyStereo, const bool bCoarse)
    {
        using namespace orbfe_adapter;
        const bool rig = pKF1->mpCamera2 != nullptr;
        const cv::Matx31f Cw = pKF1->GetCameraCenter_();
        const cv::Matx33f R2w = pKF2->GetRotation_();
        const cv::Matx31f t2w = pKF2->GetTranslation_();
        const cv::Point2f ep = pKF2->mpCamera->project(R2w * Cw + t2w); // :1215-1220 (read without a rig only)
        const cv::Matx33f R1w = pKF1->GetRotation_();
        const cv::Matx31f t1w = pKF1->GetTranslation_();
        cv::Matx33f R[4];
        cv::Matx31f t[4];
        if (!rig) {
            R[0] = R1w * R2w.t();                  // :1234-1235
            t[0] = -R1w * R2w.t() * t2w + t1w;
        } else {                                    // :1238-1248: ll, lr, rl, rr
            const cv::Matx33f R1r = pKF1->GetRightRotation_(), R2r = pKF2->GetRightRotation_();
            const cv::Matx31f t1r = pKF1->GetRightTranslation_(), t2r = pKF2->GetRightTranslation_();
            R[0] = R1w * R2w.t();
            R[1] = R1w * R2r.t();
            R[2] = R1r * R2w.t();
            R[3] = R1r * R2r.t();
            t[0] = R1w * (-R2w.t() * t2w) + t1w;
            t[1] = R1w * (-R2r.t() * t2r) + t1w;
            t[2] = R1r * (-R2w.t() * t2w) + t1r;
            t[3] = R1r * (-R2r.t() * t2r) + t1r;
        }
        const int nposes = rig ? 4 : 1;
        lastR12.assign(9 * (size_t)nposes, 0.f);
        lastT12.assign(3 * (size_t)nposes, 0.f);
        for (int c = 0; c < nposes; c++) {
            for (int i = 0; i < 9; i++) lastR12[9 * c + i] = R[c].val[i];
            for (int i = 0; i < 3; i++) lastT12[3 * c + i] = t[c].val[i];
        }
        lastEp = ep;
        float P[4][8];
        GeometricCamera* cams[4] = {pKF1->mpCamera, rig ? pKF1->mpCamera2 : pKF1->mpCamera, pKF2->mpCamera,
                                    rig ? pKF2->mpCamera2 : pKF2->mpCamera};
        for (int c = 0; c < 4; c++)
            for (int i = 0; i < 8; i++) P[c][i] = cams[c]->getParameter(i);

        const int n1 = pKF1->N, n2 = pKF2->N;
        std::vector<uint8_t> has1(n1), has2(n2), tmp1, tmp2;
        std::vector<float> xy1(2 * (size_t)n1), xy2(2 * (size_t)n2), ang1(n1), ang2(n2);
        std::vector<int32_t> oct1(n1), oct2(n2);
        for (int i = 0; i < n1; i++) {
            has1[i] = pKF1->GetMapPoint(i) ? 1 : 0;
            const cv::KeyPoint& kp = rig_keypoint(*pKF1, pKF1->NLeft, (size_t)i);
            xy1[2 * i] = kp.pt.x; xy1[2 * i + 1] = kp.pt.y; ang1[i] = kp.angle; oct1[i] = kp.octave;
        }
        for (int i = 0; i < n2; i++) {
            has2[i] = pKF2->GetMapPoint(i) ? 1 : 0;
            const cv::KeyPoint& kp = rig_keypoint(*pKF2, pKF2->NLeft, (size_t)i);
            xy2[2 * i] = kp.pt.x; xy2[2 * i + 1] = kp.pt.y; ang2[i] = kp.angle; oct2[i] = kp.octave;
        }
        const CSR c1 = toCSR(pKF1->mFeatVec), c2 = toCSR(pKF2->mFeatVec);
        orbfe_tri_kb8_args a;
        std::memset(&a, 0, sizeof(a));
        a.desc1 = dense_descriptors(pKF1->mDescriptors, n1, tmp1); a.n1 = n1; a.hasMP1 = has1.data();
        a.kp1_xy = xy1.data(); a.angle1 = ang1.data(); a.octave1 = oct1.data();
        a.uRight1 = pKF1->mvuRight.empty() ? nullptr : pKF1->mvuRight.data(); a.fv1 = c1.view(); a.Nleft1 = pKF1->NLeft;
        a.desc2 = dense_descriptors(pKF2->mDescriptors, n2, tmp2); a.n2 = n2; a.hasMP2 = has2.data();
        a.kp2_xy = xy2.data(); a.angle2 = ang2.data(); a.octave2 = oct2.data();
        a.uRight2 = pKF2->mvuRight.empty() ? nullptr : pKF2->mvuRight.data(); a.fv2 = c2.view(); a.Nleft2 = pKF2->NLeft;
        a.kb8_1L = P[0]; a.kb8_1R = rig ? P[1] : nullptr; a.kb8_2L = P[2]; a.kb8_2R = rig ? P[3] : nullptr;
        a.R12 = lastR12.data(); a.t12 = lastT12.data();
        a.ep[0] = ep.x; a.ep[1] = ep.y;
        a.scaleFactors2 = pKF2->mvScaleFactors.data();
        a.levelSigma2_1 = pKF1->mvLevelSigma2.data(); a.levelSigma2_2 = pKF2->mvLevelSigma2.data();
        a.nlevels1 = (int)pKF1->mvLevelSigma2.size(); a.nlevels2 = (int)pKF2->mvLevelSigma2.size();
        a.only_stereo = bOnlyStereo ? 1 : 0;
        a.coarse = bCoarse ? 1 : 0;
        a.check_orientation = mbCheckOrientation ? 1 : 0;
        std::vector<int32_t> pairs(2 * (size_t)std::max(n1, 1));
        const int np = orbfe_search_tri_kb8(mDevice, &a, pairs.data());
        if (np < 0) throw std::runtime_error("orbfe_search_tri_kb8 failed");
        vMatchedPairs.clear();
        vMatchedPairs.reserve(np);
        for (int k = 0; k < np; k++) vMatchedPairs.push_back(std::make_pair((size_t)pairs[2 * k], (size_t)pairs[2 * k + 1]));
        return np;
    }

    // ---- src/ORBmatcher.cc:44-197 (Tracking::SearchLocalPoints)
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3,
                           const bool bFarPoints = false, const float thFarPoints = 50.0f)
    {
        using namespace orbfe_adapter;
        const bool bFactor = th != 1.0;
        Queries q;
        for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) { // the object walk of :50-63, :137-143
            MapPoint* pMP = vpMapPoints[iMP];
            if (!pMP->mbTrackInView && !pMP->mbTrackInViewR) continue;
            if (bFarPoints && pMP->mTrackDepth > thFarPoints) continue;
            if (pMP->isBad()) continue;
            bool left = false;
            const cv::Mat d = pMP->GetDescriptor();
            if (pMP->mbTrackInView) {
                const int nPredictedLevel = pMP->mnTrackScaleLevel;
                float r = RadiusByViewingCos(pMP->mTrackViewCos);
                if (bFactor) r *= th;
                q.push(pMP, d.data, pMP->mTrackProjX, pMP->mTrackProjY, r * F.mvScaleFactors[nPredictedLevel],
                       nPredictedLevel - 1, nPredictedLevel, pMP->mTrackProjXR, 0, 0.f, pMP->Observations() > 0);
                left = true;
            }
            if (F.Nleft != -1 && pMP->mbTrackInViewR) {
                const int nPredictedLevel = pMP->mnTrackScaleLevelR;
                if (nPredictedLevel != -1) {
                    const float r = RadiusByViewingCos(pMP->mTrackViewCosR);
                    q.push(pMP, d.data, pMP->mTrackProjXR, pMP->mTrackProjYR, r * F.mvScaleFactors[nPredictedLevel],
                           nPredictedLevel - 1, nPredictedLevel, 0.f, 1 | (left ? 2 : 0), 0.f, pMP->Observations() > 0);
                }
            }
        }
        std::vector<int32_t> qMatch, featMatch;
        const int nmatches = run_projection(F, F.N, F.Nleft, F.mDescriptors, F.mvuRight.empty() ? nullptr : F.mvuRight.data(),
                                            taken_of(F.mvpMapPoints, F.N), F.mnMinX, F.mnMinY, F.mfGridElementWidthInv,
                                            F.mfGridElementHeightInv, q, 0, TH_HIGH, false, nullptr, 0, false,
                                            F.Nleft != -1 && !F.mvLeftToRightMatch.empty() ? F.mvLeftToRightMatch.data() : nullptr,
                                            F.Nleft != -1 && !F.mvRightToLeftMatch.empty() ? F.mvRightToLeftMatch.data() : nullptr,
                                            qMatch, featMatch);
        for (int i = 0; i < F.N; i++)
            if (featMatch[i] >= 0) F.mvpMapPoints[i] = q.mp[featMatch[i]]; // :112, :117-121, :171
        return nmatches;
    }

    // ---- src/ORBmatcher.cc:2193-2419 (Tracking::TrackWithMotionModel), with the right-camera searches of a
    // two-camera rig (:2326-2395): one more query per point, into the right grid
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono)
    {
        using namespace orbfe_adapter;
        const bool rig = CurrentFrame.Nleft != -1;
        float Rcw[9], tcw[3], Rlw[9], tlw[3], Rrl[9], trl[3];
        frame_pose(CurrentFrame, Rcw, tcw);
        frame_pose(LastFrame, Rlw, tlw);
        if (rig) frame_trl(CurrentFrame, Rrl, trl);
        // twc = -Rcw^T tcw; tlc = Rlw twc + tlw (:2207-2212)
        float twc[3], tlc[3];
        for (int i = 0; i < 3; i++)
            twc[i] = (float)(-((double)Rcw[i] * tcw[0] + (double)Rcw[3 + i] * tcw[1] + (double)Rcw[6 + i] * tcw[2]));
        transform(Rlw, tlw, twc, tlc);
        const bool bForward = tlc[2] > CurrentFrame.mb && !bMono;
        const bool bBackward = -tlc[2] > CurrentFrame.mb && !bMono;
        Queries q;
        for (int i = 0; i < LastFrame.N; i++) {
            MapPoint* pMP = LastFrame.mvpMapPoints[i];
            if (!pMP || LastFrame.mvbOutlier[i]) continue;
            float x3Dw[3], x3Dc[3];
            world_pos(pMP, x3Dw);
            transform(Rcw, tcw, x3Dw, x3Dc);
            const float invzc = 1.0 / x3Dc[2];
            if (invzc < 0) continue;
            const cv::Point2f uv = CurrentFrame.mpCamera->project(cv::Point3f(x3Dc[0], x3Dc[1], x3Dc[2]));
            if (uv.x < CurrentFrame.mnMinX || uv.x > CurrentFrame.mnMaxX) continue;
            if (uv.y < CurrentFrame.mnMinY || uv.y > CurrentFrame.mnMaxY) continue;
            const int nLastOctave = (LastFrame.Nleft == -1 || i < LastFrame.Nleft) ? LastFrame.mvKeys[i].octave
                                                                                   : LastFrame.mvKeysRight[i - LastFrame.Nleft].octave;
            const float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
            int lo, hi; // the arguments of GetFeaturesInArea (:2248-2253)
            if (bForward) { lo = nLastOctave; hi = -1; }
            else if (bBackward) { lo = 0; hi = nLastOctave; }
            else { lo = nLastOctave - 1; hi = nLastOctave + 1; }
            const cv::KeyPoint& kpLF = rig_keypoint(LastFrame, LastFrame.Nleft, (size_t)i);
            const float ur = uv.x - CurrentFrame.mbf * invzc; // :2272
            // (a temporal point of UpdateLastFrame has no observations: it does not hide its feature from later points)
            q.push(pMP, pMP->GetDescriptor().data, uv.x, uv.y, radius, lo, hi, ur, 0, kpLF.angle, pMP->Observations() > 0);
            if (rig) {
                // :2326-2341: the point in the right camera, projected with mpCamera (sic); same radius and levels, no
                // image-bounds test; the block stands behind `if(vIndices2.empty()) continue;` of the left search (bit 2)
                float x3Dr[3];
                transform(Rrl, trl, x3Dc, x3Dr);
                const cv::Point2f uvR = CurrentFrame.mpCamera->project(cv::Point3f(x3Dr[0], x3Dr[1], x3Dr[2]));
                q.push(pMP, pMP->GetDescriptor().data, uvR.x, uvR.y, radius, lo, hi, 0.f, 1 | 4, kpLF.angle,
                       pMP->Observations() > 0);
            }
        }
        std::vector<int32_t> qMatch, featMatch;
        const int nmatches = run_projection(CurrentFrame, CurrentFrame.N, CurrentFrame.Nleft, CurrentFrame.mDescriptors,
                                            CurrentFrame.mvuRight.empty() ? nullptr : CurrentFrame.mvuRight.data(),
                                            taken_of(CurrentFrame.mvpMapPoints, CurrentFrame.N), CurrentFrame.mnMinX,
                                            CurrentFrame.mnMinY, CurrentFrame.mfGridElementWidthInv,
                                            CurrentFrame.mfGridElementHeightInv, q, 1, TH_HIGH, mbCheckOrientation, nullptr, 0,
                                            false, nullptr, nullptr, qMatch, featMatch);
        // :2301 and the cull of :2399-2415: the features the cull cleared are reset to NULL, the survivors hold pMP
        for (int i = 0; i < CurrentFrame.N; i++)
            if (featMatch[i] >= 0) CurrentFrame.mvpMapPoints[i] = q.mp[featMatch[i]];
        for (size_t k = 0; k < q.mp.size(); k++) // a feature some point wrote that holds none now: cleared by the cull
            if (qMatch[k] >= 0 && featMatch[qMatch[k]] < 0) CurrentFrame.mvpMapPoints[qMatch[k]] = static_cast<MapPoint*>(NULL);
        return nmatches;
    }

    // ---- src/ORBmatcher.cc:1643-1841 (LocalMapping::SearchInNeighbors); bRight = the right camera of a rig (:1647-1658)
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0, const bool bRight = false)
    {
        using namespace orbfe_adapter;
        if (bRight && pKF->NLeft == -1) throw std::runtime_error("Fuse(bRight): the keyframe has no second camera");
        float Rcw[9], tcw[3], Ow[3];
        kf_pose(pKF, bRight, Rcw, tcw, Ow);
        GeometricCamera* pCamera = bRight ? pKF->mpCamera2 : pKF->mpCamera;
        const float bf = pKF->mbf;
        Queries q;
        for (size_t i = 0; i < vpMapPoints.size(); i++) { // :1690-1742, unchanged conditions
            MapPoint* pMP = vpMapPoints[i];
            if (!pMP) continue;
            if (pMP->isBad()) continue;
            if (pMP->IsInKeyFrame(pKF)) continue;
            float p3Dw[3], p3Dc[3], Pn[3];
            world_pos(pMP, p3Dw);
            transform(Rcw, tcw, p3Dw, p3Dc);
            if (p3Dc[2] < 0.0f) continue;
            const float invz = 1 / p3Dc[2];
            const cv::Point2f uv = pCamera->project(cv::Point3f(p3Dc[0], p3Dc[1], p3Dc[2]));
            if (!pKF->IsInImage(uv.x, uv.y)) continue;
            const float ur = uv.x - bf * invz;
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
            const float dist3D = (float)std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]); // cv::norm
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            normal_of(pMP, Pn);
            if ((double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2] < 0.5 * dist3D) continue;
            const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
            const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
            // the level test of :1771-1772 is the window's level range; the chi2 test of :1774-1799 runs on the device
            // (a right-camera query reads its candidates from the right grid and mvuRight with the camera-local index,
            // :1762-1775, :1801 -- both on the device)
            q.push(pMP, pMP->GetDescriptor().data, uv.x, uv.y, radius, nPredictedLevel - 1, nPredictedLevel, ur, bRight ? 1 : 0,
                   0.f, 0);
        }
        std::vector<int32_t> qMatch, featMatch;
        std::vector<uint8_t> none((size_t)std::max(pKF->N, 1), 0);
        run_projection(*pKF, pKF->N, pKF->NLeft, pKF->mDescriptors, pKF->mvuRight.data(), none, pKF->mnMinX, pKF->mnMinY,
                       pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv, q, 1, TH_LOW, false,
                       pKF->mvInvLevelSigma2.data(), (int)pKF->mvInvLevelSigma2.size(), true, nullptr, nullptr, qMatch,
                       featMatch);
        int nFused = 0;
        for (size_t k = 0; k < q.mp.size(); k++) { // :1813-1838: the sequential object logic, unchanged
            const int bestIdx = qMatch[k];
            if (bestIdx < 0) continue;
            MapPoint* pMP = q.mp[k];
            // :1690-1692 tests isBad() / IsInKeyFrame() per ITERATION, i.e. after the earlier iterations' AddObservation /
            // Replace: a point that occurs twice in vpMapPoints (a rig's keyframe holds the same point at its left and at
            // its right index, LocalMapping.cc:829-870) is fused once.  The search itself does not depend on object
            // state, so re-testing here is equivalent.
            if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
            MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
        return nFused;
    }

    // ---- src/ORBmatcher.cc:2421-2541 (Tracking::Relocalization): project the keyframe's points into the frame
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                           const int ORBdist)
    {
        using namespace orbfe_adapter;
        float Rcw[9], tcw[3], Ow[3];
        frame_pose(CurrentFrame, Rcw, tcw);
        for (int i = 0; i < 3; i++)
            Ow[i] = (float)(-((double)Rcw[i] * tcw[0] + (double)Rcw[3 + i] * tcw[1] + (double)Rcw[6 + i] * tcw[2]));
        const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
        Queries q;
        for (size_t i = 0; i < vpMPs.size(); i++) {
            MapPoint* pMP = vpMPs[i];
            if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
            float x3Dw[3], x3Dc[3];
            world_pos(pMP, x3Dw);
            transform(Rcw, tcw, x3Dw, x3Dc);
            const cv::Point2f uv = CurrentFrame.mpCamera->project(cv::Point3f(x3Dc[0], x3Dc[1], x3Dc[2]));
            if (uv.x < CurrentFrame.mnMinX || uv.x > CurrentFrame.mnMaxX) continue;
            if (uv.y < CurrentFrame.mnMinY || uv.y > CurrentFrame.mnMaxY) continue;
            const float PO[3] = {x3Dw[0] - Ow[0], x3Dw[1] - Ow[1], x3Dw[2] - Ow[2]};
            const float dist3D = norm3(PO);
            if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
            const int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
            const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
            q.push(pMP, pMP->GetDescriptor().data, uv.x, uv.y, radius, nPredictedLevel - 1, nPredictedLevel + 1, 0.f, 0,
                   pKF->mvKeysUn[i].angle, 1);
        }
        std::vector<uint8_t> taken((size_t)std::max(CurrentFrame.N, 1), 0); // :2484: any point at the feature hides it
        for (int i = 0; i < CurrentFrame.N; i++) taken[i] = CurrentFrame.mvpMapPoints[i] ? 1 : 0;
        std::vector<int32_t> qMatch, featMatch;
        const int nmatches = run_projection(CurrentFrame, CurrentFrame.N, -1, CurrentFrame.mDescriptors, nullptr, taken,
                                            CurrentFrame.mnMinX, CurrentFrame.mnMinY, CurrentFrame.mfGridElementWidthInv,
                                            CurrentFrame.mfGridElementHeightInv, q, 1, ORBdist, mbCheckOrientation, nullptr, 0,
                                            false, nullptr, nullptr, qMatch, featMatch);
        for (int i = 0; i < CurrentFrame.N; i++)
            if (featMatch[i] >= 0) CurrentFrame.mvpMapPoints[i] = q.mp[featMatch[i]];
        for (size_t k = 0; k < q.mp.size(); k++)
            if (qMatch[k] >= 0 && featMatch[qMatch[k]] < 0) CurrentFrame.mvpMapPoints[qMatch[k]] = static_cast<MapPoint*>(NULL);
        return nmatches;
    }

    // ---- src/ORBmatcher.cc:473-586 (LoopClosing: points seen from a Sim3-corrected pose) and its twin :588-704
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints,
                           std::vector<MapPoint*>& vpMatched, int th, float ratioHamming = 1.0)
    {
        std::vector<KeyFrame*> none;
        return sim3_projection(pKF, Scw, vpPoints, nullptr, vpMatched, none, th, ratioHamming);
    }
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints,
                           const std::vector<KeyFrame*>& vpPointsKFs, std::vector<MapPoint*>& vpMatched,
                           std::vector<KeyFrame*>& vpMatchedKF, int th, float ratioHamming = 1.0)
    {
        return sim3_projection(pKF, Scw, vpPoints, &vpPointsKFs, vpMatched, vpMatchedKF, th, ratioHamming);
    }

    // ---- src/ORBmatcher.cc:706-821 (monocular initialisation)
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                                int windowSize = 10)
    {
        using namespace orbfe_adapter;
        const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
        vnMatches12 = std::vector<int>(n1, -1);
        std::vector<int32_t> oct1(n1), oct2(n2);
        std::vector<float> ang1(n1), ang2(n2), prev(2 * (size_t)std::max(n1, 1)), kx2(n2), ky2(n2);
        std::vector<uint8_t> tmp1, tmp2;
        for (int i = 0; i < n1; i++) {
            oct1[i] = F1.mvKeysUn[i].octave;
            ang1[i] = F1.mvKeysUn[i].angle;
            prev[2 * i] = vbPrevMatched[i].x;
            prev[2 * i + 1] = vbPrevMatched[i].y;
        }
        for (int i = 0; i < n2; i++) {
            oct2[i] = F2.mvKeysUn[i].octave;
            ang2[i] = F2.mvKeysUn[i].angle;
            kx2[i] = F2.mvKeysUn[i].pt.x;
            ky2[i] = F2.mvKeysUn[i].pt.y;
        }
        orbfe_init_args a;
        std::memset(&a, 0, sizeof(a));
        a.desc1 = dense_descriptors(F1.mDescriptors, n1, tmp1); a.n1 = n1; a.octave1 = oct1.data(); a.angle1 = ang1.data();
        a.prev_xy = prev.data();
        a.desc2 = dense_descriptors(F2.mDescriptors, n2, tmp2); a.n2 = n2; a.kx2 = kx2.data(); a.ky2 = ky2.data();
        a.octave2 = oct2.data(); a.angle2 = ang2.data();
        a.minX = F2.mnMinX; a.minY = F2.mnMinY; a.gridWInv = F2.mfGridElementWidthInv; a.gridHInv = F2.mfGridElementHeightInv;
        a.window_size = windowSize; a.nnratio = mfNNratio; a.check_orientation = mbCheckOrientation ? 1 : 0;
        std::vector<int32_t> m((size_t)std::max(n1, 1), -1);
        const int nmatches = orbfe_search_initialization(mDevice, &a, m.data());
        if (nmatches < 0) throw std::runtime_error("orbfe_search_initialization failed");
        for (int i = 0; i < n1; i++) {
            vnMatches12[i] = m[i];
            if (m[i] >= 0) vbPrevMatched[i] = F2.mvKeysUn[m[i]].pt; // :813-816
        }
        return nmatches;
    }

    // ---- src/ORBmatcher.cc:965-1206: the cv::Mat twin of SearchForTriangulation_ (Tracking.cc:4313); F12 is not read
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat /*F12*/,
                               std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo,
                               const bool bCoarse = false)
    {
        return SearchForTriangulation_(pKF1, pKF2, cv::Matx33f(), vMatchedPairs, bOnlyStereo, bCoarse);
    }
    // ---- src/ORBmatcher.cc:1452-1641: the overload that also returns the triangulated points (declared :73-75 of the
    // header, no caller in the reference).  Neither F12 nor bOnlyStereo is read by it; its gate is
    // pCamera1->matchAndtriangulate, which only KannalaBrandt8 implements (Pinhole's returns false: no pairs).
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat /*F12*/,
                               std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool /*bOnlyStereo*/,
                               std::vector<cv::Mat>& vMatchedPoints)
    {
        using namespace orbfe_adapter;
        const bool rig1 = pKF1->NLeft != -1, rig2 = pKF2->NLeft != -1;
        GeometricCamera* cams[4] = {pKF1->mpCamera, rig1 ? pKF1->mpCamera2 : pKF1->mpCamera, pKF2->mpCamera,
                                    rig2 ? pKF2->mpCamera2 : pKF2->mpCamera};
        const bool fisheye = pKF1->mpCamera->GetType() == 1u /* GeometricCamera::CAM_FISHEYE */;
        float P[4][8], T[4][12];
        KeyFrame* kfs[2] = {pKF1, pKF2};
        for (int c = 0; c < 4; c++) {
            float R[9], t[3], O[3];
            kf_pose(kfs[c / 2], (c & 1) && (c / 2 ? rig2 : rig1), R, t, O); // GetPose / GetRightPose (:1519-1533)
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) T[c][4 * i + j] = R[3 * i + j];
                T[c][4 * i + 3] = t[i];
            }
            for (int i = 0; i < 8; i++) P[c][i] = (fisheye && cams[c]->GetType() == 1u) ? cams[c]->getParameter(i) : 0.f;
        }
        const int n1 = pKF1->N, n2 = pKF2->N;
        std::vector<uint8_t> has1(n1), has2(n2), tmp1, tmp2;
        std::vector<float> xy1(2 * (size_t)n1), xy2(2 * (size_t)n2), ang1(n1), ang2(n2);
        std::vector<int32_t> oct1(n1), oct2(n2);
        for (int i = 0; i < n1; i++) {
            has1[i] = pKF1->GetMapPoint(i) ? 1 : 0;
            const cv::KeyPoint& kp = rig_keypoint(*pKF1, pKF1->NLeft, (size_t)i);
            xy1[2 * i] = kp.pt.x; xy1[2 * i + 1] = kp.pt.y; ang1[i] = kp.angle; oct1[i] = kp.octave;
        }
        for (int i = 0; i < n2; i++) {
            has2[i] = pKF2->GetMapPoint(i) ? 1 : 0;
            const cv::KeyPoint& kp = rig_keypoint(*pKF2, pKF2->NLeft, (size_t)i);
            xy2[2 * i] = kp.pt.x; xy2[2 * i + 1] = kp.pt.y; ang2[i] = kp.angle; oct2[i] = kp.octave;
        }
        const CSR c1 = toCSR(pKF1->mFeatVec), c2 = toCSR(pKF2->mFeatVec);
        orbfe_tri3d_args a;
        std::memset(&a, 0, sizeof(a));
        a.desc1 = dense_descriptors(pKF1->mDescriptors, n1, tmp1); a.n1 = n1; a.hasMP1 = has1.data();
        a.kp1_xy = xy1.data(); a.angle1 = ang1.data(); a.octave1 = oct1.data(); a.fv1 = c1.view(); a.Nleft1 = pKF1->NLeft;
        a.desc2 = dense_descriptors(pKF2->mDescriptors, n2, tmp2); a.n2 = n2; a.hasMP2 = has2.data();
        a.kp2_xy = xy2.data(); a.angle2 = ang2.data(); a.octave2 = oct2.data(); a.fv2 = c2.view(); a.Nleft2 = pKF2->NLeft;
        a.kb8_1L = fisheye ? P[0] : nullptr; a.kb8_1R = P[1]; a.kb8_2L = P[2]; a.kb8_2R = P[3];
        a.Tcw1L = T[0]; a.Tcw1R = T[1]; a.Tcw2L = T[2]; a.Tcw2R = T[3];
        a.levelSigma2_1 = pKF1->mvLevelSigma2.data(); a.levelSigma2_2 = pKF2->mvLevelSigma2.data();
        a.nlevels1 = (int)pKF1->mvLevelSigma2.size(); a.nlevels2 = (int)pKF2->mvLevelSigma2.size();
        a.check_orientation = mbCheckOrientation ? 1 : 0;
        std::vector<int32_t> pairs(2 * (size_t)std::max(n1, 1));
        std::vector<float> points(3 * (size_t)std::max(n1, 1));
        const int np = orbfe_search_tri_3d(mDevice, &a, pairs.data(), points.data());
        if (np < 0) throw std::runtime_error("orbfe_search_tri_3d failed");
        vMatchedPairs.clear(); // :1625-1637
        vMatchedPairs.reserve(np);
        for (int k = 0; k < np; k++) {
            vMatchedPairs.push_back(std::make_pair((size_t)pairs[2 * k], (size_t)pairs[2 * k + 1]));
            cv::Mat x3D(3, 1, CV_32F);
            for (int i = 0; i < 3; i++) x3D.at<float>(i) = points[3 * (size_t)k + i];
            vMatchedPoints.push_back(x3D);
        }
        return np;
    }

    // ---- src/ORBmatcher.cc:1967-2191 (LoopClosing / merging): mutual projection search under a Sim3
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12,
                     const cv::Mat& R12, const cv::Mat& t12, const float th)
    {
        using namespace orbfe_adapter;
        float R1w[9], t1w[3], O1[3], R2w[9], t2w[3], O2[3];
        kf_pose(pKF1, false, R1w, t1w, O1);
        kf_pose(pKF2, false, R2w, t2w, O2);
        // sR12 = s12*R12; sR21 = (1/s12)*R12^T; t21 = -sR21*t12 (:1983-1985; scalings by a double, one gemm)
        float sR12[9], sR21[9], t12f[3], t21[3];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
                sR12[3 * i + j] = (float)((double)s12 * (double)R12.at<float>(i, j));
                sR21[3 * i + j] = (float)((1.0 / s12) * (double)R12.at<float>(j, i));
            }
            t12f[i] = t12.at<float>(i);
        }
        for (int i = 0; i < 3; i++)
            t21[i] = (float)(-((double)sR21[3 * i] * t12f[0] + (double)sR21[3 * i + 1] * t12f[1] + (double)sR21[3 * i + 2] * t12f[2]));
        const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
        std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
        for (int i = 0; i < N1; i++) {
            MapPoint* pMP = vpMatches12[i];
            if (pMP) {
                vbAlreadyMatched1[i] = true;
                const int idx2 = std::get<0>(pMP->GetIndexInKeyFrame(pKF2));
                if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
            }
        }
        // one direction: the points of `from` (camera pose Rw, tw) seen in `into` through (sR, t)
        auto direction = [&](KeyFrame* into, const std::vector<MapPoint*>& pts, const std::vector<bool>& done,
                             const float Rw[9], const float tw[3], const float sR[9], const float t[3],
                             std::vector<int>& vnMatch) {
            Queries q;
            for (int i = 0; i < (int)pts.size(); i++) {
                MapPoint* pMP = pts[i];
                if (!pMP || done[i] || pMP->isBad()) continue;
                float p3Dw[3], pA[3], pB[3];
                world_pos(pMP, p3Dw);
                transform(Rw, tw, p3Dw, pA);
                transform(sR, t, pA, pB);
                if (pB[2] < 0.0) continue;
                const float invz = 1.0 / pB[2];
                const float x = pB[0] * invz, y = pB[1] * invz;
                const float u = into->fx * x + into->cx, v = into->fy * y + into->cy;
                if (!into->IsInImage(u, v)) continue;
                const float dist3D = norm3(pB);
                if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
                const int nPredictedLevel = pMP->PredictScale(dist3D, into);
                q.push(pMP, pMP->GetDescriptor().data, u, v, th * into->mvScaleFactors[nPredictedLevel], nPredictedLevel - 1,
                       nPredictedLevel, 0.f, 0, 0.f, 0);
                q.src.push_back(i);
            }
            std::vector<uint8_t> none((size_t)std::max(into->N, 1), 0);
            std::vector<int32_t> qMatch, featMatch;
            run_projection(*into, into->N, -1, into->mDescriptors, nullptr, none, into->mnMinX, into->mnMinY,
                           into->mfGridElementWidthInv, into->mfGridElementHeightInv, q, 1, TH_HIGH, false, nullptr, 0, false,
                           nullptr, nullptr, qMatch, featMatch);
            for (size_t k = 0; k < q.mp.size(); k++) vnMatch[q.src[k]] = qMatch[k];
        };
        std::vector<int> vnMatch1(N1, -1), vnMatch2(N2, -1);
        direction(pKF2, vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, vnMatch1);
        direction(pKF1, vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12f, vnMatch2);
        int nFound = 0;
        for (int i1 = 0; i1 < N1; i1++) { // :2167-2188: keep what both directions agree on
            const int idx2 = vnMatch1[i1];
            if (idx2 >= 0 && vnMatch2[idx2] == i1) {
                vpMatches12[i1] = vpMapPoints2[idx2];
                nFound++;
            }
        }
        return nFound;
    }

    // ---- src/ORBmatcher.cc:1843-1965 (LoopClosing::SearchAndFuse)
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint)
    {
        using namespace orbfe_adapter;
        float Rcw[9], tcw[3], Ow[3];
        decompose_sim3(Scw, Rcw, tcw, Ow);
        const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
        Queries q;
        project_points(pKF, Rcw, tcw, Ow, vpPoints, spAlreadyFound, (float)th, q);
        std::vector<uint8_t> none((size_t)std::max(pKF->N, 1), 0);
        for (auto& b : q.blocks) b = 0; // candidates do not hide features from each other
        std::vector<int32_t> qMatch, featMatch;
        run_projection(*pKF, pKF->N, -1, pKF->mDescriptors, nullptr, none, pKF->mnMinX, pKF->mnMinY, pKF->mfGridElementWidthInv,
                       pKF->mfGridElementHeightInv, q, 1, TH_LOW, false, nullptr, 0, false, nullptr, nullptr, qMatch, featMatch);
        int nFused = 0;
        for (size_t k = 0; k < q.mp.size(); k++) { // :1941-1958
            const int bestIdx = qMatch[k];
            if (bestIdx < 0) continue;
            MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[q.src[k]] = pMPinKF;
            } else {
                q.mp[k]->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(q.mp[k], bestIdx);
            }
            nFused++;
        }
        return nFused;
    }

public:
    static const int TH_LOW = 50;
    static const int TH_HIGH = 100;
    static const int HISTO_LENGTH = 30;
    cv::Matx33f lastF12; // what SearchForTriangulation_ formed from the poses (read by the adapter's test)
    cv::Point2f lastEp;
    std::vector<float> lastR12, lastT12; // the relative poses SearchForTriangulationKB8 formed (1 or 4: ll, lr, rl, rr)

protected:
    float RadiusByViewingCos(const float& viewCos) // :199-205
    {
        if (viewCos > 0.998) return 2.5;
        else return 4.0;
    }

    // the object walk shared by the Sim3 projection overloads and Fuse(Scw) (:491-531, :1862-1905): project every
    // eligible point into the keyframe and queue one window search per point
    void project_points(KeyFrame* pKF, const float Rcw[9], const float tcw[3], const float Ow[3],
                        const std::vector<MapPoint*>& vpPoints, const std::set<MapPoint*>& spAlreadyFound, float th,
                        orbfe_adapter::Queries& q)
    {
        using namespace orbfe_adapter;
        for (int iMP = 0; iMP < (int)vpPoints.size(); iMP++) {
            MapPoint* pMP = vpPoints[iMP];
            if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
            float p3Dw[3], p3Dc[3], Pn[3];
            world_pos(pMP, p3Dw);
            transform(Rcw, tcw, p3Dw, p3Dc);
            if (p3Dc[2] < 0.0) continue;
            const cv::Point2f uv = pKF->mpCamera->project(cv::Point3f(p3Dc[0], p3Dc[1], p3Dc[2]));
            if (!pKF->IsInImage(uv.x, uv.y)) continue;
            const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
            const float dist = norm3(PO);
            if (dist < pMP->GetMinDistanceInvariance() || dist > pMP->GetMaxDistanceInvariance()) continue;
            normal_of(pMP, Pn);
            if ((double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2] < 0.5 * dist) continue;
            const int nPredictedLevel = pMP->PredictScale(dist, pKF);
            q.push(pMP, pMP->GetDescriptor().data, uv.x, uv.y, th * pKF->mvScaleFactors[nPredictedLevel], nPredictedLevel - 1,
                   nPredictedLevel, 0.f, 0, 0.f, 1);
            q.src.push_back(iMP);
        }
    }

    int sim3_projection(KeyFrame* pKF, const cv::Mat& Scw, const std::vector<MapPoint*>& vpPoints,
                        const std::vector<KeyFrame*>* vpPointsKFs, std::vector<MapPoint*>& vpMatched,
                        std::vector<KeyFrame*>& vpMatchedKF, int th, float ratioHamming)
    {
        using namespace orbfe_adapter;
        float Rcw[9], tcw[3], Ow[3];
        decompose_sim3(Scw, Rcw, tcw, Ow);
        std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
        spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
        Queries q;
        project_points(pKF, Rcw, tcw, Ow, vpPoints, spAlreadyFound, (float)th, q);
        std::vector<uint8_t> taken((size_t)std::max(pKF->N, 1), 0); // :547-548: a feature that already has a match
        for (int i = 0; i < pKF->N && i < (int)vpMatched.size(); i++) taken[i] = vpMatched[i] ? 1 : 0;
        std::vector<int32_t> qMatch, featMatch;
        // bestDist <= TH_LOW * ratioHamming on integers: th_high = floor of the product
        const int nmatches = run_projection(*pKF, pKF->N, -1, pKF->mDescriptors, nullptr, taken, pKF->mnMinX, pKF->mnMinY,
                                            pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv, q, 1,
                                            (int)std::floor(TH_LOW * ratioHamming), false, nullptr, 0, false, nullptr, nullptr,
                                            qMatch, featMatch);
        for (int i = 0; i < pKF->N; i++)
            if (featMatch[i] >= 0) {
                vpMatched[i] = q.mp[featMatch[i]];
                if (vpPointsKFs) vpMatchedKF[i] = (*vpPointsKFs)[q.src[featMatch[i]]];
            }
        return nmatches;
    }

    static std::vector<uint8_t> taken_of(const std::vector<MapPoint*>& v, int n)
    { // F.mvpMapPoints[idx] && Observations() > 0 (:83-85, :2262-2264)
        std::vector<uint8_t> t((size_t)std::max(n, 1), 0);
        for (int i = 0; i < n && i < (int)v.size(); i++) t[i] = (v[i] && v[i]->Observations() > 0) ? 1 : 0;
        return t;
    }

    template <class T>
    int run_projection(const T& f, int n, int nLeft, const cv::Mat& D, const float* uright, const std::vector<uint8_t>& taken,
                       float minX, float minY, float gwInv, float ghInv, const orbfe_adapter::Queries& q, int mode,
                       int thHigh, bool checkOri, const float* invSigma2, int nLevels, bool chi2, const int* l2r,
                       const int* r2l, std::vector<int32_t>& qMatch, std::vector<int32_t>& featMatch)
    {
        using namespace orbfe_adapter;
        std::vector<uint8_t> tmp;
        orbfe_proj_args a;
        std::memset(&a, 0, sizeof(a));
        a.n = n;
        a.uright = (nLeft == -1 || chi2) ? uright : nullptr; // (Fuse reads mvuRight of a rig's keyframe too, :1775)
        // a Frame: its side of the search lives in a handle on the device, made on first sight (see FrameHandles above)
        orbfe_frame* handle = nullptr;
        FeatureArrays fa;
        if constexpr (std::is_same<T, Frame>::value) {
            if (use_frame_handles() && n > 0) {
                FrameHandles& H = frame_handles();
                FrameHandles::Entry* e = H.find(&f, f.mnId, n, D.data, mDevice);
                if (!e) {
                    fa.fill(f, n, nLeft);
                    orbfe_proj_args fs = a;
                    fs.desc = dense_descriptors(D, n, tmp);
                    fs.kx = fa.kx.data(); fs.ky = fa.ky.data(); fs.octave = fa.octave.data(); fs.angle = fa.angle.data();
                    fs.Nleft = nLeft;
                    fs.minX = minX; fs.minY = minY; fs.gridWInv = gwInv; fs.gridHInv = ghInv;
                    fs.inv_level_sigma2 = invSigma2; fs.n_levels = nLevels;
                    e = H.slot();
                    if (orbfe_frame_create(&e->h, mDevice, &fs) < 0) {
                        e->h = nullptr;
                        throw std::runtime_error("orbfe_frame_create failed");
                    }
                    e->obj = &f; e->id = f.mnId; e->n = n; e->desc = D.data; e->device = mDevice;
                    e->used = ++H.clock;
                    H.creates++;
                }
                handle = e->h;
            }
        }
        if (!handle) {
            fa.fill(f, n, nLeft);
            a.desc = dense_descriptors(D, n, tmp);
            a.kx = fa.kx.data(); a.ky = fa.ky.data(); a.octave = fa.octave.data(); a.angle = fa.angle.data();
        }
        a.taken = taken.data();
        a.Nleft = nLeft;
        a.left_to_right = l2r; a.right_to_left = r2l;
        a.minX = minX; a.minY = minY; a.gridWInv = gwInv; a.gridHInv = ghInv;
        a.nq = q.size();
        a.qdesc = q.desc.data(); a.qx = q.x.data(); a.qy = q.y.data(); a.qr = q.r.data();
        a.qmin_level = q.minLevel.data(); a.qmax_level = q.maxLevel.data();
        a.qxr = a.uright ? q.xr.data() : nullptr;
        a.qflags = nLeft != -1 ? q.flags.data() : nullptr;
        a.qangle = q.angle.data();
        a.qblocks = q.blocks.data();
        a.mode = mode; a.nnratio = mfNNratio; a.th_high = thHigh; a.check_orientation = checkOri ? 1 : 0;
        a.inv_level_sigma2 = invSigma2; a.n_levels = nLevels; a.chi2_gate = chi2 ? 1 : 0;
        qMatch.assign((size_t)std::max(q.size(), 1), -1);
        featMatch.assign((size_t)std::max(n, 1), -1);
        const int r = handle ? orbfe_search_projection_frame(handle, &a, qMatch.data(), featMatch.data())
                             : orbfe_search_projection(mDevice, &a, qMatch.data(), featMatch.data());
        if (r < 0) throw std::runtime_error("orbfe_search_projection failed");
        return r;
    }

    float mfNNratio;
    bool mbCheckOrientation;
    int mDevice;
};

} // namespace ORB_SLAM3

#endif
