/*
 * adapters/orbslam_standins.h -- minimal stand-ins for the ORB-SLAM3 classes ORBmatcher walks (Frame, KeyFrame,
 * MapPoint, GeometricCamera / Pinhole, DBoW2::FeatureVector), with exactly the member names the reference uses
 * (include/Frame.h, include/KeyFrame.h, include/MapPoint.h, include/CameraModels/Pinhole.h,
 * Thirdparty/DBoW2/DBoW2/FeatureVector.h), so that adapters/ORBmatcher.h compiles against them here -- where OpenCV,
 * Eigen and DBoW2 are absent -- and against the real classes inside ORB-SLAM3 without a change.
 * Only what the adapted methods read or write exists; nothing here computes anything of the hot path.
 */
#ifndef ORBFE_ADAPTER_ORBSLAM_STANDINS_H
#define ORBFE_ADAPTER_ORBSLAM_STANDINS_H

#include <cmath>
#include <map>
#include <set>
#include <tuple>
#include <vector>

#include "cv_standins.h"

namespace DBoW2 {
typedef unsigned int NodeId;
typedef unsigned int WordId;
typedef double WordValue;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
class BowVector : public std::map<WordId, WordValue> {}; // Thirdparty/DBoW2/DBoW2/BowVector.h:59-60
} // namespace DBoW2

namespace ORB_SLAM3 {

class KeyFrame;

class GeometricCamera { // Pinhole (src/CameraModels/Pinhole.cpp) or, with mnType = 1, KannalaBrandt8
public:
    std::vector<float> mvParameters; // fx, fy, cx, cy (+ k0..k3 for the fisheye model)
    unsigned int mnType = 0;         // GeometricCamera::CAM_PINHOLE = 0, CAM_FISHEYE = 1 (GeometricCamera.h:84-85)
    unsigned int GetType() { return mnType; }
    float getParameter(const int i) { return mvParameters[i]; }
    cv::Point2f project(const cv::Matx31f& m) const { return project(cv::Point3f(m(0), m(1), m(2))); }
    cv::Point2f project(const cv::Point3f& p3) const
    {
        cv::Point2f p;
        if (mnType == 1) { // equidistant model: image radius = polynomial of the angle to the optical axis
            const float rho2 = p3.x * p3.x + p3.y * p3.y;
            const float th = atan2f(sqrtf(rho2), p3.z), az = atan2f(p3.y, p3.x);
            const float t2 = th * th, t3 = th * t2, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
            const float rd = th + mvParameters[4] * t3 + mvParameters[5] * t5 + mvParameters[6] * t7 + mvParameters[7] * t9;
            p.x = mvParameters[0] * rd * cosf(az) + mvParameters[2];
            p.y = mvParameters[1] * rd * sinf(az) + mvParameters[3];
            return p;
        }
        p.x = mvParameters[0] * p3.x / p3.z + mvParameters[2]; // Pinhole.cpp:45-51
        p.y = mvParameters[1] * p3.y / p3.z + mvParameters[3];
        return p;
    }
    cv::Matx33f toK_() const
    {
        cv::Matx33f K;
        K(0, 0) = mvParameters[0];
        K(1, 1) = mvParameters[1];
        K(0, 2) = mvParameters[2];
        K(1, 2) = mvParameters[3];
        K(2, 2) = 1.f;
        return K;
    }
};

class MapPoint {
public:
    long unsigned int mnId = 0;
    // Tracking::SearchLocalPoints / Frame::isInFrustum results (include/MapPoint.h:111-121)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackProjXR = 0, mTrackProjYR = 0, mTrackViewCos = 0,
          mTrackViewCosR = 0;
    bool mbTrackInView = false, mbTrackInViewR = false;
    int mnTrackScaleLevel = 0, mnTrackScaleLevelR = -1;

    bool isBad() { return mbBad; }
    int Observations() { return nObs; }
    cv::Mat GetDescriptor() { return cv::Mat(1, 32, mDescriptor, 32); }
    cv::Matx31f GetWorldPos_() { return mWorldPos; }
    cv::Matx31f GetNormal_() { return mNormalVector; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    bool IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }
    std::tuple<int, int> GetIndexInKeyFrame(KeyFrame* pKF)
    {
        auto it = mObservations.find(pKF);
        return it == mObservations.end() ? std::tuple<int, int>(-1, -1) : std::tuple<int, int>(it->second, -1);
    }
    int PredictScale(const float& currentDist, class Frame* pF); // MapPoint.cc:565-580, below
    int PredictScale(const float& currentDist, KeyFrame* pKF); // MapPoint.cc:548-563, below
    // what Fuse does to the map (MapPoint.cc Replace / AddObservation): recorded for the test
    void Replace(MapPoint* pMP)
    { // MapPoint.cc:213-270: the replaced point is dead from here on
        if (pMP == this) return;
        mbBad = true;
        mpReplaced = pMP;
    }
    void AddObservation(KeyFrame* pKF, int idx) { mObservations[pKF] = idx; }

    bool mbBad = false;
    int nObs = 1;
    uint8_t mDescriptor[32] = {0};
    cv::Matx31f mWorldPos, mNormalVector;
    float mfMinDistance = 0, mfMaxDistance = 0;
    std::map<KeyFrame*, int> mObservations;
    MapPoint* mpReplaced = nullptr;
};

class Frame {
public:
    // include/Frame.h: `static long unsigned int nNextId; long unsigned int mnId;`, Frame.cc: mnId = nNextId++
    long unsigned int mnId = next_id()++;
    static long unsigned int& next_id()
    {
        static long unsigned int n = 0;
        return n;
    }
    int N = 0, Nleft = -1;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn, mvKeysRight;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<float> mvScaleFactors;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    float mbf = 0, mb = 0;
    GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
    std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
    float mfLogScaleFactor = 0;
    int mnScaleLevels = 8;
    cv::Matx33f mRcw_; // rotation / translation of mTcw (the reference slices the 4x4 cv::Mat mTcw, ORBmatcher.cc:2204-2205)
    cv::Matx31f mtcw_;
    cv::Mat mTrl;      // 3x4 CV_32F, left camera -> right camera (two-camera rigs, ORBmatcher.cc:2327)
};

class KeyFrame {
public:
    // include/KeyFrame.h: `static long unsigned int nNextId; long unsigned int mnId;`, KeyFrame.cc: mnId = nNextId++
    static long unsigned int& next_id()
    {
        static long unsigned int n = 0;
        return n;
    }
    long unsigned int mnId = next_id()++;
    int N = 0, NLeft = -1;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn, mvKeysRight;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    float mfLogScaleFactor = 0;
    int mnScaleLevels = 8;
    GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;

    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
    std::set<MapPoint*> GetMapPoints()
    { // KeyFrame.cc:467-481: the non-null, non-bad points
        std::set<MapPoint*> s;
        for (MapPoint* p : mvpMapPoints)
            if (p && !p->isBad()) s.insert(p);
        return s;
    }
    cv::Matx33f GetRotation_() { return Rcw; }
    cv::Matx31f GetTranslation_() { return tcw; }
    cv::Matx31f GetCameraCenter_() { return Ow; }
    cv::Matx33f GetRightRotation_() { return RcwR; }
    cv::Matx31f GetRightTranslation_() { return tcwR; }
    cv::Matx31f GetRightCameraCenter_() { return OwR; }
    bool IsInImage(const float& x, const float& y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }

    std::vector<MapPoint*> mvpMapPoints;
    cv::Matx33f Rcw, RcwR; // ..R: the right camera of a rig (KeyFrame::GetRightPose, KeyFrame.cc)
    cv::Matx31f tcw, Ow, tcwR, OwR;
};

inline int MapPoint::PredictScale(const float& currentDist, KeyFrame* pKF)
{
    const float ratio = mfMaxDistance / currentDist;
    int nScale = (int)std::ceil(std::log(ratio) / pKF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pKF->mnScaleLevels) nScale = pKF->mnScaleLevels - 1;
    return nScale;
}

inline int MapPoint::PredictScale(const float& currentDist, Frame* pF)
{
    const float ratio = mfMaxDistance / currentDist;
    int nScale = (int)std::ceil(std::log(ratio) / pF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
    return nScale;
}

} // namespace ORB_SLAM3

#endif
