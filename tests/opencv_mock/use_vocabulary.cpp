// Translation unit of tests/test_adapter_syntax.py: adapters/ORBVocabulary.h as it is compiled INSIDE ORB-SLAM3 -- OpenCV on the
// include path (the declaration-only mock), the real DBoW2 containers (declared below as DBoW2 declares them:
// Thirdparty/DBoW2/DBoW2/BowVector.h:59-60, FeatureVector.h:23-24) instead of the stand-ins -- and used like Frame::ComputeBoW
// (src/Frame.cc:724-731).
#include <map>
#include <vector>

namespace DBoW2 {
typedef unsigned int NodeId;
typedef unsigned int WordId;
typedef double WordValue;
class BowVector : public std::map<WordId, WordValue> {};
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
} // namespace DBoW2
#define ORBFE_HAVE_ORBSLAM 1
#include "../../adapters/ORBVocabulary.h"

#ifndef ORBFE_HAVE_OPENCV
#error "the OpenCV branch was not selected"
#endif

double compute_bow_like_frame(ORB_SLAM3::ORBVocabulary* mpORBvocabulary, const cv::Mat& mDescriptors, DBoW2::BowVector& mBowVec,
                              DBoW2::FeatureVector& mFeatVec)
{
    if (!mpORBvocabulary->loadFromTextFile("ORBvoc.txt")) return -1;
    std::vector<cv::Mat> vCurrentDesc;
    for (int j = 0; j < mDescriptors.rows; j++) vCurrentDesc.push_back(mDescriptors.row(j));
    mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4);
    return mpORBvocabulary->score(mBowVec, mBowVec) + mpORBvocabulary->size();
}
