// Translation unit of tests/test_adapter_syntax.py: the adapters' ORBFE_HAVE_OPENCV branches seen by a compiler
// (g++ -fsyntax-only -I tests/opencv_mock).  Written like the reference's callers: Frame::ExtractORB (src/Frame.cc:413-420).
#include "../../adapters/ORBextractor.h"

#ifndef ORBFE_HAVE_OPENCV
#error "the OpenCV branches were not selected: opencv2/core/core.hpp must be found on the include path"
#endif

int extract_like_frame(ORB_SLAM3::ORBextractor* mpORBextractorLeft, const cv::Mat& im, std::vector<cv::KeyPoint>& mvKeys,
                       cv::Mat& mDescriptors)
{
    std::vector<int> vLapping = {0, 1000};
    mpORBextractorLeft->fetchPyramid = true;
    const int monoLeft = (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors, vLapping);
    const cv::Mat& level3 = mpORBextractorLeft->mvImagePyramid[3];
    return monoLeft + level3.rows + mpORBextractorLeft->GetLevels() + (int)mpORBextractorLeft->GetScaleFactors().size();
}

int stereo_like_frame(ORB_SLAM3::ORBextractor* ex, const cv::Mat& l, const cv::Mat& r)
{
    std::vector<cv::KeyPoint> kl, kr;
    cv::Mat dl, dr;
    std::vector<int> lap = {0, 0};
    std::vector<float> uR, depth;
    return ex->ExtractStereoPair(l, r, kl, dl, kr, dr, lap, lap, 0.11f, 47.9f, uR, depth);
}
