// declaration-only mock (see core/core.hpp)
#ifndef ORBFE_OPENCV_MOCK_FEATURES2D_HPP
#define ORBFE_OPENCV_MOCK_FEATURES2D_HPP
#include "../core/core.hpp"
namespace cv {
void FAST(InputArray image, std::vector<KeyPoint>& keypoints, int threshold, bool nonmaxSuppression = true);
class BFMatcher {
public:
    BFMatcher(int normType = NORM_L2, bool crossCheck = false);
    void knnMatch(InputArray queryDescriptors, InputArray trainDescriptors, std::vector<std::vector<DMatch>>& matches, int k,
                  InputArray mask = noArray(), bool compactResult = false) const;
};
}  // namespace cv
#endif
