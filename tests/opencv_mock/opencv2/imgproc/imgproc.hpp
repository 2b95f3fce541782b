// declaration-only mock (see core/core.hpp)
#ifndef ORBFE_OPENCV_MOCK_IMGPROC_HPP
#define ORBFE_OPENCV_MOCK_IMGPROC_HPP
#include "../core/core.hpp"
namespace cv {
enum InterpolationFlags { INTER_NEAREST = 0, INTER_LINEAR = 1 };
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_DEFAULT);
}  // namespace cv
#endif
