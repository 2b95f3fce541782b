// Exercises adapters/ORBextractor.h the way Frame::ExtractORB does (reference src/Frame.cc:413-420).
// Built and run by tests/test_gpu_adapter.py on the GPU box: prints keypoints + a checksum that the
// test compares with the oracle.
#define ORBFE_NO_OPENCV 1
#include <cstdio>
#include <cstdlib>

#include "ORBextractor.h"

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    const int rows = atoi(argv[2]), cols = atoi(argv[3]), nf = atoi(argv[4]);
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    cv::Mat im(rows, cols);
    if (fread(im.data, 1, (size_t)rows * cols, f) != (size_t)rows * cols) return 4;
    fclose(f);
    ORB_SLAM3::ORBextractor* mpORBextractorLeft = new ORB_SLAM3::ORBextractor(nf, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> mvKeys;
    cv::Mat mDescriptors, mask;
    std::vector<int> vLapping = {0, 1000};
    mpORBextractorLeft->fetchPyramid = true;
    int monoLeft = (*mpORBextractorLeft)(im, mask, mvKeys, mDescriptors, vLapping);
    unsigned long long h = 1469598103934665603ull;
    for (int i = 0; i < mDescriptors.rows * 32; i++) h = (h ^ mDescriptors.data[i]) * 1099511628211ull;
    for (size_t i = 0; i < mvKeys.size(); i++) {
        const unsigned char* p = reinterpret_cast<const unsigned char*>(&mvKeys[i]);
        for (int k = 0; k < 28; k++) h = (h ^ p[k]) * 1099511628211ull;
    }
    printf("%d %zu %llu %d %d %d\n", monoLeft, mvKeys.size(), h, mpORBextractorLeft->GetLevels(),
           mpORBextractorLeft->mvImagePyramid[3].rows, mpORBextractorLeft->mvImagePyramid[3].cols);
    cv::Mat empty;
    printf("%d\n", (*mpORBextractorLeft)(empty, mask, mvKeys, mDescriptors, vLapping));
    delete mpORBextractorLeft;
    return 0;
}
