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
    // (no flag set: mvImagePyramid is filled when it is indexed)
    int monoLeft = (*mpORBextractorLeft)(im, mask, mvKeys, mDescriptors, vLapping);
    unsigned long long h = 1469598103934665603ull;
    for (int i = 0; i < mDescriptors.rows * 32; i++) h = (h ^ mDescriptors.data[i]) * 1099511628211ull;
    for (size_t i = 0; i < mvKeys.size(); i++) {
        const unsigned char* p = reinterpret_cast<const unsigned char*>(&mvKeys[i]);
        for (int k = 0; k < 28; k++) h = (h ^ p[k]) * 1099511628211ull;
    }
    printf("%d %zu %llu %d %d %d\n", monoLeft, mvKeys.size(), h, mpORBextractorLeft->GetLevels(),
           mpORBextractorLeft->mvImagePyramid[3].rows, mpORBextractorLeft->mvImagePyramid[3].cols);
    {
        // the access pattern of Frame::ComputeStereoMatches (src/Frame.cc:804, :894-909) on mvImagePyramid, with no flag set:
        // the 11x11 window around every keypoint at its own octave, summed and hashed
        const int w = 5;
        const std::vector<float> inv = mpORBextractorLeft->GetInverseScaleFactors();
        unsigned long long hp = 1469598103934665603ull;
        const int nRows = mpORBextractorLeft->mvImagePyramid[0].rows;
        for (size_t i = 0; i < mvKeys.size(); i++) {
            const cv::KeyPoint& kpL = mvKeys[i];
            const float scaleFactor = inv[kpL.octave];
            const int scaleduL = (int)(kpL.pt.x * scaleFactor + 0.5f), scaledvL = (int)(kpL.pt.y * scaleFactor + 0.5f);
            cv::Mat IL = mpORBextractorLeft->mvImagePyramid[kpL.octave].rowRange(scaledvL - w, scaledvL + w + 1).colRange(scaleduL - w, scaleduL + w + 1);
            unsigned sum = 0;
            for (int r = 0; r < IL.rows; r++)
                for (int c = 0; c < IL.cols; c++) sum += IL.ptr(r)[c];
            hp = (hp ^ sum) * 1099511628211ull;
        }
        printf("%d %llu\n", nRows, hp);
    }
    cv::Mat empty;
    printf("%d\n", (*mpORBextractorLeft)(empty, mask, mvKeys, mDescriptors, vLapping));
    if (argc > 5) { // a right image: the stereo frame in one call (ExtractStereoPair), results as checksums
        FILE* g = fopen(argv[5], "rb");
        if (!g) return 3;
        cv::Mat right(rows, cols);
        if (fread(right.data, 1, (size_t)rows * cols, g) != (size_t)rows * cols) return 4;
        fclose(g);
        std::vector<cv::KeyPoint> kl, kr;
        cv::Mat dl, dr;
        std::vector<int> lap = {0, 0};
        std::vector<float> uR, depth;
        int ml = 0, mr = 0;
        const float mbf = 47.90639384423901f, mb = mbf / 435.2046959714599f;
        const int m = mpORBextractorLeft->ExtractStereoPair(im, right, kl, dl, kr, dr, lap, lap, mb, mbf, uR, depth, &ml, &mr);
        unsigned long long hl = 1469598103934665603ull, hr = hl, hu = hl;
        for (int i = 0; i < dl.rows * 32; i++) hl = (hl ^ dl.data[i]) * 1099511628211ull;
        for (size_t i = 0; i < kl.size() * 28; i++) hl = (hl ^ reinterpret_cast<const unsigned char*>(kl.data())[i]) * 1099511628211ull;
        for (int i = 0; i < dr.rows * 32; i++) hr = (hr ^ dr.data[i]) * 1099511628211ull;
        for (size_t i = 0; i < kr.size() * 28; i++) hr = (hr ^ reinterpret_cast<const unsigned char*>(kr.data())[i]) * 1099511628211ull;
        for (size_t i = 0; i < uR.size() * 4; i++) hu = (hu ^ reinterpret_cast<const unsigned char*>(uR.data())[i]) * 1099511628211ull;
        for (size_t i = 0; i < depth.size() * 4; i++) hu = (hu ^ reinterpret_cast<const unsigned char*>(depth.data())[i]) * 1099511628211ull;
        printf("%d %zu %zu %llu %llu %llu %d %d\n", m, kl.size(), kr.size(), hl, hr, hu, ml, mr);
    }
    delete mpORBextractorLeft;
    return 0;
}
