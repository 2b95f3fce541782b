/*
 * adapters/ORBextractor.h -- ORB_SLAM3::ORBextractor with its reference signatures, backed by the
 * C ABI of liborbfe.so (include/orbfe.h).
 *
 * Drop this header in place of the reference's include/ORBextractor.h (and remove
 * src/ORBextractor.cc from the build): Frame::ExtractORB (src/Frame.cc:413-420), the three
 * `new ORBextractor(...)` in Tracking (src/Tracking.cc:1151-1157) and every scale getter call
 * (src/Frame.cc:107-113) compile unchanged.  Public surface mirrored from
 * include/ORBextractor.h:43-107: ctor, operator(), GetLevels/GetScaleFactor/GetScaleFactors/
 * GetInverseScaleFactors/GetScaleSigmaSquares/GetInverseScaleSigmaSquares, mvImagePyramid.
 *
 * With OpenCV present the cv:: types are used; without it (this repo's own test build) a minimal
 * stand-in for cv::KeyPoint / cv::Mat is provided so the adapter itself can be compiled and
 * exercised.  mvImagePyramid needs no flag (round 4): the pyramid stays on the device and a level is
 * downloaded the first time it is INDEXED after an extraction, which is how its only reader -- the SAD
 * refinement of Frame::ComputeStereoMatches, src/Frame.cc:804, :894-909 -- uses it; monocular and
 * fisheye callers, which never look, pay nothing.  ORBextractor::fetchPyramid = true downloads all
 * levels right after every operator() instead (a caller that hands the extractor to another thread
 * while the next frame is being extracted).
 */
#ifndef ORBFE_ADAPTER_ORBEXTRACTOR_H
#define ORBFE_ADAPTER_ORBEXTRACTOR_H

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/orbfe.h"

#include "cv_standins.h"

namespace ORB_SLAM3 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
        : ctx(nullptr), nlevels(nlevels), scaleFactor(scaleFactor)
    {
        const int r = orbfe_create(&ctx, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device);
        if (r < 0) throw std::runtime_error(std::string("ORBextractor: ") + orbfe_error_string(r));
        mvScaleFactor.resize(nlevels);
        mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels);
        mvInvLevelSigma2.resize(nlevels);
        orbfe_get_scale_tables(ctx, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                               mvInvLevelSigma2.data());
        mvImagePyramid.bind(this, nlevels);
        pyramidStore.resize(nlevels);
        levelValid.assign((size_t)nlevels, 0);
    }
    ~ORBextractor() { orbfe_destroy(ctx); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // Compute the ORB features and descriptors on an image (mask is ignored, as in the reference :56).
    int operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                   cv::OutputArray _descriptors, std::vector<int>& vLappingArea)
    {
#ifdef ORBFE_HAVE_OPENCV
        cv::Mat image = _image.getMat();
        if (image.empty()) return -1;
        CV_Assert(image.type() == CV_8UC1);
        const uint8_t* data = image.data;
        const size_t step = image.step;
#else
        const cv::Mat& image = _image;
        if (image.empty()) return -1;
        const uint8_t* data = image.data;
        const size_t step = image.step;
#endif
        const int cap = orbfe_max_keypoints(ctx, image.rows, image.cols);
        if (cap < 0) throw std::runtime_error(std::string("ORBextractor: ") + orbfe_error_string(cap));
        static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_kp), "cv::KeyPoint layout");
        // (the cap-sized result buffers are members: no allocation and no zero-fill per frame)
        if (scratchKps.size() < (size_t)cap) scratchKps.resize((size_t)cap);
        if (scratchDesc.size() < (size_t)cap * 32) scratchDesc.resize((size_t)cap * 32);
        int n = 0;
        std::fill(levelValid.begin(), levelValid.end(), 0); // whatever mvImagePyramid held belongs to the previous frame
        haveFrame = false;
        const int mono = orbfe_extract(ctx, data, image.rows, image.cols, step, vLappingArea[0], vLappingArea[1],
                                       reinterpret_cast<orbfe_kp*>(scratchKps.data()), scratchDesc.data(), cap, &n);
        if (mono < -1) throw std::runtime_error(std::string("ORBextractor: ") + orbfe_error_string(mono));
        haveFrame = mono >= 0;
        _keypoints.assign(scratchKps.begin(), scratchKps.begin() + n);
        fill_descriptors(_descriptors, scratchDesc.data(), n);
        if (fetchPyramid)
            for (int l = 0; l < nlevels; l++) ensure_level(l);
        return mono;
    }

    // A rectified stereo frame in one go: BOTH images through one batched call on this extractor's context -- in place
    // of the two threads Frame::Frame starts for mpORBextractorLeft / mpORBextractorRight (src/Frame.cc:119-122; both
    // are built with the same parameters, src/Tracking.cc:1151-1155) -- then Frame::ComputeStereoMatches (src/Frame.cc
    // :797-967) between the two results while keypoints, descriptors and pyramids are still on the device:
    // mvuRight / mvDepth come back with the keypoints, no pyramid is downloaded.  Returns the number of stereo
    // matches, or a negative code; monoLeft / monoRight as the two operator() calls would return them.
    int ExtractStereoPair(cv::InputArray imLeft, cv::InputArray imRight, std::vector<cv::KeyPoint>& keysLeft,
                          cv::OutputArray descLeft, std::vector<cv::KeyPoint>& keysRight, cv::OutputArray descRight,
                          const std::vector<int>& lapLeft, const std::vector<int>& lapRight, float mb, float mbf,
                          std::vector<float>& mvuRight, std::vector<float>& mvDepth, int* monoLeft = nullptr,
                          int* monoRight = nullptr)
    {
#ifdef ORBFE_HAVE_OPENCV
        cv::Mat L = imLeft.getMat(), R = imRight.getMat();
#else
        const cv::Mat &L = imLeft, &R = imRight;
#endif
        if (L.empty() || R.empty()) return -1;
        if (L.rows != R.rows || L.cols != R.cols || L.step != R.step) throw std::runtime_error("ExtractStereoPair: unequal images");
        const int cap = orbfe_max_keypoints(ctx, L.rows, L.cols);
        if (cap < 0) throw std::runtime_error(std::string("ORBextractor: ") + orbfe_error_string(cap));
        if (scratchKps.size() < (size_t)2 * cap) scratchKps.resize((size_t)2 * cap);
        if (scratchDesc.size() < (size_t)2 * cap * 32) scratchDesc.resize((size_t)2 * cap * 32);
        std::vector<cv::KeyPoint>& kps = scratchKps;
        std::vector<uint8_t>& desc = scratchDesc;
        std::fill(levelValid.begin(), levelValid.end(), 0);
        haveFrame = false;
        const uint8_t* two[2] = {L.data, R.data};
        const int lap[4] = {lapLeft[0], lapLeft[1], lapRight[0], lapRight[1]};
        int n[2] = {0, 0}, mono[2] = {0, 0};
        // one call, one host wait: both images as a batch of two, ComputeStereoMatches queued behind them on the same stream
        scratchU.assign((size_t)cap, -1.0f);
        scratchD.assign((size_t)cap, -1.0f);
        std::vector<float>&uR = scratchU, &dep = scratchD;
        const int matches = orbfe_extract_stereo_pair(ctx, two[0], two[1], L.rows, L.cols, L.step, lap,
                                                      reinterpret_cast<orbfe_kp*>(kps.data()), desc.data(), cap, n, mono, mb, mbf,
                                                      uR.data(), dep.data());
        if (matches < 0) throw std::runtime_error(std::string("orbfe_extract_stereo_pair: ") + orbfe_error_string(matches));
        haveFrame = true; // (mvImagePyramid then shows the LEFT image's levels, image 0 of the pair)
        mvuRight.assign(uR.begin(), uR.begin() + n[0]);
        mvDepth.assign(dep.begin(), dep.begin() + n[0]);
        keysLeft.assign(kps.begin(), kps.begin() + n[0]);
        keysRight.assign(kps.begin() + cap, kps.begin() + cap + n[1]);
        fill_descriptors(descLeft, desc.data(), n[0]);
        fill_descriptors(descRight, desc.data() + (size_t)cap * 32, n[1]);
        if (monoLeft) *monoLeft = mono[0];
        if (monoRight) *monoRight = mono[1];
        return matches;
    }

    // Beyond the reference's interface: where the last operator() left the frame's descriptors in HBM (rows of 32 bytes, image 0
    // of the last call; null before the first frame).  Matcher calls of liborbfe.so that are handed this pointer order themselves
    // after the extraction; ORBVocabulary::computeBoW(DeviceDescriptors(), N, 4) runs Frame::ComputeBoW without the descriptors
    // crossing PCIe again.  Valid until the next call on this extractor.
    const uint8_t* DeviceDescriptors(int* cap = nullptr)
    {
        const uint8_t* d = nullptr;
        int c = 0;
        if (!haveFrame || orbfe_get_device_outputs(ctx, nullptr, &d, nullptr, &c, nullptr) != 0) return nullptr;
        if (cap) *cap = c;
        return d;
    }

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Stands where `std::vector<cv::Mat> mvImagePyramid` stood (reference include/ORBextractor.h:83).  Indexing a level
    // downloads it on first use after an extraction (the reference's readers only index: src/Frame.cc:804, :894-909);
    // size() / iteration behave like the vector's.  The Mats are views of padded buffers, like the reference's
    // (:1165-1172: mvImagePyramid[level] = temp(Rect(EDGE_THRESHOLD, EDGE_THRESHOLD, sz.width, sz.height))), so the
    // 19-px reflected frame around a level is addressable from its Mat.
    class PyramidView {
    public:
        size_t size() const { return mats.size(); }
        bool empty() const { return mats.empty(); }
        cv::Mat& operator[](size_t l)
        {
            owner->ensure_level((int)l);
            return mats[l];
        }
        const cv::Mat& operator[](size_t l) const
        {
            owner->ensure_level((int)l);
            return mats[l];
        }
        cv::Mat& at(size_t l) { return (*this)[l]; }
        std::vector<cv::Mat>::iterator begin() { return all().begin(); }
        std::vector<cv::Mat>::iterator end() { return all().end(); }
        operator const std::vector<cv::Mat>&() { return all(); }

    private:
        friend class ORBextractor;
        void bind(ORBextractor* o, int n)
        {
            owner = o;
            mats.resize((size_t)n);
        }
        std::vector<cv::Mat>& all()
        {
            for (size_t l = 0; l < mats.size(); l++) owner->ensure_level((int)l);
            return mats;
        }
        ORBextractor* owner = nullptr;
        mutable std::vector<cv::Mat> mats;
    };
    PyramidView mvImagePyramid;
    bool fetchPyramid = false; // true: every operator() downloads all levels at once (see the header comment)
    orbfe_ctx* handle() { return ctx; }

protected:
    // one level of the last extracted image, device -> the level's padded host buffer (once per frame and level)
    void ensure_level(int l)
    {
        if (l < 0 || l >= nlevels || levelValid[(size_t)l] || !haveFrame) return;
        int r = 0, c = 0;
        if (orbfe_get_level(ctx, 0, l, nullptr, 0, &r, &c) < 0) return;
#ifdef ORBFE_HAVE_OPENCV
        pyramidStore[l].create(r, c, CV_8U);
        if (orbfe_get_level(ctx, 0, l, pyramidStore[l].data, pyramidStore[l].step, &r, &c) < 0) return;
        mvImagePyramid.mats[l] = pyramidStore[l](cv::Rect(19, 19, c - 38, r - 38)); // ROI inside the padded buffer
#else
        if (pyramidStore[l].rows != r || pyramidStore[l].cols != c) pyramidStore[l].create(r, c);
        if (orbfe_get_level(ctx, 0, l, pyramidStore[l].data, pyramidStore[l].step, &r, &c) < 0) return;
        mvImagePyramid.mats[l] = cv::Mat(r - 38, c - 38, pyramidStore[l].ptr(19) + 19, pyramidStore[l].step);
#endif
        levelValid[(size_t)l] = 1;
    }
    static void fill_descriptors(cv::OutputArray out, const uint8_t* rows, int n)
    {
#ifdef ORBFE_HAVE_OPENCV
        if (n == 0) {
            out.release();
            return;
        }
        out.create(n, 32, CV_8U);
        cv::Mat d = out.getMat();
        if (d.isContinuous()) std::memcpy(d.data, rows, (size_t)n * 32);
        else
            for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), rows + (size_t)i * 32, 32);
#else
        if (n == 0) {
            out.release();
            return;
        }
        out.create(n, 32);
        std::memcpy(out.data, rows, (size_t)n * 32);
#endif
    }
    orbfe_ctx* ctx;
    int nlevels;
    double scaleFactor;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<cv::Mat> pyramidStore; // one padded buffer per level (sized by the constructor)
    std::vector<char> levelValid;      // mvImagePyramid[l] shows the last extracted image
    bool haveFrame = false;
    std::vector<cv::KeyPoint> scratchKps; // cap-sized result buffers, reused from call to call
    std::vector<uint8_t> scratchDesc;
    std::vector<float> scratchU, scratchD;
};

} // namespace ORB_SLAM3

#endif
