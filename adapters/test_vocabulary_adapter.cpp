// Exercises adapters/ORBVocabulary.h the way ORB-SLAM3 does: System loads the text file (src/System.cc:82), Frame::ComputeBoW
// (src/Frame.cc:724-731) converts mDescriptors to a vector of rows and calls transform(..., 4), KeyFrameDatabase scores two
// vectors (src/KeyFrameDatabase.cc:162).  Built and run by tests/test_gpu_vocabulary_adapter.py, which compares the printed
// vectors with the oracle's.
// usage: test_vocabulary_adapter voc.txt desc.raw n levelsup [desc2.raw n2]
#define ORBFE_NO_OPENCV 1
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "ORBVocabulary.h"

static std::vector<cv::Mat> toDescriptorVector(const cv::Mat& Descriptors) // src/Converter.cc:24-32
{
    std::vector<cv::Mat> vDesc;
    vDesc.reserve(Descriptors.rows);
    for (int j = 0; j < Descriptors.rows; j++) vDesc.push_back(Descriptors.rowRange(j, j + 1));
    return vDesc;
}

static bool load(const char* path, int n, cv::Mat& m)
{
    m.create(n, 32);
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    const bool ok = fread(m.data, 1, (size_t)n * 32, f) == (size_t)n * 32;
    fclose(f);
    return ok;
}

static void print(const char* tag, const DBoW2::BowVector& v, const DBoW2::FeatureVector& fv)
{
    printf("%s bow %zu", tag, v.size());
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it) printf(" %u:%a", it->first, it->second);
    printf("\n%s fv %zu", tag, fv.size());
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        printf(" %u[", it->first);
        for (size_t i = 0; i < it->second.size(); i++) printf(i ? ",%u" : "%u", it->second[i]);
        printf("]");
    }
    printf("\n");
}

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    ORB_SLAM3::ORBVocabulary* mpVocabulary = new ORB_SLAM3::ORBVocabulary();
    if (mpVocabulary->loadFromTextFile("/nonexistent/voc.txt")) return 3; // "Wrong path to vocabulary" (System.cc:83-88)
    if (!mpVocabulary->loadFromTextFile(argv[1])) return 4;
    printf("words %u k %d L %d\n", mpVocabulary->size(), mpVocabulary->getBranchingFactor(), mpVocabulary->getDepthLevels());
    const int n = atoi(argv[3]), levelsup = atoi(argv[4]);
    cv::Mat mDescriptors;
    if (!load(argv[2], n, mDescriptors)) return 5;
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
    if (mBowVec.empty()) { // Frame::ComputeBoW
        std::vector<cv::Mat> vCurrentDesc = toDescriptorVector(mDescriptors);
        mpVocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, levelsup);
    }
    print("A", mBowVec, mFeatVec);
    if (argc > 6) {
        const int n2 = atoi(argv[6]);
        cv::Mat d2;
        if (!load(argv[5], n2, d2)) return 5;
        // two threads at once, as Tracking and LocalMapping do; then the score of the two vectors
        DBoW2::BowVector b1, b2;
        DBoW2::FeatureVector f1, f2;
        std::thread t1([&] { for (int k = 0; k < 20; k++) mpVocabulary->transform(toDescriptorVector(mDescriptors), b1, f1, levelsup); });
        std::thread t2([&] { for (int k = 0; k < 20; k++) mpVocabulary->transform(toDescriptorVector(d2), b2, f2, levelsup); });
        t1.join();
        t2.join();
        if (b1 != mBowVec || f1 != mFeatVec) return 6;
        print("B", b2, f2);
        printf("score %a %a\n", mpVocabulary->score(mBowVec, b2), mpVocabulary->score(mBowVec, mBowVec));
        // the handle form: asynchronous, filled on request
        orbfe_bow* h = mpVocabulary->computeBoW(d2.data, n2, levelsup);
        if (!h) return 7;
        DBoW2::BowVector b3;
        DBoW2::FeatureVector f3;
        ORB_SLAM3::ORBVocabulary::fill(h, b3, f3);
        orbfe_bow_destroy(h);
        if (b3 != b2 || f3 != f2) return 8;
    }
    DBoW2::BowVector e1;
    DBoW2::FeatureVector e2;
    mpVocabulary->transform(std::vector<cv::Mat>(), e1, e2, levelsup); // no features: both vectors empty
    if (!e1.empty() || !e2.empty()) return 9;
    delete mpVocabulary;
    printf("ok\n");
    return 0;
}
