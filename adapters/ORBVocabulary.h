/*
 * adapters/ORBVocabulary.h -- ORB_SLAM3::ORBVocabulary (include/ORBVocabulary.h:30-31: DBoW2::TemplatedVocabulary<FORB::TDescriptor,
 * FORB>) with the members ORB-SLAM3 calls, the tree and ComputeBoW's work served by liborbfe.so (include/orbfe.h, orbfe_vocab_* /
 * orbfe_bow_*), so that
 *
 *     mpVocabulary->loadFromTextFile(strVocFile)                                   src/System.cc:82
 *     mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)               src/Frame.cc:724-731, src/KeyFrame.cc:105-114
 *     mpVoc->score(pKF->mBowVec, pKFi->mBowVec), mpVoc->size()                     src/KeyFrameDatabase.cc:71, :162 ...
 *
 * compile unchanged.  transform() runs TemplatedVocabulary::transform(features, v, fv, levelsup)
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1192) on the device -- descent, BowVector::addWeight / normalize and
 * FeatureVector::addFeature, bit-identical to the reference's maps -- and fills the two std::maps from the handle's host copy.
 *
 * Beyond the reference's interface (what a caller that keeps the front end on the device uses instead):
 *     computeBoW(desc, n, levelsup)      -> orbfe_bow* : asynchronous, `desc` may be the extractor's DEVICE descriptors
 *                                                        (ORBextractor::DeviceDescriptors); pass the handle's vector to
 *                                                        ORBmatcher's batched SearchByBoW as an orbfe_fv (orbfe_bow_fv) and the
 *                                                        relocalisation runs extract -> ComputeBoW -> SearchByBoW x candidates
 *                                                        without a host copy of the vector in between;
 *     fill(bow, v, fv)                   the maps from a handle, whenever the host wants them.
 *
 * transform() is const and thread-safe like the reference's (Tracking and LocalMapping call it concurrently): every call
 * borrows a handle from a small pool.
 */
#ifndef ORBFE_ADAPTER_ORBVOCABULARY_H
#define ORBFE_ADAPTER_ORBVOCABULARY_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/orbfe.h"
#include "cv_standins.h"
#ifndef ORBFE_HAVE_ORBSLAM
#include "orbslam_standins.h"
#endif

namespace ORB_SLAM3 {

class ORBVocabulary {
public:
    explicit ORBVocabulary(int device = 0) : device_(device) {}
    ~ORBVocabulary()
    {
        for (orbfe_bow* b : pool_) orbfe_bow_destroy(b);
        if (voc_) orbfe_vocab_free(voc_);
    }
    ORBVocabulary(const ORBVocabulary&) = delete;
    ORBVocabulary& operator=(const ORBVocabulary&) = delete;

    // TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1423): false when the file cannot be read or parsed
    bool loadFromTextFile(const std::string& filename)
    {
        orbfe_vocab_dev* v = nullptr;
        int k = 0, L = 0, nw = 0;
        if (orbfe_vocab_load_text(&v, device_, filename.c_str(), &k, &L, &nw) != 0) return false;
        std::lock_guard<std::mutex> g(m_);
        for (orbfe_bow* b : pool_) orbfe_bow_destroy(b);
        pool_.clear();
        if (voc_) orbfe_vocab_free(voc_);
        voc_ = v;
        k_ = k;
        L_ = L;
        nwords_ = nw;
        (void)orbfe_vocab_get_types(voc_, &weighting_, &scoring_);
        return true;
    }
    // an already built tree in CSR form (tests; a caller with its own loader)
    bool upload(const orbfe_vocab& tree, int weighting = 0, int scoring = 0)
    {
        orbfe_vocab_dev* v = nullptr;
        if (orbfe_vocab_upload(&v, device_, &tree) != 0) return false;
        (void)orbfe_vocab_set_types(v, weighting, scoring);
        std::lock_guard<std::mutex> g(m_);
        for (orbfe_bow* b : pool_) orbfe_bow_destroy(b);
        pool_.clear();
        if (voc_) orbfe_vocab_free(voc_);
        voc_ = v;
        L_ = tree.L;
        nwords_ = 0;
        for (int i = 0; i < tree.nnodes; i++) nwords_ += tree.node_word[i] >= 0;
        weighting_ = weighting;
        scoring_ = scoring;
        return true;
    }
    unsigned int size() const { return (unsigned int)nwords_; } // number of words (:122, m_words.size())
    bool empty() const { return nwords_ == 0; }
    int getBranchingFactor() const { return k_; }
    int getDepthLevels() const { return L_; }

    // transform(features, v, fv, levelsup) (:1127-1192) -- Frame::ComputeBoW passes Converter::toDescriptorVector(mDescriptors):
    // one 1 x 32 CV_8U row per feature
    void transform(const std::vector<cv::Mat>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const
    {
        v.clear();
        fv.clear();
        if (empty() || features.empty()) return; // (:1134-1137)
        std::vector<uint8_t> rows(features.size() * 32);
        for (size_t i = 0; i < features.size(); i++) std::memcpy(&rows[i * 32], features[i].ptr(0), 32);
        transform(rows.data(), (int)features.size(), v, fv, levelsup);
    }
    // the same on the descriptor matrix itself (rows of 32 bytes; host or device pointer)
    void transform(const uint8_t* desc, int n, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const
    {
        v.clear();
        fv.clear();
        if (empty() || n <= 0) return;
        orbfe_bow* b = borrow(n);
        const int r = orbfe_compute_bow(b, desc, n, levelsup);
        if (r == 0) {
            try {
                fill(b, v, fv);
            } catch (...) { // (the borrowed handle goes back whatever fill() says)
                give_back(b);
                throw;
            }
        }
        give_back(b);
        if (r != 0) throw std::runtime_error(std::string("ORBVocabulary::transform: ") + orbfe_error_string(r));
    }
    // asynchronous ComputeBoW into a handle the caller keeps (see the header comment); orbfe_bow_destroy when done
    orbfe_bow* computeBoW(const uint8_t* desc, int n, int levelsup, orbfe_bow* reuse = nullptr) const
    {
        orbfe_bow* b = reuse;
        if (!b && orbfe_bow_create(&b, voc_, n > 0 ? n : 1) != 0) return nullptr;
        if (orbfe_compute_bow(b, desc, n, levelsup) != 0) {
            if (!reuse) orbfe_bow_destroy(b);
            return nullptr;
        }
        return b;
    }
    static void fill(orbfe_bow* b, DBoW2::BowVector& v, DBoW2::FeatureVector& fv)
    {
        v.clear();
        fv.clear();
        orbfe_bow_view w;
        const int r = orbfe_bow_host(b, &w);
        if (r != 0) throw std::runtime_error(std::string("ORBVocabulary: ") + orbfe_error_string(r));
        for (int i = 0; i < w.nw; i++) v.insert(v.end(), DBoW2::BowVector::value_type(w.word_ids[i], w.word_values[i])); // ascending ids
        for (int s = 0; s < w.nn; s++) {
            DBoW2::FeatureVector::iterator it = fv.insert(fv.end(), DBoW2::FeatureVector::value_type(w.node_ids[s], std::vector<unsigned int>()));
            it->second.assign(w.indices + w.offsets[s], w.indices + w.offsets[s + 1]);
        }
    }

    // score(v1, v2): m_scoring_object->score (ScoringObject.cpp).  Host code of KeyFrameDatabase, outside the device path; the
    // three scorings whose formula is a plain merge-join are here (ORBvoc.txt is L1_NORM), the others are refused.
    double score(const DBoW2::BowVector& v1, const DBoW2::BowVector& v2) const
    {
        if (scoring_ != 0 && scoring_ != 1 && scoring_ != 5)
            throw std::runtime_error("ORBVocabulary::score: only L1_NORM, L2_NORM and DOT_PRODUCT are provided");
        DBoW2::BowVector::const_iterator a = v1.begin(), b = v2.begin();
        double s = 0;
        while (a != v1.end() && b != v2.end()) {
            if (a->first == b->first) {
                const double vi = a->second, wi = b->second;
                s += scoring_ == 0 ? std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi) : vi * wi; // ScoringObject.cpp:41 / :91
                ++a;
                ++b;
            } else if (a->first < b->first) a = v1.lower_bound(b->first);
            else b = v2.lower_bound(a->first);
        }
        if (scoring_ == 0) return -s / 2.0;                               // ScoringObject.cpp:65
        if (scoring_ == 1) return s >= 1 ? 1.0 : 1.0 - std::sqrt(1.0 - s); // ScoringObject.cpp:114-117
        return s;                                                          // DotProductScoring
    }
    orbfe_vocab_dev* handle() const { return voc_; }

private:
    orbfe_bow* borrow(int n) const
    {
        {
            std::lock_guard<std::mutex> g(m_);
            for (size_t i = 0; i < pool_.size(); i++)
                if (caps_[i] >= n) {
                    orbfe_bow* b = pool_[i];
                    pool_.erase(pool_.begin() + (long)i);
                    const int cap = caps_[i];
                    caps_.erase(caps_.begin() + (long)i);
                    borrowedCap_.push_back(std::make_pair(b, cap));
                    return b;
                }
        }
        const int cap = n < 2048 ? 2048 : (n > 65535 ? 65535 : n + n / 4 > 65535 ? 65535 : n + n / 4);
        orbfe_bow* b = nullptr;
        const int r = orbfe_bow_create(&b, voc_, cap);
        if (r != 0) throw std::runtime_error(std::string("ORBVocabulary: ") + orbfe_error_string(r));
        std::lock_guard<std::mutex> g(m_);
        borrowedCap_.push_back(std::make_pair(b, cap));
        return b;
    }
    void give_back(orbfe_bow* b) const
    {
        std::lock_guard<std::mutex> g(m_);
        int cap = 0;
        for (size_t i = 0; i < borrowedCap_.size(); i++)
            if (borrowedCap_[i].first == b) {
                cap = borrowedCap_[i].second;
                borrowedCap_.erase(borrowedCap_.begin() + (long)i);
                break;
            }
        if (pool_.size() < 8) {
            pool_.push_back(b);
            caps_.push_back(cap);
        } else orbfe_bow_destroy(b);
    }
    int device_ = 0;
    orbfe_vocab_dev* voc_ = nullptr;
    int k_ = 0, L_ = 0, nwords_ = 0, weighting_ = 0, scoring_ = 0;
    mutable std::mutex m_;
    mutable std::vector<orbfe_bow*> pool_;
    mutable std::vector<int> caps_;
    mutable std::vector<std::pair<orbfe_bow*, int>> borrowedCap_;
};

} // namespace ORB_SLAM3

#endif
