// Drives adapters/ORBmatcher.h the way Tracking / LocalMapping / LoopClosing call ORBmatcher, on object graphs
// (stand-in KeyFrame / Frame / MapPoint, adapters/orbslam_standins.h) built from a scenario file, and writes what each
// method left in the objects to a result file.  tests/test_gpu_matcher_adapter.py generates the scenarios, derives
// the expected object state with the oracle from its OWN flattening of the same data, and compares.
//   test_matcher_adapter scenario.bin result.bin
// File format: records of { u32 name_len, name, u32 dtype (0 u8, 1 i32, 2 f32), u32 count, data }.
#define ORBFE_NO_OPENCV 1
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>

#include "ORBmatcher.h"

using namespace ORB_SLAM3;

struct Arr {
    int dtype = 0;
    std::vector<uint8_t> raw;
    size_t count = 0;
    const uint8_t* u8() const { return raw.data(); }
    const int32_t* i32() const { return reinterpret_cast<const int32_t*>(raw.data()); }
    const float* f32() const { return reinterpret_cast<const float*>(raw.data()); }
};
static std::map<std::string, Arr> g_in;
static FILE* g_out = nullptr;

static bool load(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    for (;;) {
        uint32_t nl = 0, dt = 0, cnt = 0;
        if (fread(&nl, 4, 1, f) != 1) break;
        std::string name(nl, ' ');
        if (fread(&name[0], 1, nl, f) != nl || fread(&dt, 4, 1, f) != 1 || fread(&cnt, 4, 1, f) != 1) return false;
        Arr a;
        a.dtype = (int)dt;
        a.count = cnt;
        a.raw.resize((size_t)cnt * (dt == 0 ? 1 : 4));
        if (!a.raw.empty() && fread(a.raw.data(), 1, a.raw.size(), f) != a.raw.size()) return false;
        g_in[name] = a;
    }
    fclose(f);
    return true;
}
static const Arr& in(const std::string& n)
{
    auto it = g_in.find(n);
    if (it == g_in.end()) {
        fprintf(stderr, "missing array %s\n", n.c_str());
        exit(3);
    }
    return it->second;
}
static bool has(const std::string& n) { return g_in.count(n) != 0; }
static void put(const std::string& name, int dtype, const void* data, size_t count)
{
    const uint32_t nl = (uint32_t)name.size(), dt = (uint32_t)dtype, cnt = (uint32_t)count;
    fwrite(&nl, 4, 1, g_out);
    fwrite(name.data(), 1, nl, g_out);
    fwrite(&dt, 4, 1, g_out);
    fwrite(&cnt, 4, 1, g_out);
    fwrite(data, dtype == 0 ? 1 : 4, count, g_out);
}
static void put_i(const std::string& name, const std::vector<int32_t>& v) { put(name, 1, v.data(), v.size()); }

static std::vector<std::unique_ptr<MapPoint>> g_points; // owns every MapPoint of the run
static MapPoint* new_point(long id)
{
    g_points.emplace_back(new MapPoint());
    g_points.back()->mnId = (unsigned long)id;
    return g_points.back().get();
}
static void set_keys(std::vector<cv::KeyPoint>& v, int n, const float* x, const float* y, const float* ang, const int32_t* oct)
{
    v.resize(n);
    for (int i = 0; i < n; i++) {
        v[i].pt.x = x ? x[i] : 0.f;
        v[i].pt.y = y ? y[i] : 0.f;
        v[i].angle = ang ? ang[i] : 0.f;
        v[i].octave = oct ? oct[i] : 0;
        v[i].size = 31.f;
        v[i].response = 0.f;
        v[i].class_id = -1;
    }
}
static void set_desc(cv::Mat& D, int n, const uint8_t* d)
{
    D.create(std::max(n, 1), 32);
    if (n) memcpy(D.data, d, (size_t)n * 32);
    D.rows = n;
}
static void set_featvec(DBoW2::FeatureVector& fv, int n, const int32_t* node)
{
    for (int i = 0; i < n; i++)
        if (node[i] >= 0) fv[(DBoW2::NodeId)node[i]].push_back((unsigned)i);
}
// state per feature: 0 none, 1 good MapPoint, 2 bad MapPoint, 3 good MapPoint without observations
static void set_points(std::vector<MapPoint*>& v, int n, const int32_t* state, long idBase)
{
    v.assign(n, nullptr);
    for (int i = 0; i < n; i++)
        if (state[i]) {
            v[i] = new_point(idBase + i);
            v[i]->mbBad = state[i] == 2;
            v[i]->nObs = state[i] == 3 ? 0 : 2;
        }
}
static cv::Matx33f m33(const float* p)
{
    cv::Matx33f m;
    for (int i = 0; i < 9; i++) m.val[i] = p[i];
    return m;
}
static cv::Matx31f m31(const float* p)
{
    cv::Matx31f m;
    for (int i = 0; i < 3; i++) m.val[i] = p[i];
    return m;
}

static cv::Mat mat32(const float* v, int r, int c)
{
    cv::Mat m(r, c, CV_32F);
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) m.at<float>(i, j) = v[i * c + j];
    return m;
}

static void test_bow_kf_f(const std::string& P)
{
    const int n1 = (int)in(P + "a1").count, n2 = (int)in(P + "a2").count, Nleft = in(P + "Nleft").i32()[0];
    KeyFrame kf;
    kf.N = n1;
    set_keys(kf.mvKeysUn, n1, nullptr, nullptr, in(P + "a1").f32(), nullptr);
    set_desc(kf.mDescriptors, n1, in(P + "d1").u8());
    set_featvec(kf.mFeatVec, n1, in(P + "node1").i32());
    set_points(kf.mvpMapPoints, n1, in(P + "mp1").i32(), 0);
    Frame F;
    GeometricCamera cam2;
    F.N = n2;
    F.Nleft = Nleft;
    const float* a2 = in(P + "a2").f32();
    if (Nleft == -1) {
        set_keys(F.mvKeys, n2, nullptr, nullptr, a2, nullptr);
    } else {
        set_keys(F.mvKeys, Nleft, nullptr, nullptr, a2, nullptr);
        set_keys(F.mvKeysRight, n2 - Nleft, nullptr, nullptr, a2 + Nleft, nullptr);
        F.mpCamera2 = &cam2;
    }
    set_desc(F.mDescriptors, n2, in(P + "d2").u8());
    set_featvec(F.mFeatVec, n2, in(P + "node2").i32());
    ORBmatcher matcher(in(P + "ratio").f32()[0], in(P + "ori").i32()[0] != 0);
    std::vector<MapPoint*> vpMapPointMatches;
    const int nm = matcher.SearchByBoW(&kf, F, vpMapPointMatches);
    std::vector<int32_t> out(n2, -1);
    for (int i = 0; i < n2; i++)
        if (vpMapPointMatches[i]) out[i] = (int32_t)vpMapPointMatches[i]->mnId; // id == keyframe feature index
    put_i(P + "match", out);
    put_i(P + "n", std::vector<int32_t>(1, nm));
    // keyframe handles (round 4): the search above created the keyframe's handle; the same search again only hits it, and a
    // search with changed MapPoint flags still hits it (the flags travel with the call); handles off gives the same results
    orbfe_adapter::KeyFrameHandles& H = orbfe_adapter::keyframe_handles();
    const long c0 = H.creates, h0 = H.hits;
    auto run = [&](std::vector<int32_t>& o) {
        std::vector<MapPoint*> v;
        o.push_back(matcher.SearchByBoW(&kf, F, v));
        for (int i = 0; i < n2; i++) o.push_back(v[i] ? (int32_t)v[i]->mnId : -1);
    };
    std::vector<int32_t> ra1, ra2, rb1, rb2;
    run(ra1);
    std::vector<MapPoint*> saved = kf.mvpMapPoints;
    for (int i = 0; i < n1; i += 3) kf.mvpMapPoints[i] = nullptr; // (the map changed)
    run(ra2);
    const long dc = H.creates - c0, dh = H.hits - h0;
    orbfe_adapter::use_keyframe_handles() = false;
    run(rb2);
    kf.mvpMapPoints = saved;
    run(rb1);
    orbfe_adapter::use_keyframe_handles() = true;
    put_i(P + "kfhandles", std::vector<int32_t>{(int32_t)dc, (int32_t)dh, (ra1 == rb1 && ra2 == rb2 && ra1[0] == nm) ? 1 : 0});
    // a KeyFrame built from a frame the motion model tracked has an EMPTY mFeatVec until LocalMapping's ComputeBoW
    // (src/LocalMapping.cc:380) while Tracking already searches it as mpReferenceKF (src/Tracking.cc:3290).  Searched in that
    // window it must behave like the reference -- no common node, no match -- WITHOUT leaving a handle behind; searched again
    // after ComputeBoW it must find what the keyframe above finds, through a handle made then.
    {
        KeyFrame late;
        late.N = n1;
        late.mnId = kf.mnId + 1000;
        late.mvKeysUn = kf.mvKeysUn;
        late.mDescriptors = kf.mDescriptors;
        late.mvpMapPoints = kf.mvpMapPoints;
        const long c1 = H.creates;
        std::vector<MapPoint*> v;
        const int before = matcher.SearchByBoW(&late, F, v);
        const long createdBefore = H.creates - c1;
        late.mFeatVec = kf.mFeatVec; // ComputeBoW
        std::vector<MapPoint*> v2;
        const int after = matcher.SearchByBoW(&late, F, v2);
        bool same = after == nm;
        for (int i = 0; i < n2 && same; i++) same = (v2[i] ? (int32_t)v2[i]->mnId : -1) == out[i];
        put_i(P + "fvlate", std::vector<int32_t>{before, (int32_t)createdBefore, same ? 1 : 0, (int32_t)(H.creates - c1)});
    }
}

static void test_bow_kf_kf(const std::string& P)
{
    const int n1 = (int)in(P + "a1").count, n2 = (int)in(P + "a2").count;
    KeyFrame k1, k2;
    k1.N = n1;
    k2.N = n2;
    set_keys(k1.mvKeysUn, n1, nullptr, nullptr, in(P + "a1").f32(), nullptr);
    set_keys(k2.mvKeysUn, n2, nullptr, nullptr, in(P + "a2").f32(), nullptr);
    set_desc(k1.mDescriptors, n1, in(P + "d1").u8());
    set_desc(k2.mDescriptors, n2, in(P + "d2").u8());
    set_featvec(k1.mFeatVec, n1, in(P + "node1").i32());
    set_featvec(k2.mFeatVec, n2, in(P + "node2").i32());
    set_points(k1.mvpMapPoints, n1, in(P + "mp1").i32(), 0);
    set_points(k2.mvpMapPoints, n2, in(P + "mp2").i32(), 0);
    ORBmatcher matcher(in(P + "ratio").f32()[0], in(P + "ori").i32()[0] != 0);
    std::vector<MapPoint*> vpMatches12;
    const int nm = matcher.SearchByBoW(&k1, &k2, vpMatches12);
    std::vector<int32_t> out(n1, -1);
    for (int i = 0; i < n1; i++)
        if (vpMatches12[i]) out[i] = (int32_t)vpMatches12[i]->mnId; // id == index in keyframe 2
    put_i(P + "match", out);
    put_i(P + "n", std::vector<int32_t>(1, nm));
    orbfe_adapter::KeyFrameHandles& H = orbfe_adapter::keyframe_handles();
    const long c0 = H.creates, h0 = H.hits;
    auto run = [&](std::vector<int32_t>& o) {
        std::vector<MapPoint*> v;
        o.push_back(matcher.SearchByBoW(&k1, &k2, v));
        for (int i = 0; i < n1; i++) o.push_back(v[i] ? (int32_t)v[i]->mnId : -1);
    };
    std::vector<int32_t> a1, b1;
    run(a1);
    const long dc = H.creates - c0, dh = H.hits - h0;
    orbfe_adapter::use_keyframe_handles() = false;
    run(b1);
    orbfe_adapter::use_keyframe_handles() = true;
    put_i(P + "kfhandles", std::vector<int32_t>{(int32_t)dc, (int32_t)dh, (a1 == b1 && a1[0] == nm) ? 1 : 0});
}

static void fill_kf_geom(KeyFrame& k, const std::string& P, const char* sfx)
{
    const std::string s(sfx);
    const int n = (int)in(P + "a" + s).count;
    k.N = n;
    set_keys(k.mvKeysUn, n, in(P + "x" + s).f32(), in(P + "y" + s).f32(), in(P + "a" + s).f32(), in(P + "oct" + s).i32());
    set_desc(k.mDescriptors, n, in(P + "d" + s).u8());
    k.mvuRight.assign(in(P + "ur" + s).f32(), in(P + "ur" + s).f32() + n);
    const Arr& sf = in(P + "sf");
    k.mvScaleFactors.assign(sf.f32(), sf.f32() + sf.count);
    k.mvLevelSigma2.resize(sf.count);
    k.mvInvLevelSigma2.resize(sf.count);
    for (size_t i = 0; i < sf.count; i++) {
        k.mvLevelSigma2[i] = k.mvScaleFactors[i] * k.mvScaleFactors[i];
        k.mvInvLevelSigma2[i] = 1.0f / k.mvLevelSigma2[i];
    }
}

static void test_tri(const std::string& P)
{
    KeyFrame k1, k2;
    GeometricCamera c1, c2;
    fill_kf_geom(k1, P, "1");
    fill_kf_geom(k2, P, "2");
    set_featvec(k1.mFeatVec, k1.N, in(P + "node1").i32());
    set_featvec(k2.mFeatVec, k2.N, in(P + "node2").i32());
    set_points(k1.mvpMapPoints, k1.N, in(P + "mp1").i32(), 0);
    set_points(k2.mvpMapPoints, k2.N, in(P + "mp2").i32(), 0);
    c1.mvParameters.assign(in(P + "cam1").f32(), in(P + "cam1").f32() + 4);
    c2.mvParameters.assign(in(P + "cam2").f32(), in(P + "cam2").f32() + 4);
    k1.mpCamera = &c1;
    k2.mpCamera = &c2;
    k1.Rcw = m33(in(P + "R1").f32());
    k1.tcw = m31(in(P + "t1").f32());
    k1.Ow = m31(in(P + "O1").f32());
    k2.Rcw = m33(in(P + "R2").f32());
    k2.tcw = m31(in(P + "t2").f32());
    ORBmatcher matcher(0.6f, in(P + "ori").i32()[0] != 0);
    std::vector<std::pair<size_t, size_t>> pairs;
    const int nm = matcher.SearchForTriangulation_(&k1, &k2, cv::Matx33f(), pairs, in(P + "stereo").i32()[0] != 0,
                                                   in(P + "coarse").i32()[0] != 0);
    std::vector<int32_t> out;
    for (auto& pr : pairs) {
        out.push_back((int32_t)pr.first);
        out.push_back((int32_t)pr.second);
    }
    put_i(P + "pairs", out);
    put_i(P + "n", std::vector<int32_t>(1, nm));
    put(P + "F12", 2, matcher.lastF12.val, 9);
    const float ep[2] = {matcher.lastEp.x, matcher.lastEp.y};
    put(P + "ep", 2, ep, 2);
    {
        orbfe_adapter::KeyFrameHandles& H = orbfe_adapter::keyframe_handles();
        const long c0 = H.creates, h0 = H.hits;
        auto run = [&](std::vector<int32_t>& o) {
            std::vector<std::pair<size_t, size_t>> pr;
            o.push_back(matcher.SearchForTriangulation_(&k1, &k2, cv::Matx33f(), pr, in(P + "stereo").i32()[0] != 0,
                                                        in(P + "coarse").i32()[0] != 0));
            for (auto& q : pr) {
                o.push_back((int32_t)q.first);
                o.push_back((int32_t)q.second);
            }
        };
        std::vector<int32_t> a1, a2, b1, b2;
        run(a1);
        std::vector<MapPoint*> s1 = k1.mvpMapPoints, s2 = k2.mvpMapPoints;
        for (int i = 0; i < k1.N; i += 2) k1.mvpMapPoints[i] = nullptr; // more features without a MapPoint: more rows search
        for (int i = 1; i < k2.N; i += 2) k2.mvpMapPoints[i] = nullptr;
        run(a2);
        const long dc = H.creates - c0, dh = H.hits - h0;
        orbfe_adapter::use_keyframe_handles() = false;
        run(b2);
        k1.mvpMapPoints = s1;
        k2.mvpMapPoints = s2;
        run(b1);
        orbfe_adapter::use_keyframe_handles() = true;
        put_i(P + "kfhandles", std::vector<int32_t>{(int32_t)dc, (int32_t)dh, (a1 == b1 && a2 == b2 && a1[0] == nm) ? 1 : 0});
    }
    std::vector<std::pair<size_t, size_t>> pairs3; // pinhole cameras: Pinhole::matchAndtriangulate accepts nothing
    std::vector<cv::Mat> pts3;
    put_i(P + "n3d", std::vector<int32_t>(1, matcher.SearchForTriangulation(&k1, &k2, cv::Mat(), pairs3, false, pts3)));
}

// fisheye keyframes: a monocular pair, or a two-camera rig (features [0, NLeft) from the left camera)
static void test_tri_kb8(const std::string& P)
{
    KeyFrame k1, k2;
    GeometricCamera c[4]; // 1L, 1R, 2L, 2R
    fill_kf_geom(k1, P, "1");
    fill_kf_geom(k2, P, "2");
    set_featvec(k1.mFeatVec, k1.N, in(P + "node1").i32());
    set_featvec(k2.mFeatVec, k2.N, in(P + "node2").i32());
    set_points(k1.mvpMapPoints, k1.N, in(P + "mp1").i32(), 0);
    set_points(k2.mvpMapPoints, k2.N, in(P + "mp2").i32(), 0);
    const char* names[4] = {"cam1L", "cam1R", "cam2L", "cam2R"};
    const bool rig = in(P + "rig").i32()[0] != 0;
    for (int i = 0; i < 4; i++) {
        if (!rig && (i & 1)) continue;
        c[i].mvParameters.assign(in(P + names[i]).f32(), in(P + names[i]).f32() + 8);
        c[i].mnType = 1;
    }
    k1.mpCamera = &c[0];
    k2.mpCamera = &c[2];
    k1.Rcw = m33(in(P + "R1").f32()); k1.tcw = m31(in(P + "t1").f32()); k1.Ow = m31(in(P + "O1").f32());
    k2.Rcw = m33(in(P + "R2").f32()); k2.tcw = m31(in(P + "t2").f32());
    if (rig) {
        KeyFrame* ks[2] = {&k1, &k2};
        for (int j = 0; j < 2; j++) {
            KeyFrame& k = *ks[j];
            const std::string S = j ? "2" : "1";
            k.NLeft = in(P + "NLeft" + S).i32()[0];
            k.mvKeys.assign(k.mvKeysUn.begin(), k.mvKeysUn.begin() + k.NLeft);
            k.mvKeysRight.assign(k.mvKeysUn.begin() + k.NLeft, k.mvKeysUn.end());
            set_keys(k.mvKeysUn, k.NLeft, nullptr, nullptr, nullptr, nullptr); // (a rig's mvKeysUn holds NLeft entries, :857)
            k.mpCamera2 = &c[2 * j + 1];
            k.RcwR = m33(in(P + "R" + S + "R").f32());
            k.tcwR = m31(in(P + "t" + S + "R").f32());
        }
    }
    ORBmatcher matcher(0.6f, in(P + "ori").i32()[0] != 0);
    std::vector<std::pair<size_t, size_t>> pairs;
    const int nm = matcher.SearchForTriangulation_(&k1, &k2, cv::Matx33f(), pairs, in(P + "stereo").i32()[0] != 0,
                                                   in(P + "coarse").i32()[0] != 0);
    std::vector<int32_t> out;
    for (auto& pr : pairs) {
        out.push_back((int32_t)pr.first);
        out.push_back((int32_t)pr.second);
    }
    put_i(P + "pairs", out);
    put_i(P + "n", std::vector<int32_t>(1, nm));
    put(P + "R12", 2, matcher.lastR12.data(), matcher.lastR12.size());
    put(P + "t12", 2, matcher.lastT12.data(), matcher.lastT12.size());
    const float ep[2] = {matcher.lastEp.x, matcher.lastEp.y};
    put(P + "ep", 2, ep, 2);
    // the overload that also triangulates (src/ORBmatcher.cc:1452-1641) on the same keyframes
    std::vector<std::pair<size_t, size_t>> pairs3;
    std::vector<cv::Mat> pts3;
    const int n3 = matcher.SearchForTriangulation(&k1, &k2, cv::Mat(), pairs3, false, pts3);
    std::vector<int32_t> out3;
    std::vector<float> x3;
    for (size_t k = 0; k < pairs3.size(); k++) {
        out3.push_back((int32_t)pairs3[k].first);
        out3.push_back((int32_t)pairs3[k].second);
        for (int i = 0; i < 3; i++) x3.push_back(pts3[k].at<float>(i));
    }
    put_i(P + "pairs3d", out3);
    put(P + "points3d", 2, x3.data(), x3.size());
    put_i(P + "n3d", std::vector<int32_t>(1, n3));
}

static void fill_frame(Frame& F, const std::string& P)
{
    const int n = (int)in(P + "kx").count;
    F.N = n;
    set_keys(F.mvKeysUn, n, in(P + "kx").f32(), in(P + "ky").f32(), in(P + "ang").f32(), in(P + "oct").i32());
    F.mvKeys = F.mvKeysUn;
    if (has(P + "Nleft") && in(P + "Nleft").i32()[0] != -1) { // two-camera rig: mvKeys ++ mvKeysRight, mvKeysUn is not read
        F.Nleft = in(P + "Nleft").i32()[0];
        F.mvKeysRight.assign(F.mvKeys.begin() + F.Nleft, F.mvKeys.end());
        F.mvKeys.resize(F.Nleft);
        set_keys(F.mvKeysUn, n, nullptr, nullptr, nullptr, nullptr);
    }
    set_desc(F.mDescriptors, n, in(P + "desc").u8());
    if (has(P + "uright")) F.mvuRight.assign(in(P + "uright").f32(), in(P + "uright").f32() + n);
    else F.mvuRight.assign(n, -1.f);
    set_points(F.mvpMapPoints, n, in(P + "fstate").i32(), 100000);
    const float* g = in(P + "grid").f32(); // minX, minY, maxX, maxY, gridWInv, gridHInv
    F.mnMinX = g[0]; F.mnMinY = g[1]; F.mnMaxX = g[2]; F.mnMaxY = g[3];
    F.mfGridElementWidthInv = g[4]; F.mfGridElementHeightInv = g[5];
    const Arr& sf = in(P + "sf");
    F.mvScaleFactors.assign(sf.f32(), sf.f32() + sf.count);
}
static void dump_frame_points(const Frame& F, const std::string& name)
{
    std::vector<int32_t> out(F.N, -1);
    for (int i = 0; i < F.N; i++)
        if (F.mvpMapPoints[i]) out[i] = (int32_t)F.mvpMapPoints[i]->mnId;
    put_i(name, out);
}

static void test_proj_local(const std::string& P)
{
    Frame F;
    fill_frame(F, P);
    const int m = (int)in(P + "px").count;
    std::vector<MapPoint*> pts(m);
    const int32_t* pstate = in(P + "pstate").i32(); // bit0 in view, bit1 bad, bit2 no observations
    for (int k = 0; k < m; k++) {
        MapPoint* p = pts[k] = new_point(k);
        p->mbTrackInView = pstate[k] & 1;
        p->mbBad = (pstate[k] & 2) != 0;
        p->nObs = (pstate[k] & 4) ? 0 : 3;
        p->mTrackProjX = in(P + "px").f32()[k];
        p->mTrackProjY = in(P + "py").f32()[k];
        p->mTrackProjXR = in(P + "pxr").f32()[k];
        p->mTrackViewCos = in(P + "pcos").f32()[k];
        p->mTrackDepth = in(P + "pdepth").f32()[k];
        p->mnTrackScaleLevel = in(P + "plevel").i32()[k];
        if (F.Nleft != -1) { // Frame::isInFrustumChecks for the right camera
            p->mbTrackInViewR = (pstate[k] & 8) != 0;
            p->mTrackProjYR = in(P + "pyR").f32()[k];
            p->mTrackViewCosR = in(P + "pcosR").f32()[k];
            p->mnTrackScaleLevelR = in(P + "plevelR").i32()[k];
        }
        memcpy(p->mDescriptor, in(P + "pdesc").u8() + 32 * (size_t)k, 32);
    }
    if (F.Nleft != -1) {
        F.mvLeftToRightMatch.assign(in(P + "l2r").i32(), in(P + "l2r").i32() + in(P + "l2r").count);
        F.mvRightToLeftMatch.assign(in(P + "r2l").i32(), in(P + "r2l").i32() + in(P + "r2l").count);
    }
    ORBmatcher matcher(in(P + "ratio").f32()[0], true);
    const int nm = matcher.SearchByProjection(F, pts, in(P + "th").f32()[0], in(P + "far").i32()[0] != 0, in(P + "thfar").f32()[0]);
    dump_frame_points(F, P + "points");
    put_i(P + "n", std::vector<int32_t>(1, nm));
}

static void test_proj_last(const std::string& P)
{
    Frame C, L;
    GeometricCamera cam;
    fill_frame(C, P);
    cam.mvParameters.assign(in(P + "cam").f32(), in(P + "cam").f32() + 4);
    C.mpCamera = &cam;
    C.mbf = in(P + "bf").f32()[0];
    C.mb = in(P + "bf").f32()[1];
    C.mRcw_ = m33(in(P + "Rc").f32());
    C.mtcw_ = m31(in(P + "tc").f32());
    L.mRcw_ = m33(in(P + "Rl").f32());
    L.mtcw_ = m31(in(P + "tl").f32());
    if (has(P + "Trl")) C.mTrl = mat32(in(P + "Trl").f32(), 3, 4);
    const int nl = (int)in(P + "loct").count;
    L.N = nl;
    set_keys(L.mvKeysUn, nl, nullptr, nullptr, in(P + "lang").f32(), in(P + "loct").i32());
    L.mvKeys = L.mvKeysUn;
    L.mvpMapPoints.assign(nl, nullptr);
    L.mvbOutlier.assign(nl, false);
    const int32_t* ls = in(P + "lstate").i32(); // 0 none, 1 point, 2 point flagged outlier, 3 point without observations
    for (int i = 0; i < nl; i++) {
        if (!ls[i]) continue;
        MapPoint* p = L.mvpMapPoints[i] = new_point(i);
        p->nObs = ls[i] == 3 ? 0 : 2;
        L.mvbOutlier[i] = ls[i] == 2;
        p->mWorldPos = m31(in(P + "lpos").f32() + 3 * (size_t)i);
        memcpy(p->mDescriptor, in(P + "ldesc").u8() + 32 * (size_t)i, 32);
    }
    ORBmatcher matcher(0.9f, in(P + "ori").i32()[0] != 0);
    const std::vector<MapPoint*> before = C.mvpMapPoints;
    const int nm = matcher.SearchByProjection(C, L, in(P + "th").f32()[0], in(P + "mono").i32()[0] != 0);
    dump_frame_points(C, P + "points");
    put_i(P + "n", std::vector<int32_t>(1, nm));
    // Tracking's pattern on ONE frame (src/Tracking.cc:2817-2827): the last frame at th, again at 2 th, and a third search --
    // with frame handles the frame side was uploaded once (by the search above) and the three searches below only hit the
    // handle; without handles every search uploads it again.  Same results either way.
    orbfe_adapter::FrameHandles& H = orbfe_adapter::frame_handles();
    const float th = in(P + "th").f32()[0];
    const bool mono = in(P + "mono").i32()[0] != 0;
    auto three = [&](std::vector<int32_t>& out) {
        for (float t : {th, 2 * th, th}) {
            C.mvpMapPoints = before;
            out.push_back(matcher.SearchByProjection(C, L, t, mono));
            for (int i = 0; i < C.N; i++) out.push_back(C.mvpMapPoints[i] ? (int32_t)C.mvpMapPoints[i]->mnId : -1);
        }
    };
    const long c0 = H.creates, h0 = H.hits;
    std::vector<int32_t> withHandles, without;
    three(withHandles);
    const long dc = H.creates - c0, dh = H.hits - h0;
    orbfe_adapter::use_frame_handles() = false;
    three(without);
    orbfe_adapter::use_frame_handles() = true;
    put_i(P + "handles", std::vector<int32_t>{(int32_t)dc, (int32_t)dh, withHandles == without ? 1 : 0});
}

static void test_fuse(const std::string& P)
{
    KeyFrame k;
    GeometricCamera cam;
    const int n = (int)in(P + "kx").count;
    k.N = n;
    set_keys(k.mvKeysUn, n, in(P + "kx").f32(), in(P + "ky").f32(), nullptr, in(P + "oct").i32());
    set_desc(k.mDescriptors, n, in(P + "desc").u8());
    k.mvuRight.assign(in(P + "uright").f32(), in(P + "uright").f32() + n);
    set_points(k.mvpMapPoints, n, in(P + "fstate").i32(), 100000);
    for (int i = 0; i < n; i++)
        if (k.mvpMapPoints[i]) k.mvpMapPoints[i]->nObs = in(P + "fobs").i32()[i];
    const float* g = in(P + "grid").f32();
    k.mnMinX = g[0]; k.mnMinY = g[1]; k.mnMaxX = g[2]; k.mnMaxY = g[3];
    k.mfGridElementWidthInv = g[4]; k.mfGridElementHeightInv = g[5];
    const Arr& sf = in(P + "sf");
    k.mvScaleFactors.assign(sf.f32(), sf.f32() + sf.count);
    k.mvInvLevelSigma2.resize(sf.count);
    for (size_t i = 0; i < sf.count; i++) k.mvInvLevelSigma2[i] = 1.0f / (k.mvScaleFactors[i] * k.mvScaleFactors[i]);
    k.mnScaleLevels = (int)sf.count;
    k.mfLogScaleFactor = in(P + "logsf").f32()[0];
    cam.mvParameters.assign(in(P + "cam").f32(), in(P + "cam").f32() + 4);
    k.mpCamera = &cam;
    k.mbf = in(P + "bf").f32()[0];
    k.Rcw = m33(in(P + "R").f32());
    k.tcw = m31(in(P + "t").f32());
    k.Ow = m31(in(P + "O").f32());
    GeometricCamera cam2;
    bool bRight = false;
    if (has(P + "NLeft")) { // two-camera rig: mvKeys ++ mvKeysRight, the right camera's pose and model
        k.NLeft = in(P + "NLeft").i32()[0];
        k.mvKeys.assign(k.mvKeysUn.begin(), k.mvKeysUn.begin() + k.NLeft);
        k.mvKeysRight.assign(k.mvKeysUn.begin() + k.NLeft, k.mvKeysUn.end());
        set_keys(k.mvKeysUn, n, nullptr, nullptr, nullptr, nullptr);
        cam2.mvParameters.assign(in(P + "cam2").f32(), in(P + "cam2").f32() + 4);
        k.mpCamera2 = &cam2;
        k.RcwR = m33(in(P + "RR").f32());
        k.tcwR = m31(in(P + "tR").f32());
        k.OwR = m31(in(P + "OR").f32());
        bRight = in(P + "bRight").i32()[0] != 0;
    }
    const int m = (int)in(P + "pstate").count;
    std::vector<MapPoint*> pts(m, nullptr);
    const int32_t* ps = in(P + "pstate").i32(); // 0 null, 1 ok, 2 bad, 3 already observed in the keyframe
    const int32_t* dup = has(P + "pdup") ? in(P + "pdup").i32() : nullptr; // the same point twice in the list (>= 0: of which entry)
    for (int q = 0; q < m; q++) {
        if (!ps[q]) continue;
        if (dup && dup[q] >= 0) {
            pts[q] = pts[dup[q]];
            continue;
        }
        MapPoint* p = pts[q] = new_point(q);
        p->mbBad = ps[q] == 2;
        if (ps[q] == 3) p->mObservations[&k] = 0;
        p->nObs = in(P + "pobs").i32()[q];
        p->mWorldPos = m31(in(P + "ppos").f32() + 3 * (size_t)q);
        p->mNormalVector = m31(in(P + "pnormal").f32() + 3 * (size_t)q);
        p->mfMinDistance = in(P + "pdist").f32()[2 * q];
        p->mfMaxDistance = in(P + "pdist").f32()[2 * q + 1];
        memcpy(p->mDescriptor, in(P + "pdesc").u8() + 32 * (size_t)q, 32);
    }
    const std::vector<MapPoint*> original = k.mvpMapPoints; // the points the keyframe held before
    ORBmatcher matcher;
    const int nf = matcher.Fuse(&k, pts, in(P + "th").f32()[0], bRight);
    // what Fuse left in the objects: per candidate the feature it was added to as an observation (-1 none) and the id
    // of the point that replaced it (-1 none); per keyframe feature the id of the point it holds now and, for the
    // keyframe's original point there, the id of the point that replaced it.  (ids: candidates q, originals 100000 + i)
    std::vector<int32_t> obsIdx(m, -1), replacedBy(m, -1), kfPoint(n, -1), kfReplacedBy(n, -1);
    for (int q = 0; q < m; q++) {
        if (!pts[q]) continue;
        auto it = pts[q]->mObservations.find(&k);
        if (ps[q] != 3 && it != pts[q]->mObservations.end()) obsIdx[q] = it->second;
        if (pts[q]->mpReplaced) replacedBy[q] = (int32_t)pts[q]->mpReplaced->mnId;
    }
    for (int i = 0; i < n; i++)
        if (k.mvpMapPoints[i]) kfPoint[i] = (int32_t)k.mvpMapPoints[i]->mnId;
    for (int i = 0; i < n; i++)
        if (original[i] && original[i]->mpReplaced) kfReplacedBy[i] = (int32_t)original[i]->mpReplaced->mnId;
    put_i(P + "obsIdx", obsIdx);
    put_i(P + "replacedBy", replacedBy);
    put_i(P + "kfPoint", kfPoint);
    put_i(P + "kfReplacedBy", kfReplacedBy);
    put_i(P + "n", std::vector<int32_t>(1, nf));
}

// keyframe with features, grid, scale tables, camera (pinhole members fx.. as KeyFrame has them) and identity-free pose
static void fill_keyframe(KeyFrame& k, GeometricCamera& cam, const std::string& P, const char* sfx = "")
{
    const std::string S(sfx);
    const int n = (int)in(P + "kx" + S).count;
    k.N = n;
    set_keys(k.mvKeysUn, n, in(P + "kx" + S).f32(), in(P + "ky" + S).f32(), has(P + "ang" + S) ? in(P + "ang" + S).f32() : nullptr,
             in(P + "oct" + S).i32());
    set_desc(k.mDescriptors, n, in(P + "desc" + S).u8());
    k.mvuRight.assign(n, -1.f);
    const float* g = in(P + "grid").f32();
    k.mnMinX = g[0]; k.mnMinY = g[1]; k.mnMaxX = g[2]; k.mnMaxY = g[3];
    k.mfGridElementWidthInv = g[4]; k.mfGridElementHeightInv = g[5];
    const Arr& sf = in(P + "sf");
    k.mvScaleFactors.assign(sf.f32(), sf.f32() + sf.count);
    k.mvInvLevelSigma2.resize(sf.count);
    for (size_t i = 0; i < sf.count; i++) k.mvInvLevelSigma2[i] = 1.0f / (k.mvScaleFactors[i] * k.mvScaleFactors[i]);
    k.mnScaleLevels = (int)sf.count;
    k.mfLogScaleFactor = in(P + "logsf").f32()[0];
    cam.mvParameters.assign(in(P + "cam").f32(), in(P + "cam").f32() + 4);
    k.mpCamera = &cam;
    k.fx = cam.mvParameters[0]; k.fy = cam.mvParameters[1]; k.cx = cam.mvParameters[2]; k.cy = cam.mvParameters[3];
}
static void set_point_geometry(MapPoint* p, const std::string& P, const char* sfx, size_t q)
{
    const std::string S(sfx);
    p->mWorldPos = m31(in(P + "ppos" + S).f32() + 3 * q);
    if (has(P + "pnormal" + S)) p->mNormalVector = m31(in(P + "pnormal" + S).f32() + 3 * q);
    p->mfMinDistance = in(P + "pdist" + S).f32()[2 * q];
    p->mfMaxDistance = in(P + "pdist" + S).f32()[2 * q + 1];
    memcpy(p->mDescriptor, in(P + "pdesc" + S).u8() + 32 * q, 32);
}
static void test_reloc(const std::string& P)
{
    Frame C;
    GeometricCamera cam, camK;
    fill_frame(C, P);
    cam.mvParameters.assign(in(P + "cam").f32(), in(P + "cam").f32() + 4);
    C.mpCamera = &cam;
    C.mRcw_ = m33(in(P + "Rc").f32());
    C.mtcw_ = m31(in(P + "tc").f32());
    C.mfLogScaleFactor = in(P + "logsf").f32()[0];
    C.mnScaleLevels = (int)in(P + "sf").count;
    KeyFrame k;
    const int nk = (int)in(P + "kstate").count;
    k.N = nk;
    set_keys(k.mvKeysUn, nk, nullptr, nullptr, in(P + "kang").f32(), nullptr);
    k.mvpMapPoints.assign(nk, nullptr);
    std::set<MapPoint*> found;
    const int32_t* ks = in(P + "kstate").i32(); // 0 none, 1 ok, 2 bad, 3 already found
    for (int i = 0; i < nk; i++) {
        if (!ks[i]) continue;
        MapPoint* p = k.mvpMapPoints[i] = new_point(i);
        p->mbBad = ks[i] == 2;
        if (ks[i] == 3) found.insert(p);
        set_point_geometry(p, P, "", (size_t)i);
    }
    ORBmatcher matcher(0.9f, in(P + "ori").i32()[0] != 0);
    const int nm = matcher.SearchByProjection(C, &k, found, in(P + "th").f32()[0], in(P + "orbdist").i32()[0]);
    dump_frame_points(C, P + "points");
    put_i(P + "n", std::vector<int32_t>(1, nm));
}

static void test_sim3_projection(const std::string& P)
{
    KeyFrame k;
    GeometricCamera cam;
    fill_keyframe(k, cam, P);
    const int n = k.N, m = (int)in(P + "pstate").count;
    std::vector<MapPoint*> pts(m);
    std::vector<KeyFrame> kfs(5);
    std::vector<KeyFrame*> ptKFs(m);
    const int32_t* ps = in(P + "pstate").i32(); // 1 ok, 2 bad
    for (int q = 0; q < m; q++) {
        pts[q] = new_point(q);
        pts[q]->mbBad = ps[q] == 2;
        set_point_geometry(pts[q], P, "", (size_t)q);
        ptKFs[q] = &kfs[q % 5];
    }
    std::vector<MapPoint*> vpMatched(n, nullptr);
    std::vector<KeyFrame*> vpMatchedKF(n, nullptr);
    const int32_t* pre = in(P + "mpre").i32(); // -1 none, >= 0 that candidate point, -2 a foreign point
    for (int i = 0; i < n; i++)
        if (pre[i] >= 0) vpMatched[i] = pts[pre[i]];
        else if (pre[i] == -2) vpMatched[i] = new_point(500000 + i);
    ORBmatcher matcher;
    const cv::Mat Scw = mat32(in(P + "Scw").f32(), 4, 4);
    int nm;
    if (in(P + "twin").i32()[0])
        nm = matcher.SearchByProjection(&k, Scw, pts, ptKFs, vpMatched, vpMatchedKF, in(P + "th").i32()[0], in(P + "ratio").f32()[0]);
    else
        nm = matcher.SearchByProjection(&k, Scw, pts, vpMatched, in(P + "th").i32()[0], in(P + "ratio").f32()[0]);
    std::vector<int32_t> out(n, -1), outKF(n, -1);
    for (int i = 0; i < n; i++) {
        if (vpMatched[i]) out[i] = (int32_t)vpMatched[i]->mnId;
        if (vpMatchedKF[i]) outKF[i] = (int32_t)(vpMatchedKF[i] - kfs.data());
    }
    put_i(P + "matched", out);
    put_i(P + "matchedKF", outKF);
    put_i(P + "n", std::vector<int32_t>(1, nm));
}

static void test_fuse_sim3(const std::string& P)
{
    KeyFrame k;
    GeometricCamera cam;
    fill_keyframe(k, cam, P);
    const int n = k.N, m = (int)in(P + "pstate").count;
    set_points(k.mvpMapPoints, n, in(P + "fstate").i32(), 100000);
    std::vector<MapPoint*> pts(m);
    const int32_t* ps = in(P + "pstate").i32(); // 1 ok, 2 bad, 3 one of the keyframe's own points
    for (int q = 0; q < m; q++) {
        if (ps[q] == 3) {
            pts[q] = k.mvpMapPoints[in(P + "own").i32()[q]];
            continue;
        }
        pts[q] = new_point(q);
        pts[q]->mbBad = ps[q] == 2;
        set_point_geometry(pts[q], P, "", (size_t)q);
    }
    std::vector<MapPoint*> vpReplacePoint(m, nullptr);
    ORBmatcher matcher;
    const int nf = matcher.Fuse(&k, mat32(in(P + "Scw").f32(), 4, 4), pts, in(P + "th").f32()[0], vpReplacePoint);
    std::vector<int32_t> repl(m, -1), obsIdx(m, -1), kfPoint(n, -1);
    for (int q = 0; q < m; q++) {
        if (vpReplacePoint[q]) repl[q] = (int32_t)vpReplacePoint[q]->mnId;
        if (ps[q] != 3) {
            auto it = pts[q]->mObservations.find(&k);
            if (it != pts[q]->mObservations.end()) obsIdx[q] = it->second;
        }
    }
    for (int i = 0; i < n; i++)
        if (k.mvpMapPoints[i]) kfPoint[i] = (int32_t)k.mvpMapPoints[i]->mnId;
    put_i(P + "replace", repl);
    put_i(P + "obsIdx", obsIdx);
    put_i(P + "kfPoint", kfPoint);
    put_i(P + "n", std::vector<int32_t>(1, nf));
}

static void test_search_by_sim3(const std::string& P)
{
    KeyFrame k1, k2;
    GeometricCamera c1, c2;
    fill_keyframe(k1, c1, P, "1");
    fill_keyframe(k2, c2, P, "2");
    k1.Rcw = m33(in(P + "R1").f32()); k1.tcw = m31(in(P + "t1").f32());
    k2.Rcw = m33(in(P + "R2").f32()); k2.tcw = m31(in(P + "t2").f32());
    auto points = [&](KeyFrame& k, const char* sfx, long idBase) {
        const std::string S(sfx);
        const int32_t* st = in(P + "kstate" + S).i32(); // 0 none, 1 ok, 2 bad
        k.mvpMapPoints.assign(k.N, nullptr);
        for (int i = 0; i < k.N; i++) {
            if (!st[i]) continue;
            MapPoint* p = k.mvpMapPoints[i] = new_point(idBase + i);
            p->mbBad = st[i] == 2;
            p->mObservations[&k] = i;
            set_point_geometry(p, P, sfx, (size_t)i);
        }
    };
    points(k1, "1", 300000);
    points(k2, "2", 400000);
    std::vector<MapPoint*> vpMatches12(k1.N, nullptr);
    const int32_t* pre = in(P + "pre12").i32(); // -1, or the index in keyframe 2 it is already matched to
    for (int i = 0; i < k1.N; i++)
        if (pre[i] >= 0) vpMatches12[i] = k2.mvpMapPoints[pre[i]];
    ORBmatcher matcher;
    const float s12 = in(P + "s12").f32()[0];
    const int nf = matcher.SearchBySim3(&k1, &k2, vpMatches12, s12, mat32(in(P + "R12").f32(), 3, 3), mat32(in(P + "t12").f32(), 3, 1),
                                        in(P + "th").f32()[0]);
    std::vector<int32_t> out(k1.N, -1);
    for (int i = 0; i < k1.N; i++)
        if (vpMatches12[i]) out[i] = (int32_t)vpMatches12[i]->mnId - 400000;
    put_i(P + "matches12", out);
    put_i(P + "n", std::vector<int32_t>(1, nf));
}

static void test_initialization(const std::string& P)
{
    Frame F1, F2;
    const int n1 = (int)in(P + "oct1").count, n2 = (int)in(P + "oct2").count;
    F1.N = n1;
    F2.N = n2;
    set_keys(F1.mvKeysUn, n1, nullptr, nullptr, in(P + "ang1").f32(), in(P + "oct1").i32());
    set_keys(F2.mvKeysUn, n2, in(P + "kx2").f32(), in(P + "ky2").f32(), in(P + "ang2").f32(), in(P + "oct2").i32());
    set_desc(F1.mDescriptors, n1, in(P + "desc1").u8());
    set_desc(F2.mDescriptors, n2, in(P + "desc2").u8());
    const float* g = in(P + "grid").f32();
    F2.mnMinX = g[0]; F2.mnMinY = g[1]; F2.mfGridElementWidthInv = g[2]; F2.mfGridElementHeightInv = g[3];
    std::vector<cv::Point2f> prev(n1);
    for (int i = 0; i < n1; i++) {
        prev[i].x = in(P + "prev").f32()[2 * i];
        prev[i].y = in(P + "prev").f32()[2 * i + 1];
    }
    std::vector<int> vnMatches12;
    ORBmatcher matcher(in(P + "ratio").f32()[0], in(P + "ori").i32()[0] != 0);
    const int nm = matcher.SearchForInitialization(F1, F2, prev, vnMatches12, in(P + "window").i32()[0]);
    std::vector<float> prevOut(2 * (size_t)n1);
    for (int i = 0; i < n1; i++) {
        prevOut[2 * i] = prev[i].x;
        prevOut[2 * i + 1] = prev[i].y;
    }
    put_i(P + "matches12", std::vector<int32_t>(vnMatches12.begin(), vnMatches12.end()));
    put(P + "prev", 2, prevOut.data(), prevOut.size());
    put_i(P + "n", std::vector<int32_t>(1, nm));
}

int main(int argc, char** argv)
{
    if (argc < 3 || !load(argv[1])) return 2;
    g_out = fopen(argv[2], "wb");
    if (!g_out) return 2;
    try {
        for (int k = 0; has("bow" + std::to_string(k) + ".a1"); k++) test_bow_kf_f("bow" + std::to_string(k) + ".");
        for (int k = 0; has("kk" + std::to_string(k) + ".a1"); k++) test_bow_kf_kf("kk" + std::to_string(k) + ".");
        for (int k = 0; has("tri" + std::to_string(k) + ".a1"); k++) test_tri("tri" + std::to_string(k) + ".");
        for (int k = 0; has("tk" + std::to_string(k) + ".a1"); k++) test_tri_kb8("tk" + std::to_string(k) + ".");
        for (int k = 0; has("p0_" + std::to_string(k) + ".kx"); k++) test_proj_local("p0_" + std::to_string(k) + ".");
        for (int k = 0; has("p1_" + std::to_string(k) + ".kx"); k++) test_proj_last("p1_" + std::to_string(k) + ".");
        for (int k = 0; has("fu" + std::to_string(k) + ".kx"); k++) test_fuse("fu" + std::to_string(k) + ".");
        for (int k = 0; has("p2_" + std::to_string(k) + ".kx"); k++) test_reloc("p2_" + std::to_string(k) + ".");
        for (int k = 0; has("s3_" + std::to_string(k) + ".kx"); k++) test_sim3_projection("s3_" + std::to_string(k) + ".");
        for (int k = 0; has("fs" + std::to_string(k) + ".kx"); k++) test_fuse_sim3("fs" + std::to_string(k) + ".");
        for (int k = 0; has("ss" + std::to_string(k) + ".kx1"); k++) test_search_by_sim3("ss" + std::to_string(k) + ".");
        for (int k = 0; has("in" + std::to_string(k) + ".oct1"); k++) test_initialization("in" + std::to_string(k) + ".");
    } catch (const std::exception& e) {
        fprintf(stderr, "adapter threw: %s\n", e.what());
        return 4;
    }
    fclose(g_out);
    return 0;
}
