#include "ORBVocabulary.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace ORB_SLAM2
{

static void check(int status, const char* what)
{
    if (status != ORBX_OK)
        throw std::runtime_error(std::string(what) + ": " + orbx_status_string(status) + " (" + orbx_last_error() + ")");
}

ORBVocabulary::ORBVocabulary() {}
ORBVocabulary::~ORBVocabulary() { orbx_vocabulary_destroy(mpVoc); }

unsigned int ORBVocabulary::size() const { return (unsigned int)orbx_vocabulary_words(mpVoc); }

bool ORBVocabulary::loadFromArrays(int k, int L, DBoW2::ScoringType scoring, DBoW2::WeightingType weighting, int nNodes, const int* parent,
                                   const unsigned char* isLeaf, const unsigned char* descriptors, const double* weight)
{
    orbx_vocabulary* v = nullptr;
    if (orbx_vocabulary_create(mnDevice, k, L, (int)scoring, (int)weighting, nNodes, parent, isLeaf, descriptors, weight, &v) != ORBX_OK) return false;
    orbx_vocabulary_destroy(mpVoc);
    mpVoc = v; mk = k; mL = L; mScoring = scoring; mWeighting = weighting;
    return true;
}

bool ORBVocabulary::loadFromTextFile(const std::string& filename)
{
    FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) return false;
    std::vector<char> line(1 << 12);
    int k = 0, L = 0, n1 = -1, n2 = -1;
    if (!std::fgets(line.data(), (int)line.size(), f) || std::sscanf(line.data(), "%d %d %d %d", &k, &L, &n1, &n2) != 4 ||
        k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {       // TemplatedVocabulary.h:1372
        std::fclose(f);
        return false;
    }
    std::vector<int> parent(1, -1);
    std::vector<unsigned char> leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    while (std::fgets(line.data(), (int)line.size(), f)) {
        char* p = line.data();
        char* e = nullptr;
        const long pid = std::strtol(p, &e, 10);
        if (e == p) continue;                                    // blank line
        p = e;
        const long isLeaf = std::strtol(p, &e, 10);
        if (e == p) { std::fclose(f); return false; }
        p = e;
        unsigned char d[32];
        for (int i = 0; i < 32; ++i) {
            const long b = std::strtol(p, &e, 10);
            if (e == p) { std::fclose(f); return false; }
            d[i] = (unsigned char)b; p = e;
        }
        const double w = std::strtod(p, &e);
        if (e == p) { std::fclose(f); return false; }
        parent.push_back((int)pid); leaf.push_back(isLeaf > 0 ? 1 : 0); weight.push_back(w);
        desc.insert(desc.end(), d, d + 32);
    }
    std::fclose(f);
    return loadFromArrays(k, L, (DBoW2::ScoringType)n1, (DBoW2::WeightingType)n2, (int)parent.size(), parent.data(), leaf.data(),
                          desc.data(), weight.data());
}

bool ORBVocabulary::loadFromBinaryFile(const std::string& filename)
{
    FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) return false;
    unsigned int nbNodes = 0, sizeNode = 0;
    int k = 0, L = 0, scoring = 0, weighting = 0;
    bool ok = std::fread(&nbNodes, 4, 1, f) == 1 && std::fread(&sizeNode, 4, 1, f) == 1 && std::fread(&k, 4, 1, f) == 1 &&
              std::fread(&L, 4, 1, f) == 1 && std::fread(&scoring, 4, 1, f) == 1 && std::fread(&weighting, 4, 1, f) == 1;
    // record: parent (4 bytes), descriptor (32), weight as float (4), leaf flag (1), TemplatedVocabulary.h:1519, 1526-1532
    ok = ok && sizeNode >= 41 && nbNodes >= 2 && nbNodes < 0x7FFFFFFFu;
    if (!ok) { std::fclose(f); return false; }
    std::vector<int> parent(nbNodes, -1);
    std::vector<unsigned char> leaf(nbNodes, 0), desc((size_t)nbNodes * 32, 0), rec(sizeNode);
    std::vector<double> weight(nbNodes, 0.0);
    for (unsigned int i = 1; i < nbNodes; ++i) {
        if (std::fread(rec.data(), sizeNode, 1, f) != 1) { std::fclose(f); return false; }
        std::memcpy(&parent[i], rec.data(), 4);
        std::memcpy(&desc[(size_t)i * 32], rec.data() + 4, 32);
        float w; std::memcpy(&w, rec.data() + 36, 4);
        weight[i] = w;
        leaf[i] = rec[40] ? 1 : 0;
    }
    std::fclose(f);
    return loadFromArrays(k, L, (DBoW2::ScoringType)scoring, (DBoW2::WeightingType)weighting, (int)nbNodes, parent.data(), leaf.data(),
                          desc.data(), weight.data());
}

void ORBVocabulary::transformRows(const unsigned char* rows, int n, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const
{
    v.clear(); fv.clear();
    if (!mpVoc || n == 0) return;                                // empty vocabulary: both stay empty (TemplatedVocabulary.h:1147-1150)
    std::vector<uint32_t> ids((size_t)n), nodes((size_t)n);
    std::vector<double> vals((size_t)n);
    std::vector<int32_t> ptr((size_t)n + 1), feats((size_t)n);
    int32_t nw = 0, nn = 0;
    check(orbx_vocabulary_transform(mpVoc, rows, n, levelsup, ids.data(), vals.data(), &nw, nodes.data(), ptr.data(), feats.data(), &nn),
          "ORBVocabulary::transform");
    for (int i = 0; i < nw; ++i) v.insert(v.end(), std::make_pair(ids[i], vals[i]));
    for (int k = 0; k < nn; ++k) {
        std::vector<unsigned int>& list = fv.insert(fv.end(), std::make_pair(nodes[k], std::vector<unsigned int>()))->second;
        list.assign(feats.begin() + ptr[k], feats.begin() + ptr[k + 1]);
    }
}

void ORBVocabulary::transform(const std::vector<cv::Mat>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const
{
    std::vector<unsigned char> rows(features.size() * 32);
    for (size_t i = 0; i < features.size(); ++i) std::memcpy(&rows[i * 32], features[i].ptr(), 32);
    transformRows(rows.data(), (int)features.size(), v, fv, levelsup);
}

void ORBVocabulary::transform(const cv::Mat& descriptors, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const
{
    if (descriptors.rows > 0 && descriptors.isContinuous()) { transformRows(descriptors.ptr(), descriptors.rows, v, fv, levelsup); return; }
    std::vector<unsigned char> rows((size_t)descriptors.rows * 32);
    for (int i = 0; i < descriptors.rows; ++i) std::memcpy(&rows[(size_t)i * 32], descriptors.ptr(i), 32);
    transformRows(rows.data(), descriptors.rows, v, fv, levelsup);
}

double ORBVocabulary::score(const DBoW2::BowVector& a, const DBoW2::BowVector& b) const
{
    if (mScoring != DBoW2::L1_NORM) throw std::runtime_error("ORBVocabulary::score: only L1 scoring is provided");
    // over the words both vectors hold, in word order: sum |a_i - b_i| - |a_i| - |b_i|; the score is minus half of it
    double s = 0;
    DBoW2::BowVector::const_iterator ia = a.begin(), ib = b.begin();
    while (ia != a.end() && ib != b.end()) {
        if (ia->first == ib->first) { s += std::fabs(ia->second - ib->second) - std::fabs(ia->second) - std::fabs(ib->second); ++ia; ++ib; }
        else if (ia->first < ib->first) ia = a.lower_bound(ib->first);
        else ib = b.lower_bound(ia->first);
    }
    return -s / 2.0;
}

} // namespace ORB_SLAM2
