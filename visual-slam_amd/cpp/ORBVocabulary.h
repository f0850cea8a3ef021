// Drop-in for what the reference uses of its vocabulary class: ORB_SLAM2::ORBVocabulary (include/ORBVocabulary.h:52-53,
// a DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) as Frame::ComputeBoW / KeyFrame::ComputeBoW call it
// (src/Frame.cc:395-402, src/KeyFrame.cc:77-87): loading (text and binary forms, Thirdparty/DBoW2/DBoW2/
// TemplatedVocabulary.h:1351-1508), transform() into a BowVector and a FeatureVector, size(), score().
// The tree lives in device memory (orbx_vocabulary); the descent runs there, the two small maps are built on the host.
// Vocabulary training (create(), HKmeansStep) is offline tooling of DBoW2 and is not rebuilt.
#ifndef ORBX_ORBVOCABULARY_H
#define ORBX_ORBVOCABULARY_H

#include <map>
#include <string>
#include <vector>
#include "cv_compat.h"
#include "../../include/orbx.h"

namespace DBoW2
{
typedef unsigned int WordId;    // BowVector.h:20-26
typedef double WordValue;
typedef unsigned int NodeId;
enum WeightingType { TF_IDF, TF, IDF, BINARY };                                           // BowVector.h:36-42
enum ScoringType { L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT };        // BowVector.h:45-53

class BowVector : public std::map<WordId, WordValue> {};                                  // BowVector.h:56-58
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {};             // FeatureVector.h:21-22
}

namespace ORB_SLAM2
{

class ORBVocabulary
{
public:
    ORBVocabulary();
    ~ORBVocabulary();
    ORBVocabulary(const ORBVocabulary&) = delete;
    ORBVocabulary& operator=(const ORBVocabulary&) = delete;

    // TemplatedVocabulary.h:1351-1437: "k L scoring weighting" then one line per node: parent, leaf flag, 32 byte values, weight
    bool loadFromTextFile(const std::string& filename);
    // TemplatedVocabulary.h:1467-1508: header (node count, record size, k, L, scoring, weighting) then fixed-size records
    bool loadFromBinaryFile(const std::string& filename);
    // From arrays (node 0 = root); what the two loaders end in
    bool loadFromArrays(int k, int L, DBoW2::ScoringType scoring, DBoW2::WeightingType weighting, int nNodes, const int* parent,
                        const unsigned char* isLeaf, const unsigned char* descriptors, const double* weight);

    // TemplatedVocabulary.h:1139-1207
    void transform(const std::vector<cv::Mat>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const;
    // the same for the rows of an N x 32 descriptor matrix (saves Converter::toDescriptorVector, src/Frame.cc:399)
    void transform(const cv::Mat& descriptors, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const;

    unsigned int size() const;                                    // number of words
    bool empty() const { return size() == 0; }
    int getBranchingFactor() const { return mk; }
    int getDepthLevels() const { return mL; }
    DBoW2::WeightingType getWeightingType() const { return mWeighting; }
    DBoW2::ScoringType getScoringType() const { return mScoring; }
    // ScoringObject.cpp, L1Scoring::score: the score LoopClosing and KeyFrameDatabase take of two BowVectors
    // (only the L1 form, the one ORBvoc is built with; other scoring types throw)
    double score(const DBoW2::BowVector& a, const DBoW2::BowVector& b) const;

    void SetDevice(int device) { mnDevice = device; }
    orbx_vocabulary* Handle() const { return mpVoc; }

private:
    void transformRows(const unsigned char* rows, int n, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const;
    orbx_vocabulary* mpVoc = nullptr;
    int mk = 0, mL = 0, mnDevice = -1;
    DBoW2::ScoringType mScoring = DBoW2::L1_NORM;
    DBoW2::WeightingType mWeighting = DBoW2::TF_IDF;
};

} // namespace ORB_SLAM2

#endif
