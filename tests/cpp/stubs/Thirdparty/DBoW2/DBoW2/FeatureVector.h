// Compile-only stand-in: DBoW2::FeatureVector is a std::map<NodeId, std::vector<unsigned int> >
// (Thirdparty/DBoW2/DBoW2/FeatureVector.h:21-22 of the reference).
#pragma once
#include <map>
#include <vector>
namespace DBoW2 {
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {};
}
