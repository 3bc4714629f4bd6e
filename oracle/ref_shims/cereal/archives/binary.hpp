// -*- c++ -*-
// Placeholder for cereal's binary archives: the `.cereal` problem cache of the reference
// (src/rootba/bal/bal_problem_io.hpp) is NOT part of the oracle/_ref build - its byte layout is defined by
// cereal and by basalt-headers' Eigen/Sophus serialisers, none of which is available here.
// TEST INFRASTRUCTURE ONLY, see Eigen/Dense in this directory.
#pragma once
#include <iosfwd>
namespace cereal {
class BinaryOutputArchive {
 public:
  explicit BinaryOutputArchive(std::ostream&) {}
  template <class... T>
  void operator()(T&&...) {}
};
class BinaryInputArchive {
 public:
  explicit BinaryInputArchive(std::istream&) {}
  template <class... T>
  void operator()(T&&...) {}
};
}  // namespace cereal
