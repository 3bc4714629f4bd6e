// -*- c++ -*-
// Stand-in for the oneTBB calls of the reference's hot path (TBB headers are absent here).
// TEST INFRASTRUCTURE ONLY - see Eigen/Dense in this directory. Everything runs on the calling thread;
// ranges of 8 or more items are cut into up to four consecutive pieces so that the split constructors
// and join() functions of the reference's reduction bodies are exercised (a valid TBB schedule; TBB
// leaves the reduction order unspecified).
#pragma once
#include <cstddef>
#include <functional>
#include <unordered_map>
#include <utility>
#include <vector>

namespace tbb {
struct split {};
template <class T>
class blocked_range {
 public:
  using const_iterator = T;
  blocked_range(T b, T e, std::size_t grain = 1) : b_(b), e_(e), g_(grain) {}
  T begin() const { return b_; }
  T end() const { return e_; }
  std::size_t size() const { return std::size_t(e_ - b_); }
  bool empty() const { return !(b_ < e_); }
  std::size_t grainsize() const { return g_; }

 private:
  T b_, e_;
  std::size_t g_;
};
namespace shim {
template <class T>
std::vector<blocked_range<T>> pieces(const blocked_range<T>& r) {
  std::vector<blocked_range<T>> out;
  const std::size_t n = r.size();
  if (n < 8) {
    out.push_back(r);
    return out;
  }
  const std::size_t k = 4;
  T b = r.begin();
  for (std::size_t i = 0; i < k; ++i) {
    const T e = i + 1 == k ? r.end() : T(r.begin() + T((n * (i + 1)) / k));
    out.emplace_back(b, e);
    b = e;
  }
  return out;
}
// generic ranges (concurrent_unordered_map::range_type): one piece
template <class R>
std::vector<R> pieces(const R& r) {
  return {r};
}
}  // namespace shim

template <class Range, class Body>
void parallel_for(const Range& range, const Body& body) {
  for (const auto& p : shim::pieces(range)) body(p);
}
template <class Range, class Value, class Func, class Reduction>
Value parallel_reduce(const Range& range, const Value& identity, const Func& func, const Reduction& reduction) {
  const auto ps = shim::pieces(range);
  Value acc = func(ps[0], identity);
  for (std::size_t i = 1; i < ps.size(); ++i) acc = reduction(acc, func(ps[i], identity));
  return acc;
}
template <class Range, class Body>
void parallel_reduce(const Range& range, Body& body) {
  const auto ps = shim::pieces(range);
  body(ps[0]);
  for (std::size_t i = 1; i < ps.size(); ++i) {
    Body right(body, split());
    right(ps[i]);
    body.join(right);
  }
}

template <class K, class V, class H = std::hash<K>, class E = std::equal_to<K>>
class concurrent_unordered_map : public std::unordered_map<K, V, H, E> {
 public:
  using Base = std::unordered_map<K, V, H, E>;
  using Base::Base;
  using iterator = typename Base::iterator;
  using const_iterator = typename Base::const_iterator;
  class range_type {
   public:
    range_type(iterator b, iterator e) : b_(b), e_(e) {}
    iterator begin() const { return b_; }
    iterator end() const { return e_; }
    bool empty() const { return b_ == e_; }

   private:
    iterator b_, e_;
  };
  range_type range() { return range_type(this->begin(), this->end()); }
};
}  // namespace tbb
