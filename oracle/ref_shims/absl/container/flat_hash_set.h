// -*- c++ -*-
// Stand-in for absl::flat_hash_set (used by the reference's sparsity statistic; TEST INFRASTRUCTURE ONLY).
#pragma once
#include <unordered_set>
namespace absl {
template <class T, class H = std::hash<T>, class E = std::equal_to<T>>
using flat_hash_set = std::unordered_set<T, H, E>;
}
