// -*- c++ -*-
// SHADOWS the reference's src/rootba/options/visitable_options.hpp for the oracle/_ref build.
// TEST INFRASTRUCTURE ONLY (see Eigen/Dense in this directory).
//
// The reference declares its option structs (src/rootba/bal/solver_options.hpp,
// bal_residual_options.hpp) through visit_struct / wise_enum / flags macros so that a CLI, a TOML
// reader and a GUI can enumerate them. Those three libraries are un-vendored submodules and the
// enumeration machinery is outside the hot path (out of scope). Here the same macros
// expand to plain members: the option structs that the solver reads are still the reference's own
// declarations - names, types AND default values come from the reference's files - only the
// reflection layer is gone.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <glog/logging.h>

#include "rootba/util/assert.hpp"

namespace rootba {
struct OptionsBase {};
template <class Derived>
struct VisitableOptions : public OptionsBase {};

namespace ref_shim {
template <class T>
struct MetaInit {
  T value_{};
  template <class V>
  MetaInit& init(const V& v) {
    value_ = T(v);
    return *this;
  }
  MetaInit& init_default_construct() { return *this; }
  template <class A, class B>
  MetaInit& range(const A&, const B&) {
    return *this;
  }
  MetaInit& help(const char*) { return *this; }
  MetaInit& help(const std::string&) { return *this; }
  MetaInit& logscale() { return *this; }
  MetaInit& noop() { return *this; }
  template <class F>
  MetaInit& flags(const F&) {
    return *this;
  }
  const T& get() const { return value_; }
};
}  // namespace ref_shim
}  // namespace rootba

namespace wise_enum {
template <class E>
std::string to_string(E e) {
  return "enum(" + std::to_string(static_cast<long long>(e)) + ")";
}
}  // namespace wise_enum

// ---- enum declaration: entries are NAME or (NAME, value) --------------------------------------
#define RS_CAT(a, b) RS_CAT_I(a, b)
#define RS_CAT_I(a, b) a##b
#define RS_PROBE() ~, 1
#define RS_IS_PAREN_PROBE(...) RS_PROBE()
#define RS_CHECK_N(x, n, ...) n
#define RS_CHECK(...) RS_CHECK_N(__VA_ARGS__, 0, )
#define RS_IS_PAREN(x) RS_CHECK(RS_IS_PAREN_PROBE x)
#define RS_PAIR(n, v) n = v
#define RS_ENTRY_0(x) x
#define RS_ENTRY_1(x) RS_PAIR x
#define RS_ENTRY(x) RS_CAT(RS_ENTRY_, RS_IS_PAREN(x))(x)
#define RS_NARG_I(a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, N, ...) N
#define RS_NARG(...) RS_NARG_I(__VA_ARGS__, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0)
#define RS_FE_1(a) RS_ENTRY(a)
#define RS_FE_2(a, ...) RS_ENTRY(a), RS_FE_1(__VA_ARGS__)
#define RS_FE_3(a, ...) RS_ENTRY(a), RS_FE_2(__VA_ARGS__)
#define RS_FE_4(a, ...) RS_ENTRY(a), RS_FE_3(__VA_ARGS__)
#define RS_FE_5(a, ...) RS_ENTRY(a), RS_FE_4(__VA_ARGS__)
#define RS_FE_6(a, ...) RS_ENTRY(a), RS_FE_5(__VA_ARGS__)
#define RS_FE_7(a, ...) RS_ENTRY(a), RS_FE_6(__VA_ARGS__)
#define RS_FE_8(a, ...) RS_ENTRY(a), RS_FE_7(__VA_ARGS__)
#define RS_FE_9(a, ...) RS_ENTRY(a), RS_FE_8(__VA_ARGS__)
#define RS_FE_10(a, ...) RS_ENTRY(a), RS_FE_9(__VA_ARGS__)
#define WISE_ENUM_CLASS_MEMBER(name, ...) \
  enum class name { RS_CAT(RS_FE_, RS_NARG(__VA_ARGS__))(__VA_ARGS__) }
#define WISE_ENUM_CLASS(name, ...) enum class name { RS_CAT(RS_FE_, RS_NARG(__VA_ARGS__))(__VA_ARGS__) }

// ---- members -------------------------------------------------------------------------------------
#define BEGIN_VISITABLES(T) static_assert(true, "")
#define END_VISITABLES static_assert(true, "")
#define VISITABLE(TYPE, NAME) TYPE NAME
#define VISITABLE_INIT(TYPE, NAME, VALUE) TYPE NAME = VALUE
#define VISITABLE_META(TYPE, NAME, META_INITIALIZER) \
  TYPE NAME = ::rootba::ref_shim::MetaInit<TYPE>().META_INITIALIZER.get()
#define VISITABLE_META_DEFAULT(TYPE, NAME) TYPE NAME{}
#define VISITABLE_OPTIONS_DEFAULT_META(INITIALIZER) static_assert(true, "")
