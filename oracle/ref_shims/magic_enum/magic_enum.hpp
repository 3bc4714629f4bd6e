// -*- c++ -*-
// Stand-in for magic_enum::enum_name (used in two log lines; TEST INFRASTRUCTURE ONLY).
#pragma once
#include <string>
namespace magic_enum {
template <class E>
std::string enum_name(E e) {
  return "enum(" + std::to_string(static_cast<long long>(e)) + ")";
}
}  // namespace magic_enum
