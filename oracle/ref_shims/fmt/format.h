// -*- c++ -*-
// Stand-in for the slice of {fmt} that the reference's log lines use (TEST INFRASTRUCTURE ONLY, see
// Eigen/Dense here): fmt::format with "{}" and "{:[[fill]align][width][.precision][type]}" fields,
// positional order only. Arithmetic arguments go through snprintf, everything else through operator<<.
#pragma once
#include <cstdio>
#include <functional>
#include <sstream>
#include <string>
#include <string_view>
#include <type_traits>
#include <vector>

#define FMT_VERSION 90000

namespace fmt {
inline std::string_view runtime(std::string_view s) { return s; }
namespace shim {
struct Spec {
  char fill = ' ', align = 0, type = 0;
  int width = -1, prec = -1;
  bool plus = false, zero = false;
};
inline Spec parse(std::string_view s) {
  Spec sp;
  size_t i = 0;
  auto is_align = [](char c) { return c == '<' || c == '>' || c == '^'; };
  if (s.size() >= 2 && is_align(s[1])) {
    sp.fill = s[0];
    sp.align = s[1];
    i = 2;
  } else if (!s.empty() && is_align(s[0])) {
    sp.align = s[0];
    i = 1;
  }
  if (i < s.size() && s[i] == '+') {
    sp.plus = true;
    ++i;
  }
  if (i < s.size() && s[i] == '0') {
    sp.zero = true;
    ++i;
  }
  if (i < s.size() && std::isdigit(static_cast<unsigned char>(s[i]))) {
    sp.width = 0;
    while (i < s.size() && std::isdigit(static_cast<unsigned char>(s[i]))) sp.width = sp.width * 10 + (s[i++] - '0');
  }
  if (i < s.size() && s[i] == '.') {
    ++i;
    sp.prec = 0;
    while (i < s.size() && std::isdigit(static_cast<unsigned char>(s[i]))) sp.prec = sp.prec * 10 + (s[i++] - '0');
  }
  if (i < s.size()) sp.type = s[i];
  return sp;
}
inline std::string pad(std::string v, const Spec& sp, bool numeric) {
  if (sp.width < 0 || int(v.size()) >= sp.width) return v;
  const size_t n = size_t(sp.width) - v.size();
  char a = sp.align ? sp.align : (numeric ? '>' : '<');
  const char fill = (sp.zero && !sp.align) ? '0' : sp.fill;
  if (a == '<') return v + std::string(n, fill);
  if (a == '>') return std::string(n, fill) + v;
  return std::string(n / 2, fill) + v + std::string(n - n / 2, fill);
}
template <class T>
std::string one(std::string_view spec, const T& v) {
  const Spec sp = parse(spec);
  if constexpr (std::is_same_v<T, bool>) {
    return pad(v ? "true" : "false", sp, false);
  } else if constexpr (std::is_floating_point_v<T>) {
    char buf[512];
    if (sp.type == 0 && sp.prec < 0) {
      std::ostringstream os;
      os.precision(std::is_same_v<T, float> ? 9 : 17);
      // shortest representation that round-trips is what {fmt} prints; %g with enough digits is close enough for logs
      std::snprintf(buf, sizeof buf, std::is_same_v<T, float> ? "%.7g" : "%.15g", double(v));
    } else {
      std::string f = "%";
      if (sp.plus) f += '+';
      if (sp.prec >= 0) f += "." + std::to_string(sp.prec);
      f += sp.type ? sp.type : 'g';
      std::snprintf(buf, sizeof buf, f.c_str(), double(v));
    }
    return pad(buf, sp, true);
  } else if constexpr (std::is_integral_v<T> && !std::is_same_v<T, char>) {
    char buf[128];
    if (sp.type == 'x')
      std::snprintf(buf, sizeof buf, "%llx", static_cast<unsigned long long>(v));
    else if constexpr (std::is_signed_v<T>)
      std::snprintf(buf, sizeof buf, sp.plus ? "%+lld" : "%lld", static_cast<long long>(v));
    else
      std::snprintf(buf, sizeof buf, "%llu", static_cast<unsigned long long>(v));
    return pad(buf, sp, true);
  } else {
    std::ostringstream os;
    os << v;
    return pad(os.str(), sp, false);
  }
}
}  // namespace shim

template <class... A>
std::string format(std::string_view f, const A&... a) {
  std::vector<std::function<std::string(std::string_view)>> args;
  (args.emplace_back([&a](std::string_view s) { return shim::one(s, a); }), ...);
  std::string out;
  size_t next = 0;
  for (size_t i = 0; i < f.size(); ++i) {
    if (f[i] == '{') {
      if (i + 1 < f.size() && f[i + 1] == '{') {
        out += '{';
        ++i;
        continue;
      }
      const size_t e = f.find('}', i);
      if (e == std::string_view::npos) break;
      std::string_view field = f.substr(i + 1, e - i - 1);
      size_t idx = next;
      const size_t colon = field.find(':');
      std::string_view id = colon == std::string_view::npos ? field : field.substr(0, colon);
      std::string_view spec = colon == std::string_view::npos ? std::string_view() : field.substr(colon + 1);
      if (!id.empty() && std::isdigit(static_cast<unsigned char>(id[0])))
        idx = size_t(std::stoi(std::string(id)));
      else
        ++next;
      out += idx < args.size() ? args[idx](spec) : std::string("{?}");
      i = e;
    } else if (f[i] == '}' && i + 1 < f.size() && f[i + 1] == '}') {
      out += '}';
      ++i;
    } else {
      out += f[i];
    }
  }
  return out;
}
template <class... A>
void print(std::string_view f, const A&... a) {
  std::fputs(format(f, a...).c_str(), stdout);
}
}  // namespace fmt
